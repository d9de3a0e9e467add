#!/usr/bin/env python3
"""Condenses rocprofv3 csv output (kernel trace/stats + FETCH_SIZE / WRITE_SIZE passes) into
profiles/: a per-kernel summary and profiles/hbm_traffic.json (HBM bytes per launch, corrected as
/opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes: counters are in KiB-units of the
memory-side requests; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced
read stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for key in ("cn_minsum_lfree_kernel", "cn_minsum_kernel", "hl_minsum_kernel", "vn_kernel", "cn_staged_kernel", "hl_level_kernel", "ingest_kernel",
                "emit_kernel", "pack_hard_kernel", "syndrome_bits_kernel", "latch_kernel", "init_group_kernel"):
        if key in name:
            return key
    return name[:60]


def main():
    out = sys.argv[1]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(repo, "profiles")
    os.makedirs(prof, exist_ok=True)
    lines = []
    # ---- kernel trace: duration per kernel ------------------------------------------------
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    total = sum(sum(v) for v in dur.values()) or 1
    lines.append("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline")
    lines.append(f"{'kernel':28s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'share':>7s}")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k:28s} {len(v):7d} {sum(v)/1e6:10.2f} {sum(v)/len(v)/1e3:10.1f} {min(v)/1e3:10.1f} "
                     f"{max(v)/1e3:10.1f} {100*sum(v)/total:6.1f}%")
    # ---- PMC passes -------------------------------------------------------------------------
    traffic = {}
    for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == counter:
                    acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            traffic.setdefault(k, {})[counter] = sum(v) / len(v)
            traffic[k][counter + "_launches"] = len(v)
    lines.append("")
    lines.append("HBM counters per launch (separate --pmc passes; raw counter unit = KiB):")
    lines.append(f"{'kernel':28s} {'FETCH_SIZE':>14s} {'WRITE_SIZE':>14s} {'bytes (2*F + W)*1024':>22s}")
    result = {}
    for k, c in sorted(traffic.items()):
        fs, ws = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        b = (2.0 * fs + ws) * 1024.0
        result[k + "_bytes_per_launch"] = b
        result[k + "_raw"] = {"FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": ws}
        lines.append(f"{k:28s} {fs:14.1f} {ws:14.1f} {b:22.0f}")
    result["correction"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)"
    text = "\n".join(lines) + "\n"
    open(os.path.join(out, "summary_r01.txt"), "w").write(text)
    json.dump(result, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
    print(text)


if __name__ == "__main__":
    main()
