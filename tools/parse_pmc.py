#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 output directories: `kernel_trace` csv -> calls / avg duration,
`counter_collection` csv -> average counter value per launch.  Usage:
  python3 tools/parse_pmc.py <dir> [<dir> ...] [--match substring]
Each <dir> is one rocprofv3 -d directory (one pass).  HBM bytes per launch are derived as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950
(FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read stream; unit KiB)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(?:ldpc::dev::)?(\w+_kernel)(<[^(]*>)?", name)
    if m:
        t = m.group(2) or ""
        t = re.sub(r"\(ldpc::dev::\w+\)", "", t)
        return (m.group(1) + t)[:70]
    return name.split("(")[0][:70]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    dur = defaultdict(list)
    ctr = defaultdict(lambda: defaultdict(list))
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                ctr[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    names = sorted(set(dur) | set(ctr), key=lambda k: -sum(dur.get(k, [0])))
    result = {}
    total = sum(sum(v) for v in dur.values()) or 1
    for k in names:
        if match and match not in k:
            continue
        line = f"{k}"
        rec = {}
        if k in dur:
            v = dur[k]
            rec.update(calls=len(v), avg_us=sum(v) / len(v) / 1e3, total_ms=sum(v) / 1e6, share=sum(v) / total)
            line += f"\n    calls {len(v)}  avg {rec['avg_us']:.1f} us  min {min(v)/1e3:.1f}  max {max(v)/1e3:.1f}  total {rec['total_ms']:.2f} ms  share {100*rec['share']:.1f}%"
        c = {name: sum(vals) / len(vals) for name, vals in ctr.get(k, {}).items()}
        if c:
            rec["counters_per_launch"] = c
            line += "\n    " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items()))
            if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
                rec["hbm_bytes_per_launch"] = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
                line += f"\n    HBM bytes per launch (2*FETCH+WRITE)*1024 = {rec['hbm_bytes_per_launch']:.4g}"
        result[k] = rec
        print(line)
    print("JSON " + json.dumps(result))


if __name__ == "__main__":
    main()
