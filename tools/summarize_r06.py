#!/usr/bin/env python3
"""Condenses one tools/profile_r06.sh run (gpurun_out/prof_<tag>/) into the committed evidence:
  profiles/r06_rocprofv3_summary.txt     headline command: per-kernel durations + HBM counters
  profiles/r06_config3_counters.txt      config 3 (HLTanhf32): per-kernel durations, HBM and SQ counters
  profiles/hbm_traffic.json              the per-launch figures bench.py quotes (with their provenance)
Usage: python3 tools/summarize_r06.py gpurun_out/prof_r06 [output directory, default profiles/]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(dirs, match=None):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "parse_pmc.py")] + dirs + (["--match", match] if match else [])
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
    text = "\n".join(ln for ln in out.splitlines() if not ln.startswith("JSON "))
    data = json.loads([ln for ln in out.splitlines() if ln.startswith("JSON ")][0][5:])
    return text, data


def main():
    d = sys.argv[1]
    prof = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    head_text, head = parse([os.path.join(d, x) for x in ("head_trace", "head_fetch", "head_write")])
    c3_text, c3 = parse([os.path.join(d, x) for x in ("c3_trace", "c3_fetch", "c3_write", "c3_sq1", "c3_sq2")])
    open(os.path.join(prof, "r06_rocprofv3_summary.txt"), "w").write(
        "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-realistic --no-config3 --no-live-traffic --lanes 1\n"
        "(+ separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of the same command with --steps 1 --warmup 0; tools/profile_r06.sh)\n\n"
        + head_text + "\n")
    open(os.path.join(prof, "r06_config3_counters.txt"), "w").write(
        "BASELINE config 3: 5G NR BG1 Zc=384, HLTanhf32, 8192 frames, 10 iterations, fixed work (tools/perf_probe.py --sigma 1.565),\n"
        "one rocprofv3 pass per counter set (tools/profile_r06.sh): kernel trace; FETCH_SIZE; WRITE_SIZE; two SQ sets.\n"
        "Counters are per launch (one dependency level of one 4096-codeword execution lane), summed over the chip; the call counts include the probe's warm-up pass (2 passes x 10 iterations x 2 lanes).\n\n"
        + c3_text + "\n")
    out = {"collected": "round 6, tools/profile_r06.sh + tools/summarize_r06.py",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)"}
    for key in ("cn_minsum_rec_kernel", "cn_minsum_lfree_kernel", "vn_kernel", "vn_free_rec_kernel"):
        # the template variant that makes the bulk of the launches (FIRST = false)
        cands = [(rec.get("calls", 0), name, rec) for name, rec in head.items() if name.startswith(key) and "hbm_bytes_per_launch" in rec]
        if cands:
            calls, name, rec = max(cands, key=lambda c: c[0])
            out[key + "_bytes_per_launch"] = rec["hbm_bytes_per_launch"]
            out[key + "_avg_us"] = rec.get("avg_us")
            out[key + "_variant"] = name
    # config 3: the level kernels of the iterations after the first (FIRST = false), weighted by their launch counts
    tot_bytes = tot_valu = tot_salu = tot_launch = 0.0
    for name, rec in c3.items():
        if name.startswith("hl_level_reg_kernel") and name.rstrip(">").endswith("false") and "counters_per_launch" in rec:
            calls = rec.get("calls", 0)
            c = rec["counters_per_launch"]
            tot_launch += calls
            tot_bytes += calls * rec.get("hbm_bytes_per_launch", 0.0)
            tot_valu += calls * c.get("SQ_INSTS_VALU", 0.0)
            tot_salu += calls * c.get("SQ_INSTS_SALU", 0.0)
            out[name + "_per_launch"] = {"calls": calls, "avg_us": rec.get("avg_us"), "hbm_bytes": rec.get("hbm_bytes_per_launch"),
                                         "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": c.get("SQ_INSTS_SALU"),
                                         "SQ_WAVES": c.get("SQ_WAVES"), "SQ_WAIT_INST_LDS": c.get("SQ_WAIT_INST_LDS")}
    if tot_launch:
        out["hl_level_reg_kernel_tanh_bytes_per_launch"] = tot_bytes / tot_launch
        out["hl_level_reg_kernel_tanh_valu_wave_insts_per_launch"] = tot_valu / tot_launch
        out["hl_level_reg_kernel_tanh_salu_wave_insts_per_launch"] = tot_salu / tot_launch
    json.dump(out, open(os.path.join(prof, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
