#!/usr/bin/env python3
"""One early-termination call of BASELINE config 2 (DVB-S2 1/2 Minsumf32, 4096 frames, +2 dB) for a kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/p2t -- python3 tools/p2_timeline.py
  python3 tools/p2_timeline.py --read gpurun_out/p2t     (the launches of the LAST call in order, durations and gaps)"""
import os, sys, glob, csv
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if "--read" in sys.argv:
    d = sys.argv[sys.argv.index("--read") + 1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ing = [i for i, r in enumerate(rows) if "ingest_kernel" in r["Kernel_Name"]]
    last = ing[-1]
    if "--lanes" in sys.argv:   # a two-lane call: start at the first ingest of the last call (two ingests per call)
        last = ing[-2]
        t0 = int(rows[last]["Start_Timestamp"])
        import re
        print("columns:", [k for k in rows[0].keys()][:14])
        key = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
        for r in rows[last:]:
            name = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
            nm = (name.group(1) + (name.group(2) or "")[:22]) if name else r["Kernel_Name"][:30]
            s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print(f"{(s0 - t0) / 1e3:9.1f} us  {(e0 - s0) / 1e3:8.1f} us  q {r.get(key)}  {nm}")
        sys.exit(0)
    t0 = int(rows[last]["Start_Timestamp"])
    prev_end = t0
    import re
    it = 0
    for r in rows[last:]:
        name = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = name.group(1) if name else r["Kernel_Name"][:30]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if name.startswith("cn_"): it += 1
        print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  it {it:2d}  {name}")
        prev_end = e
    print(f"total {(prev_end - t0) / 1e3:.1f} us")
    sys.exit(0)
import torch
import ldpc_toolbox_amd as lt
import bench
dev = torch.device("cuda:0")
if "--config3" in sys.argv:   # BASELINE config 3 at +2 dB: 5G NR BG1 Zc=384 HLTanhf32, 8192 frames
    spec, impl, batch = "nr5g:1:384", "HLTanhf32", 8192
else:
    spec, impl, batch = "dvbs2:R1_2", "Minsumf32", 4096
alist = lt.code_alist(spec)
dec = lt.LdpcDecoder(alist, impl, device=0)
s = torch.cuda.Stream(device=dev)
# (bench.realistic_point: frames from the library's generator, option throttle = 1, three synchronised calls)
print(bench.realistic_point(dec, alist, impl, batch, dev, 0, s))
