#!/bin/bash
# where the time of an early-terminating batch goes (GPU box): tools/p2_trace.sh <spec> <impl> <ebn0_db> <batch> [key=value ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/p2_trace; rm -rf $OUT; mkdir -p $OUT
python3 $R/tools/p2_probe.py "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/p2_probe.py "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over 6 calls = {tot/6e6:.2f} ms per call")
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.1f}%")
PY
rm -rf $OUT/trace
