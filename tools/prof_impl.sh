#!/bin/bash
# per-kernel time of one implementation on BG1 Zc=384 (GPU box): tools/prof_impl.sh <impl> [set-options]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_impl; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/perf_probe.py --spec ${SPEC:-nr5g:1:384} --impl $1 --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.565 --set ${2:-lanes=1} > $OUT/run.log 2>&1
grep -E "group" $OUT/run.log | tail -1 | cut -c1-150
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in rows[:14]:
    print(f"{r['Name'][:80]:80s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.1f}%")
PY
find $OUT -name "*kernel_trace.csv" -delete
