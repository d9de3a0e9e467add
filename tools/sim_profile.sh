#!/bin/bash
# kernel-time breakdown of the simulation step (GPU box): tools/sim_profile.sh [ebn0_db]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/sim_prof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/sim_profile_probe.py ${1:-2.0} 6 > $OUT/run.log 2>&1
grep "frames/s" $OUT/run.log
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over 7 batches")
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.1f}%")
PY
find $OUT -name "*kernel_trace.csv" -delete
