#!/bin/bash
# kernel timeline of a two-lane flooding run (GPU box): tools/lane_trace.sh <impl> <set-options>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/lane_trace_$1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl $1 --batch 4096 --iters 8 --groups 4096 --reps 1 --set $2 > $OUT/run.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Queue_Id','?')) for r in rows if 'cn_staged' in r['Kernel_Name'] or 'vn_kernel' in r['Kernel_Name']]
ks.sort()
ks=ks[-44:]
t0=ks[0][0]
for s,e,n,q in ks:
    tag='CN' if 'cn_staged' in n else 'VN'
    print(f"{tag} q{q} start {(s-t0)/1e3:9.1f} us  end {(e-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}")
PY
rm -rf $OUT/trace
