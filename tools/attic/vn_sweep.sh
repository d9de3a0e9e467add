#!/bin/bash
# headline configuration: variable-node kernel knobs (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for opts in lanes=1 nt_vn=1 "nt_vn=1,waves_vn=262144" "nt_vn=1,lfree_nt_in=1" "nt_vn=1,nt=0" "nt_vn=1,tile=128" "nt_vn=1,unroll_vn=4" "nt_vn=1,lfree_unroll=8" lanes=1 nt_vn=1; do
  echo -n "$opts  "
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 4096 --iters ${VN_ITERS:-20} --groups 4096 --reps 3 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c20-190
done
for impl in Minsumf64 Tanhf32 Minstarapproxi8; do for opts in lanes=0 nt_vn=1; do
  echo -n "$impl $opts  "
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl $impl --batch 4096 --iters 10 --groups 4096 --reps 2 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c20-190
done; done
for opts in lanes=0 nt_vn=1; do
  echo -n "nr5g Minsumf32 $opts  "
  python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl Minsumf32 --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.565 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c20-190
  echo -n "ar4ja Minsumf32 $opts  "
  python3 $R/tools/perf_probe.py --spec ar4ja:1/2:1024 --impl Minsumf32 --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.3 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c20-190
done
