#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for opts in lanes=1 lanes=2 lanes=2,lane_skew=1; do
  echo -n "nr5g Tanhf32 50it $opts  "
  python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl Tanhf32 --batch 8192 --iters 50 --groups 8192 --reps 2 --sigma 1.565 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-120
  echo -n "dvbs2 Tanhf32 2dB $opts  "
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Tanhf32 --batch 4096 --iters 50 --groups 4096 --reps 2 --sigma 0.794 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-140
  echo -n "dvbs2 Tanhf32 1.0dB $opts  "
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Tanhf32 --batch 4096 --iters 50 --groups 4096 --reps 2 --sigma 0.891 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-140
  echo -n "ar4ja Tanhf32 50it $opts  "
  python3 $R/tools/perf_probe.py --spec ar4ja:1/2:1024 --impl Tanhf32 --batch 8192 --iters 50 --groups 8192 --reps 2 --sigma 1.3 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-120
done
