#!/bin/bash
# flooding rules with one lane, two lanes, and two lanes half an iteration apart ("lane_skew"); GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
for impl in ${LANES_IMPLS:-Tanhf32 Aminstarf32 Phif32 Minstarapproxf32 Minstarapproxi8 Minsumf32}; do
 for opts in lanes=1 lanes=2 lanes=2,lane_skew=1; do
  echo -n "dvbs2 $impl $opts  "
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl $impl --batch 4096 --iters ${LANES_ITERS:-10} --groups 4096 --reps 2 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-120
 done
done
for impl in Tanhf32; do
 for opts in lanes=1 lanes=2,lane_skew=1; do
  echo -n "nr5g $impl $opts  "
  python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.565 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-120
 done
done
