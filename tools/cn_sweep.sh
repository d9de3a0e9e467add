#!/bin/bash
# headline configuration: check-node kernel knobs (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for opts in lanes=1 waves=131072 waves=524288 waves=1048576 pad_kb=32 pad_kb=64 pad_kb=128 pad_kb=1024 tile=128 "tile=128,waves=524288" lfree_unroll=2 lanes=1; do
  echo -n "$opts  "
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 4096 --iters 20 --groups 4096 --reps 3 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c20-190
done
for opts in lanes=0 waves=131072 waves=524288 waves=1048576 block=128 hl_reg=0; do
  echo -n "HLTanhf32 $opts  "
  python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.565 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c20-150
done
