#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for impl in HLTanhf32 HLMinsumf32 HLAminstari8; do
for opts in lanes=1 lanes=2 lanes=1,hl_reg=1; do
  echo -n "nr5g $impl $opts  "
  python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.565 --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-150
done; done
echo -n "nr5g HLTanhf32 batch 16384 lanes=2  "
python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 16384 --iters 10 --groups 16384 --reps 2 --sigma 1.565 --set lanes=2 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-150
echo -n "nr5g HLTanhf32 batch 32768 lanes=2  "
python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 32768 --iters 10 --groups 32768 --reps 2 --sigma 1.565 --set lanes=2 2>&1 | grep -E "group|Error" | tail -1 | cut -c1-150
