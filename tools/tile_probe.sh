#!/bin/bash
# headline code with early termination: layout tile size (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for sig in 0.794 0.8318 0.7079 1.0; do
for opts in lanes=1 tile=128 tile=64 tile=512; do
  echo -n "sigma $sig $opts  "
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 4096 --iters 50 --groups 4096 --reps 3 --sigma $sig --set $opts 2>&1 | grep -E "group|Error" | tail -1 | cut -c20-150
done; done
