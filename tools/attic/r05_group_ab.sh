#!/bin/bash
# GPU box, round 5: does a larger group help the early-terminating regime?  (The tail of a group -- partial tiles, re-packing,
# launch-bound last iterations -- is a fixed number of tiles; more codewords per group make it a smaller share.)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_group; mkdir -p $OUT
{
for rep in 1 2; do
for g in 4096 8192 16384 32768; do echo -n "config 2, 32768 frames, group $g: "; P2_DISTINCT=4096 python3 $R/tools/p2_probe.py dvbs2:R1_2 Minsumf32 2.0 32768 group_size=$g throttle=1 2>&1 | tail -1 | cut -c1-220; done
for g in 8192 16384 32768; do echo -n "config 3, 65536 frames, group $g: "; P2_DISTINCT=4096 python3 $R/tools/p2_probe.py nr5g:1:384 HLTanhf32 2.0 65536 group_size=$g throttle=1 2>&1 | tail -1 | cut -c1-220; done
done
echo "fixed work at the same group sizes (perf_probe, 10 iterations):"
for g in 4096 16384; do python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 16384 --iters 10 --groups $g --reps 2 2>&1 | grep -E "group|Error" | cut -c1-150; done
} > $OUT/group.txt 2>&1
cat $OUT/group.txt
