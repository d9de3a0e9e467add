#!/bin/bash
# GPU box, round 5: two-lane flooding calls under option throttle -- one enqueuing thread for both lanes, unpaced (lane_threads=0:
# rounds 3-4) against a thread per lane, each following its own group with tail checkpoints (lane_threads=1).  Alternating.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_flood_lanes; mkdir -p $OUT
{
for rep in 1 2 3; do for lt in 0 1; do
  for case in "dvbs2:R3_5 Minsumf32 2.6 8192" "dvbs2:R9_10 Minsumf32 4.2 8192" "dvbs2:R2_3 Minsumf32 2.9 8192" "nr5g:1:384 Minsumf32 2.0 8192" "dvbs2:R1_2 Tanhf32 2.0 2048"; do
    echo -n "lane_threads=$lt rep $rep: "; python3 $R/tools/p2_probe.py $case throttle=1 lane_threads=$lt 2>&1 | tail -1 | cut -c1-200
  done
done; done
for lt in 0 1; do echo -n "fixed work lane_threads=$lt: "; python3 $R/tools/perf_probe.py --spec dvbs2:R3_5 --impl Minsumf32 --batch 8192 --iters 20 --groups 4096 --reps 3 --set throttle=1,lane_threads=$lt 2>&1 | grep -E "group|Error" | cut -c1-150; done
} > $OUT/lanes.txt 2>&1
cat $OUT/lanes.txt
