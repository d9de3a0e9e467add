#!/bin/bash
# GPU box, round 5: the first convergences' L-free rebuild inside the variable-node launch (vn_event=1, the default) against a
# launch of its own behind it in every iteration (vn_event=0: rounds 3-4); headline at fixed work and config 2 at +2 dB, alternating.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_vn_event; mkdir -p $OUT
{
for rep in 1 2 3 4; do for ev in 0 1; do
  echo -n "fixed work vn_event=$ev rep $rep: "; python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 4096 --iters 50 --groups 4096 --reps 3 --set vn_event=$ev 2>&1 | grep -E "group|Error" | cut -c1-170
done; done
for rep in 1 2 3; do for ev in 0 1; do
  echo -n "+2 dB vn_event=$ev rep $rep: "; python3 $R/tools/p2_probe.py dvbs2:R1_2 Minsumf32 2.0 4096 throttle=1 vn_event=$ev 2>&1 | tail -1 | cut -c38-140
done; done
} > $OUT/vn_event.txt 2>&1
cat $OUT/vn_event.txt
