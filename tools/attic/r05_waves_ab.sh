#!/bin/bash
# GPU box, round 5: the level launches' wave count on config 3 -- a wavefront of the default launch (256 K waves asked for)
# takes ONE check row and pays the kernel prologue for it; fewer waves take several rows each.  Alternating A/B, fixed work
# (10 iterations) and early termination (+2 dB through tools/p2_probe.py).
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_waves; mkdir -p $OUT
P="python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 8192 --groups 8192 --sigma 1.565 --iters 10 --reps 3"
{
for rep in 1 2 3 4; do for w in 0 32768 16384 8192 4096; do echo -n "fixed work waves=$w rep $rep: "; $P --set waves=$w 2>&1 | grep -E "group|Error" | cut -c1-110; done; done
for rep in 1 2 3; do for w in 0 16384 8192; do echo -n "waves=$w rep $rep: "; python3 $R/tools/p2_probe.py nr5g:1:384 HLTanhf32 2.0 8192 waves=$w throttle=1 2>&1 | tail -1 | cut -c1-200; done; done
for rep in 1 2; do for w in 0 8192; do for impl in HLMinsumf32 HLAminstarf32; do echo -n "$impl waves=$w rep $rep: "; python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --groups 8192 --sigma 1.565 --iters 10 --reps 3 --set waves=$w 2>&1 | grep -E "group|Error" | cut -c1-110; done; done; done
} > $OUT/waves.txt 2>&1
cat $OUT/waves.txt
