#!/bin/bash
# GPU box, round 5: config 3 (5G NR BG1 Zc=384 HLTanhf32, 8192 frames) -- the per-wave prologue diet (divisions by launch
# invariants as multiplications) against round 4's library, the waves-per-launch knob, and the SQ counters of the level kernels.
#   tools/r05_config3.sh [tag ...]     (extra library builds to time: ldpc_toolbox_amd/lib/libldpc_toolbox_<tag>.so)
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_config3; mkdir -p $OUT
P="python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 8192 --groups 8192 --sigma 1.565"
{
for rep in 1 2 3; do for tag in "" "$@"; do
  export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
  echo -n "${tag:-product} rep $rep: "; $P --iters 10 --reps 3 2>&1 | grep -E "group|Error" | cut -c1-170
done; done
unset LDPC_TOOLBOX_LIB
for w in 0 131072 65536 32768 16384 8192; do echo -n "waves=$w: "; $P --iters 10 --reps 3 --set waves=$w 2>&1 | grep -E "group|Error" | cut -c1-170; done
for impl in HLMinstarapproxf32 HLAminstarf32 HLPhif32 HLMinsumf32; do for tag in "" "$@"; do
  export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
  echo -n "$impl ${tag:-product}: "; python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --groups 8192 --sigma 1.565 --iters 10 --reps 3 2>&1 | grep -E "group|Error" | cut -c1-170
done; done
unset LDPC_TOOLBOX_LIB
} > $OUT/timing.txt 2>&1
MATCH=hl_level LINES=40 $R/tools/pmc_sets.sh $P --iters 4 --reps 1 > $OUT/counters.txt 2>&1
cat $OUT/timing.txt; head -30 $OUT/counters.txt
