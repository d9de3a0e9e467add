#!/bin/bash
# GPU box, round 5: (1) the headline's check-node launch, round 4's library against round 5's (publishing the progress word in
# cn_minsum_rec_kernel must cost nothing); (2) blended pack stores for partially frozen slices (-DLDPC_BLEND_STORES) at the
# early-terminating operating points; alternating runs.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_blend; mkdir -p $OUT
{
for rep in 1 2 3; do for tag in r4 "" blend; do
  export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
  echo -n "${tag:-r5} fixed work rep $rep: "; python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 4096 --iters 50 --groups 4096 --reps 3 2>&1 | grep -E "group|Error" | cut -c1-170
done; done
for rep in 1 2 3; do for tag in "" blend; do
  export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
  echo -n "${tag:-r5} rep $rep: "; python3 $R/tools/p2_probe.py dvbs2:R1_2 Minsumf32 2.0 4096 throttle=1 2>&1 | tail -1 | cut -c38-140
  echo -n "${tag:-r5} rep $rep: "; python3 $R/tools/p2_probe.py dvbs2:R3_5 Minsumf32 2.6 4096 throttle=1 2>&1 | tail -1 | cut -c38-140
  echo -n "${tag:-r5} rep $rep: "; python3 $R/tools/p2_probe.py nr5g:1:384 Minsumf32 2.0 8192 throttle=1 2>&1 | tail -1 | cut -c38-140
  echo -n "${tag:-r5} rep $rep: "; python3 $R/tools/p2_probe.py dvbs2:R1_2 Tanhf32 2.0 4096 throttle=1 2>&1 | tail -1 | cut -c38-140
done; done
export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox_blend.so
timeout 900 python3 -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_stress.py -m gpu -q -x -k "compaction or records or random_configuration or early or minsum" 2>&1 | tail -3
unset LDPC_TOOLBOX_LIB
} > $OUT/blend.txt 2>&1
cat $OUT/blend.txt
