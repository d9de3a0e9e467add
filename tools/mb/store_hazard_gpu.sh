#!/bin/bash
# GPU box, round 5: (1) the standalone reproducer, (2) the determinism test at a size where stores queue up, on the product
# library and on the experiment builds named on the command line (tools/ab_variants.sh build <tag> ...), (3) what the pad
# costs: the kernels that store 128-bit packs through buffer descriptors, timed on each build.  Output: gpurun_out/store_hazard/
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$R/gpurun_out/store_hazard; mkdir -p $OUT
timeout 300 $R/tools/mb/store_hazard_repro > $OUT/repro.txt 2>&1; echo "repro rc=$?" >> $OUT/repro.txt
for tag in "" "$@"; do
  lib=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
  f=$OUT/determinism_${tag:-product}.txt
  echo "== determinism test (LDPC_DET_BATCH=${DET_BATCH:-4096}) on $lib" > $f
  LDPC_DET_BATCH=${DET_BATCH:-4096} LDPC_TOOLBOX_LIB=$lib timeout 1200 python3 -m pytest $R/tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | tail -25 | cut -c1-400 >> $f
done
for rep in 1 2; do for tag in "" "$@"; do
  export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
  echo "== ${tag:-product} (rep $rep)"
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf32 --batch 4096 --iters 50 --groups 4096 --reps 3 2>&1 | grep -E "group|Error" | cut -c1-200
  python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Minsumf64 --batch 4096 --iters 20 --groups 4096 --reps 2 2>&1 | grep -E "group|Error" | cut -c1-200
  python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLMinsumf32 --batch 8192 --iters 20 --groups 8192 --reps 2 --sigma 1.565 2>&1 | grep -E "group|Error" | cut -c1-200
  python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl Minsumf32 --batch 8192 --iters 20 --groups 8192 --reps 2 --sigma 1.565 2>&1 | grep -E "group|Error" | cut -c1-200
done; done > $OUT/pad_cost.txt 2>&1
unset LDPC_TOOLBOX_LIB
tail -3 $OUT/repro.txt; for f in $OUT/determinism_*.txt; do tail -n 3 $f; done; cat $OUT/pad_cost.txt
