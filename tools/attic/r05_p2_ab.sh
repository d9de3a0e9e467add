#!/bin/bash
# GPU box, round 5: the early-terminating call of config 2 (and config 3) -- round 4's library against round 5's, with the
# paced flooding host / adaptive tail checkpoints (option throttle) and without; alternating runs, tools/p2_probe.py.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_p2; mkdir -p $OUT
{
for rep in 1 2 3; do
  for tag in r4 ""; do for thr in 0 1; do
    export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
    echo -n "${tag:-r5} throttle=$thr rep $rep: "; python3 $R/tools/p2_probe.py dvbs2:R1_2 Minsumf32 2.0 4096 throttle=$thr 2>&1 | tail -1 | cut -c38-200
  done; done
done
for rep in 1 2; do for tag in r4 ""; do
  export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox${tag:+_$tag}.so
  echo -n "${tag:-r5} rep $rep: "; python3 $R/tools/p2_probe.py nr5g:1:384 HLTanhf32 2.0 8192 throttle=1 2>&1 | tail -1 | cut -c38-200
  echo -n "${tag:-r5} rep $rep: "; python3 $R/tools/p2_probe.py dvbs2:R3_5 Minsumf32 2.6 4096 throttle=1 2>&1 | tail -1 | cut -c38-200
  echo -n "${tag:-r5} rep $rep: "; python3 $R/tools/p2_probe.py nr5g:1:384 Minsumf32 2.0 8192 throttle=1 2>&1 | tail -1 | cut -c38-200
done; done
unset LDPC_TOOLBOX_LIB
} > $OUT/p2.txt 2>&1
cat $OUT/p2.txt
