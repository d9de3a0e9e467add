#!/bin/bash
# Kernel trace + counter passes of one python command on the GPU box, per-kernel averages printed:
#   tools/pmc_cmd.sh <match> <lines> <script.py> [args...]
# passes: kernel trace; FETCH_SIZE; WRITE_SIZE; two SQ sets; TCC hit/miss (each its own run, as gpurun requires)
R=${GRAFT_REPO_ROOT:-/root/repo}
MATCH=$1; LINES=$2; shift 2
OUT=$R/gpurun_out/pmc_cmd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 "$@" > $OUT/t.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 "$@" > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 "$@" > $OUT/w.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/c -- python3 "$@" > $OUT/c.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/t $OUT/f $OUT/w $OUT/a $OUT/c --match "$MATCH" | grep -v JSON | head -$LINES
rm -rf $OUT/t $OUT/f $OUT/w $OUT/a $OUT/c
