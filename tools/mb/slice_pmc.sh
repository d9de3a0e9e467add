#!/bin/bash
# SQ counters of one slice_bench variant (GPU box): tools/mb/slice_pmc.sh <variant> [slice_bench args after the graph]
R=$(cd "$(dirname "$0")/../.." && pwd)
V=$1; shift
OUT=$R/gpurun_out/slice_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/tools/mb/slice_bench_$V $R/tools/mb/graph_bg1_384.bin $*"
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- $B > $OUT/t.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $OUT/c -- $B > $OUT/c.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/t $OUT/a $OUT/b $OUT/c --match hl_slice | grep -v JSON | head -8
rm -rf $OUT/t $OUT/a $OUT/b $OUT/c
