#!/bin/bash
# SQ counter sets (validated on this pool, one rocprofv3 pass each, every pass under its own timeout) for the kernels whose
# name contains MATCH (GPU box):   MATCH=hl_level tools/pmc_sets.sh python3 tools/perf_probe.py ...
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/pmc_sets; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="$*"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- $P > $OUT/t.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- $P > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/b -- $P > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --output-format csv -d $OUT/c -- $P > $OUT/c.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/t $OUT/a $OUT/b $OUT/c --match ${MATCH:-_kernel} | grep -v JSON | head -${LINES:-16}
rm -rf $OUT/t $OUT/a $OUT/b $OUT/c
