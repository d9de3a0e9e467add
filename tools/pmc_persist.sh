#!/bin/bash
# SQ counters of the slice-persistent layered kernel on config 3 (GPU box): tools/pmc_persist.sh [extra --set options] -- an experiments build (LDPC_TOOLBOX_LIB=...libldpc_toolbox_exp.so: the kernel left the product in round 5); every pass under its own timeout
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_persist; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl ${IMPL:-HLTanhf32} --batch 8192 --iters 4 --groups 8192 --reps 1 --sigma 1.565 --set lanes=1,hl_persist=2,hl_slice=${SLICE:-32}$1"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- $P > $OUT/t.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- $P > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/b -- $P > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/c -- $P > $OUT/c.log 2>&1
cat $OUT/t.log | tail -3
python3 $R/tools/parse_pmc.py $OUT/t $OUT/a $OUT/b $OUT/c --match ${MATCH:-hl_} | grep -v JSON | head -${LINES:-12}
rm -rf $OUT/t $OUT/a $OUT/b $OUT/c
