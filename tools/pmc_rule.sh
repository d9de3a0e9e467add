#!/bin/bash
# SQ counters of one rule's check-node / level kernel (GPU box): tools/pmc_rule.sh <spec> <impl> <batch> <sigma>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_rule; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/perf_probe.py --spec $1 --impl $2 --batch $3 --iters 4 --groups $3 --reps 1 --sigma $4"
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- $P > $OUT/t.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- $P > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $OUT/b -- $P > $OUT/b.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/t $OUT/a $OUT/b --match ${5:-kernel} | grep -v JSON | head -${6:-12}
rm -rf $OUT/t $OUT/a $OUT/b
