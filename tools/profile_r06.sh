#!/bin/bash
# rocprofv3 evidence for round 6 (run on the GPU box through gpurun).  Usage: tools/profile_r06.sh [tag]
# Passes (counters never combined with trace domains other than --kernel-trace; TCC has 4 slots:
# FETCH_SIZE takes 3, WRITE_SIZE 2, so they are separate passes; SQ has 8 slots per pass):
#   head_*   the headline command (bench.py, DVB-S2 1/2 Minsumf32)
#   c3_*     BASELINE config 3 (5G NR BG1 Zc=384, HLTanhf32, 8192 frames) through tools/perf_probe.py
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--lanes 1: every traced / counted launch is a whole-group launch, like the bracketed region the roofline block reads)
HEAD="python3 $R/bench.py --no-cpu-baseline --no-realistic --no-config3 --no-live-traffic --lanes 1 "
C3="python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.565"
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
timeout 120 rocprofv3 -L > $OUT/counters_available.txt 2>&1
if [ "$2" != "c3only" ]; then
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/head_trace -- $HEAD --steps 3 --warmup 1 > $OUT/head_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/head_fetch -- $HEAD --steps 1 --warmup 0 > $OUT/head_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/head_write -- $HEAD --steps 1 --warmup 0 > $OUT/head_write.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/head_trace $OUT/head_fetch $OUT/head_write > $OUT/head_summary.txt 2>&1
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_trace -- $C3 > $OUT/c3_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c3_fetch -- $C3 > $OUT/c3_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c3_write -- $C3 > $OUT/c3_write.log 2>&1
timeout 600 rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/c3_sq1 -- $C3 > $OUT/c3_sq1.log 2>&1
timeout 600 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/c3_sq2 -- $C3 > $OUT/c3_sq2.log 2>&1
python3 $R/tools/parse_pmc.py $OUT/c3_trace $OUT/c3_fetch $OUT/c3_write $OUT/c3_sq1 $OUT/c3_sq2 > $OUT/c3_summary.txt 2>&1
python3 $R/tools/summarize_r06.py $OUT $OUT/summary > $OUT/summarize.log 2>&1
# keep only the summaries and logs small enough to travel back
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -5 $OUT/*.log
head -60 $OUT/c3_summary.txt
