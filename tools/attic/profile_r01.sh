#!/bin/bash
# rocprofv3 evidence for round 1 (run on the GPU box through gpurun): kernel trace + stats, then
# the two HBM counters in separate passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-realistic > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-realistic > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-realistic > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -20
python3 $R/tools/parse_rocprof.py $OUT
