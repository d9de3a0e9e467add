R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_hl
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLMinsumf32 --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.8 > $OUT/log.txt 2>&1
grep group $OUT/log.txt
python3 $R/tools/trace_gaps.py $OUT
