python bench.py > gpurun_out/bench38.json 2> gpurun_out/bench38.err; tail -1 gpurun_out/bench38.json
python tools/bench_rules.py > gpurun_out/rules38.txt 2>&1; tail -60 gpurun_out/rules38.txt
bash tools/profile_r01.sh 2>&1 | tail -40
