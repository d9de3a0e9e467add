python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/bench_rules.py > gpurun_out/rules42.txt 2>&1; grep -v amdgpu gpurun_out/rules42.txt
