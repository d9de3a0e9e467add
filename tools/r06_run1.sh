#!/bin/bash
# round 6, first GPU call: the new tests, the L2-gather probe, the full GPU suite, a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
python -m pytest tests/test_gpu_determinism.py -x -q -m gpu -k "finishes_inside" > gpurun_out/r06/t_vn.log 2>&1; echo "vn regression rc=$?" | tee -a gpurun_out/r06/summary.txt
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "headline_operating_point" > gpurun_out/r06/t_p1.log 2>&1; echo "headline point rc=$?" | tee -a gpurun_out/r06/summary.txt
python -m pytest tests/test_gpu_launcher.py -x -q -m gpu -s > gpurun_out/r06/t_launcher.log 2>&1; echo "launcher rc=$?" | tee -a gpurun_out/r06/summary.txt
for m in 0 1 2 3; do ./tools/mb/l2_gather_probe --mode $m; done > gpurun_out/r06/l2_probe.txt 2>&1
for m in 1 2; do ./tools/mb/l2_gather_probe --mode $m --rows 16200; ./tools/mb/l2_gather_probe --mode $m --rows 8100; done >> gpurun_out/r06/l2_probe.txt 2>&1
cat gpurun_out/r06/l2_probe.txt
python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all.log 2>&1; echo "full gpu suite rc=$?" | tee -a gpurun_out/r06/summary.txt
tail -3 gpurun_out/r06/t_all.log
python bench.py > gpurun_out/r06/bench1.json 2> gpurun_out/r06/bench1.err; echo "bench rc=$?" | tee -a gpurun_out/r06/summary.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/bench1.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "iter_us", d["roofline"]["iteration_us"], "realistic", d.get("realistic",{}).get("codewords_per_s"), d.get("realistic",{}).get("fraction_of_iteration_proportional_bound"))
print("config3", d.get("config3",{}).get("value"), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("matches_gpu_output"))
PY
