#!/bin/bash
# round 6, fifth GPU call: the final build -- suite, long stress, parity sweep, rules table, bench line, the eight-rank rehearsal line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > gpurun_out/r06/t_all5.log 2>&1; echo "full gpu suite rc=$?" | tee -a gpurun_out/r06/summary5.txt
tail -4 gpurun_out/r06/t_all5.log | cut -c1-300
LDPC_STRESS_SEEDS=3000 python -m pytest tests/test_gpu_stress.py -q -m gpu -x > gpurun_out/r06/t_stress5.log 2>&1; echo "3000 stress seeds rc=$?" | tee -a gpurun_out/r06/summary5.txt
tail -3 gpurun_out/r06/t_stress5.log | cut -c1-300
python tools/parity_sweep.py > gpurun_out/r06/parity_sweep.txt 2>&1; echo "parity sweep rc=$?" | tee -a gpurun_out/r06/summary5.txt
cat gpurun_out/r06/parity_sweep.txt | tail -14
python tools/bench_rules.py > gpurun_out/r06/rules_table_final.txt 2> gpurun_out/r06/rules_table_final.err; echo "rules rc=$?" | tee -a gpurun_out/r06/summary5.txt
cat gpurun_out/r06/rules_table_final.txt
python bench.py > gpurun_out/r06/bench5.json 2> gpurun_out/r06/bench5.err; echo "bench rc=$?" | tee -a gpurun_out/r06/summary5.txt
python bench.py --gpus 8 --share-device --no-cpu-baseline --no-realistic --no-config3 --no-live-traffic > gpurun_out/r06/share_device.json 2> gpurun_out/r06/share_device.err; echo "share-device rc=$?" | tee -a gpurun_out/r06/summary5.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/bench5.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "iter_us", d["roofline"]["iteration_us"], "traffic_frac", d["roofline"]["traffic_frac_of_peak"])
print("realistic", d["realistic"]["codewords_per_s"], d["realistic"]["fraction_of_iteration_proportional_bound"], "config3", d["config3"]["value"], d["config3"]["whole_job_frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["matches_gpu_output"])
s=json.load(open("gpurun_out/r06/share_device.json"))
print("share-device", s["n_gpus"], s["value"], [round(r["ms_per_step"],1) for r in s["launch"]["per_rank"]], s["launch"]["slowest_rank"])
PY
