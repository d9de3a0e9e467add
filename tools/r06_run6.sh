#!/bin/bash
# round 6, sixth GPU call: rocprofv3 evidence and the bench line from ONE box; the launcher tests on the final sweep job;
# all DVB-S2 rates; the host-buffer entry
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
bash tools/profile_r06.sh r06b > gpurun_out/r06/profile6.log 2>&1; echo "profile rc=$?" | tee -a gpurun_out/r06/summary6.txt
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r06/bench6.json 2> gpurun_out/r06/bench6.err; echo "bench rc=$?" | tee -a gpurun_out/r06/summary6.txt
head -16 gpurun_out/prof_r06b/summary/r06_rocprofv3_summary.txt | cut -c1-160
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/bench6.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], d["roofline"]["launch_us"], "traffic_frac", d["roofline"]["traffic_frac_of_peak"])
PY
python -m pytest tests/test_gpu_launcher.py -q -m gpu > gpurun_out/r06/t_launcher6.log 2>&1; echo "launcher rc=$?" | tee -a gpurun_out/r06/summary6.txt
tail -3 gpurun_out/r06/t_launcher6.log | cut -c1-200
echo "# flooding Minsumf32 on every DVB-S2 normal-frame rate, 4096 frames, 20 iterations at a noise level where no frame converges (tools/perf_probe.py), round 6 final build" > gpurun_out/r06/dvbs2_all_rates.txt
for r in R1_4 R1_3 R2_5 R1_2 R3_5 R2_3 R3_4 R4_5 R5_6 R8_9 R9_10; do
  echo -n "$r  " >> gpurun_out/r06/dvbs2_all_rates.txt
  python tools/perf_probe.py --spec dvbs2:$r --impl Minsumf32 --batch 4096 --iters 20 --groups 4096 --reps 2 --sigma 1.6 2>&1 | grep -E "group" | tail -1 >> gpurun_out/r06/dvbs2_all_rates.txt
done
cat gpurun_out/r06/dvbs2_all_rates.txt | cut -c1-220
python tools/host_path_probe.py dvbs2:R1_2 Minsumf32 16384 50 > gpurun_out/r06/host_path.txt 2>&1; cat gpurun_out/r06/host_path.txt | grep -v amdgpu
