#!/bin/bash
# round 6, third GPU call (the trimmed library: 27 options, 447 kernels): suite, rules table, rocprof evidence, bench line,
# config 5 as one job on 1 rank and on 8 ranks sharing the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
python -m pytest tests -q -m gpu > gpurun_out/r06/t_all3.log 2>&1; echo "full gpu suite rc=$?" | tee -a gpurun_out/r06/summary3.txt
tail -4 gpurun_out/r06/t_all3.log | cut -c1-300
python tools/bench_rules.py > gpurun_out/r06/rules_table.txt 2> gpurun_out/r06/rules_table.err; echo "rules rc=$?" | tee -a gpurun_out/r06/summary3.txt
cat gpurun_out/r06/rules_table.txt
bash tools/profile_r06.sh r06 > gpurun_out/r06/profile.log 2>&1; echo "profile rc=$?" | tee -a gpurun_out/r06/summary3.txt
cd "$GRAFT_REPO_ROOT"
python bench.py > gpurun_out/r06/bench3.json 2> gpurun_out/r06/bench3.err; echo "bench rc=$?" | tee -a gpurun_out/r06/summary3.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06/bench3.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], "iter_us", d["roofline"]["iteration_us"], d["roofline"]["launch_us"], "traffic_frac", d["roofline"]["traffic_frac_of_peak"])
print("realistic", d.get("realistic",{}).get("codewords_per_s"), d.get("realistic",{}).get("fraction_of_iteration_proportional_bound"))
c3=d.get("config3",{}); print("config3", c3.get("value"), c3.get("whole_job_frac"), c3.get("realistic",{}).get("codewords_per_s"), c3.get("realistic",{}).get("fraction_of_iteration_proportional_bound"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("matches_gpu_output"))
PY
C5="--codes dvbs2:normal --grid waterfall --decoder Minsumf32 --max-iter 50 --frame-errors 100 --max-frames 1048576 --seed 7 --verbose"
( time python -m ldpc_toolbox_amd.ber $C5 --output-dir gpurun_out/r06/c5_one ) > gpurun_out/r06/c5_one.txt 2>&1; echo "config5 one rank rc=$?" | tee -a gpurun_out/r06/summary3.txt
tail -6 gpurun_out/r06/c5_one.txt
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 -m ldpc_toolbox_amd.ber $C5 --share-device --output-dir gpurun_out/r06/c5_eight ) > gpurun_out/r06/c5_eight.txt 2>&1; echo "config5 eight ranks (one GPU) rc=$?" | tee -a gpurun_out/r06/summary3.txt
tail -14 gpurun_out/r06/c5_eight.txt
python - <<'PY'
import os
def table(path):
    rows=[]
    for ln in open(path).read().splitlines():
        c=[x.strip() for x in ln.split("|")]
        if len(c)==11 and c[0].replace(".","").replace("-","").isdigit(): rows.append(c[:9])
    return rows
a="gpurun_out/r06/c5_one"; b="gpurun_out/r06/c5_eight"
same=True
for f in sorted(os.listdir(a)):
    ta, tb = table(os.path.join(a,f)), table(os.path.join(b,f)) if os.path.exists(os.path.join(b,f)) else None
    ok = ta==tb and len(ta)==8
    same = same and ok
    print(f, "rows", len(ta), "identical counter columns" if ok else "DIFFERENT")
print("config 5: one rank vs eight ranks:", "IDENTICAL" if same else "DIFFERENT")
PY
