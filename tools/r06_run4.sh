#!/bin/bash
# round 6, fourth GPU call: fused f64 phi (A/B against round 5's library on one box), round 5 / round 6 rules table on one box,
# the i8 fold forms, decoder-level straggler pooling, config 5 on eight ranks sharing the GPU against the one-rank tables
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
R5=$PWD/ldpc_toolbox_amd/lib/libldpc_toolbox_r05.so
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pooling_inside or transcendental or rows_beyond or Phi" > gpurun_out/r06/t_new4.log 2>&1; echo "new tests rc=$?" | tee -a gpurun_out/r06/summary4.txt
tail -5 gpurun_out/r06/t_new4.log | cut -c1-400
./tools/mb/i8_fold_bench --d 7 > gpurun_out/r06/i8_fold.txt 2>&1; ./tools/mb/i8_fold_bench --d 19 >> gpurun_out/r06/i8_fold.txt 2>&1; cat gpurun_out/r06/i8_fold.txt
for rep in 1 2; do
  echo "== round 5 library"; LDPC_TOOLBOX_LIB=$R5 BENCH_RULES_NO_CPU=1 python tools/bench_rules.py
  echo "== round 6 library"; BENCH_RULES_NO_CPU=1 python tools/bench_rules.py
done > gpurun_out/r06/rules_ab.txt 2>&1
cat gpurun_out/r06/rules_ab.txt
python tools/pooling_probe_decoder.py > gpurun_out/r06/pooling_decoder.txt 2>&1; cat gpurun_out/r06/pooling_decoder.txt
C5="--codes dvbs2:normal --grid waterfall --decoder Minsumf32 --max-iter 50 --frame-errors 100 --max-frames 1048576 --seed 7 --verbose"
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 -m ldpc_toolbox_amd.ber $C5 --share-device --output-dir gpurun_out/r06/c5_eight ) > gpurun_out/r06/c5_eight.txt 2>&1; echo "config5 eight ranks (one GPU) rc=$?" | tee -a gpurun_out/r06/summary4.txt
grep -v "^# rank [0-9]: dvbs2\|Gloo\|amdgpu.ids\|^\[W\|^$" gpurun_out/r06/c5_eight.txt | tail -16 | cut -c1-250
python - <<'PY'
import os
def table(path):
    rows=[]
    for ln in open(path).read().splitlines():
        c=[x.strip() for x in ln.split("|")]
        if len(c)==11 and c[0].replace(".","").replace("-","").isdigit(): rows.append(c[:9])
    return rows
a="profiles/r06_config5"; b="gpurun_out/r06/c5_eight"
same=True
for f in sorted(os.listdir(a)):
    ta = table(os.path.join(a,f)); tb = table(os.path.join(b,f)) if os.path.exists(os.path.join(b,f)) else None
    ok = ta==tb and len(ta)==8
    same = same and ok
    print(f, "rows", len(ta), "identical counter columns" if ok else "DIFFERENT")
print("config 5: one rank (profiles/r06_config5) vs eight ranks sharing the GPU:", "IDENTICAL" if same else "DIFFERENT")
PY
python -m pytest tests -q -m gpu > gpurun_out/r06/t_all4.log 2>&1; echo "full gpu suite rc=$?" | tee -a gpurun_out/r06/summary4.txt
tail -4 gpurun_out/r06/t_all4.log | cut -c1-300
