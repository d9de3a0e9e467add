set -x
python -m pytest tests -m gpu -q 2>&1 | tail -15
python bench.py 2>&1 | grep -v amdgpu.ids | tail -3
for w in 131072 262144 524288; do for b in 64 256; do echo "== waves $w block $b"; LDPC_TOOLBOX_BLOCK=$b LDPC_TOOLBOX_WAVES=$w python tools/perf_probe.py --groups 4096 --reps 1 2>&1 | grep group; done; done
echo "== waves 131072 unroll vn 4"; LDPC_TOOLBOX_UNROLL_VN=4 LDPC_TOOLBOX_WAVES=131072 python tools/perf_probe.py --groups 4096 --reps 1 2>&1 | grep group
echo "== 5G HLTanhf32"; python tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 8192 --iters 10 --groups 8192,2048 --reps 1 --sigma 1.8 2>&1 | grep group
echo "== 5G HLMinsumf32"; python tools/perf_probe.py --spec nr5g:1:384 --impl HLMinsumf32 --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.8 2>&1 | grep group
echo "== DVB Tanhf32"; python tools/perf_probe.py --impl Tanhf32 --batch 4096 --iters 10 --groups 4096 --reps 1 2>&1 | grep group
