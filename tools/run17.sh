python -m pytest tests -m gpu -q -x 2>&1 | tail -3
echo "== 5G HLTanhf32"; python tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.8 2>&1 | grep group
echo "== 5G HLMinsumf32"; python tools/perf_probe.py --spec nr5g:1:384 --impl HLMinsumf32 --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.8 2>&1 | grep group
echo "== 5G HLMinstarapproxf32"; python tools/perf_probe.py --spec nr5g:1:384 --impl HLMinstarapproxf32 --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.8 2>&1 | grep group
echo "== 5G HLAminstarf32"; python tools/perf_probe.py --spec nr5g:1:384 --impl HLAminstarf32 --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.8 2>&1 | grep group
echo "== DVB Tanhf32"; python tools/perf_probe.py --impl Tanhf32 --batch 4096 --iters 10 --groups 4096 --reps 1 2>&1 | grep group
echo "== DVB Phif32"; python tools/perf_probe.py --impl Phif32 --batch 4096 --iters 10 --groups 4096 --reps 1 2>&1 | grep group
echo "== DVB Aminstarf32"; python tools/perf_probe.py --impl Aminstarf32 --batch 4096 --iters 10 --groups 4096 --reps 1 2>&1 | grep group
