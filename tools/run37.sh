python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for impl in Minstarapproxf32 Tanhf32 Minstarapproxi8; do python tools/perf_probe.py --impl $impl --batch 4096 --iters 10 --groups 4096 --reps 1 2>&1 | grep group | cut -c1-200; done
for impl in HLMinstarapproxf32 HLTanhf32 HLMinstarapproxi8; do python tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --iters 10 --groups 8192 --reps 1 --sigma 1.8 2>&1 | grep group | cut -c1-200; done
