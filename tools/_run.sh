python -m pytest tests -m gpu -q -x -k "i8 or stress or golden" 2>&1 | tail -3
for impl in Minstarapproxi8 Aminstari8JonesPartialHardLimitDeg1Clip; do python tools/perf_probe.py --impl $impl --batch 4096 --iters 10 --groups 4096 --reps 2 2>&1 | grep group | cut -c1-200; done
for impl in HLMinstarapproxi8 HLAminstari8; do python tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.8 2>&1 | grep group | cut -c1-200; done
