python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for s in 0.7943 0.8318; do echo "== sigma $s"; python tools/ab_tune.py --rounds 3 --iters 50 --sigma $s --configs "compact=0;compact=1" 2>&1 | grep -v amdgpu | cut -c1-120; done
echo "== P1"; python tools/ab_tune.py --rounds 3 --iters 50 --sigma 1.0 --configs "compact=0;compact=1" 2>&1 | grep -v amdgpu | cut -c1-200
echo "== 5G HLMinsum 2dB-ish"; python tools/ab_tune.py --spec nr5g:1:384 --impl HLMinsumf32 --batch 8192 --rounds 2 --iters 30 --sigma 1.05 --configs "compact=0;compact=1" 2>&1 | grep -v amdgpu | cut -c1-120
