python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/ab_tune.py --rounds 5 --iters 10 --configs "lfree=0;lfree=1;lfree=1,waves_vn=65536;lfree=1,waves_vn=262144;lfree=1,unroll_vn=4" 2>&1 | grep -v amdgpu
