python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/ab_tune.py --rounds 5 --iters 10 --configs "tile=256;tile=256,nt_vn=1;tile=256,unroll_vn=4;tile=256,unroll_cn=4;tile=256,unroll_cn=4,unroll_vn=4;tile=512;tile=256,waves=131072;tile=256,waves=524288;tile=256,block=128" 2>&1 | grep -v amdgpu
