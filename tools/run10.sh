python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python tools/ab_tune.py --rounds 5 --iters 10 --configs "tile=4096;tile=1024;tile=512;tile=256;tile=128;tile=256,waves=65536;tile=256,waves=1048576;tile=256,unroll_vn=4;tile=512,waves=1048576" 2>&1 | grep -v amdgpu
