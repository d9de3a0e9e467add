python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python tools/ab_tune.py --rounds 5 --iters 10 --configs "tile=256;tile=256,unroll_vn=4" 2>&1 | grep -v amdgpu
echo "== fake sequential gather (timing only)"
LDPC_TOOLBOX_FAKE_SEQ=1 python tools/ab_tune.py --rounds 5 --iters 10 --configs "tile=256;tile=256,unroll_vn=4" 2>&1 | grep -v amdgpu
