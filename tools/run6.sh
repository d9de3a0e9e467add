echo "== same config repeated (no realloc)"
python tools/ab_tune.py --rounds 10 --iters 10 --configs "waves=262144" 2>&1 | grep -v amdgpu
python tools/ab_tune.py --rounds 10 --iters 10 --configs "waves=262144;waves=262144,group_size=2048" 2>&1 | grep -v amdgpu
echo "== batch 3840 group 3840"
python tools/ab_tune.py --batch 3840 --rounds 6 --iters 10 --configs "waves=262144,group_size=3840;waves=262144,group_size=1280" 2>&1 | grep -v amdgpu
echo "== batch 4352 group 4352"
python tools/ab_tune.py --batch 4352 --rounds 6 --iters 10 --configs "waves=262144,group_size=4352;waves=262144,group_size=2304" 2>&1 | grep -v amdgpu
echo "== batch 8192 group 8192 / 4096"
python tools/ab_tune.py --batch 8192 --rounds 4 --iters 10 --configs "waves=262144,group_size=8192;waves=262144,group_size=4096" 2>&1 | grep -v amdgpu
