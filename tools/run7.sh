export LDPC_TOOLBOX_DEBUG=1
echo "== slab, realloc each time via group toggle"
python tools/ab_tune.py --rounds 8 --iters 10 --configs "waves=262144;waves=262144,group_size=2048" 2>&1 | grep -v amdgpu | grep -v "G=2048" | tail -12
echo "== pads"
python tools/ab_tune.py --rounds 3 --iters 10 --configs "pad_kb=0;pad_kb=4;pad_kb=16;pad_kb=64;pad_kb=256;pad_kb=516;pad_kb=1028;pad_kb=1040" 2>&1 | grep -v amdgpu | grep -v "workspace"
