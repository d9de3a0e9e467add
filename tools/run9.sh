export LDPC_TOOLBOX_DEBUG=1
python tools/ab_tune.py --verbose --rounds 14 --iters 6 --configs "alloc_mode=1;alloc_mode=1,group_size=2048;alloc_mode=1,group_size=1024" 2>&1 | grep -v amdgpu | grep -v "G=2048\|G=1024\|group_size" | cut -c1-200
