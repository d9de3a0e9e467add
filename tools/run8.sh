export LDPC_TOOLBOX_DEBUG=1
for mode in 0 1 2 0 1 2 1 1; do
echo "== fresh process alloc_mode $mode"
python tools/ab_tune.py --rounds 3 --iters 10 --configs "alloc_mode=$mode" 2>&1 | grep -v amdgpu | tail -2 | cut -c1-250
done
