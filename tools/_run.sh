for i in 1 2; do
for lib in libldpc_toolbox_prev.so libldpc_toolbox.so; do echo "== $lib"; LDPC_TOOLBOX_LIB=$PWD/ldpc_toolbox_amd/lib/$lib python tools/perf_probe.py --impl Minsumf32 --batch 4096 --iters 50 --groups 4096 --reps 2 2>&1 | grep group | cut -c60-220; done; done
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/scalar_probe.py 2>&1 | grep -v amdgpu
