python tools/parity_stats.py 2>&1 | grep -v amdgpu.ids
echo "== groups"; python tools/perf_probe.py --groups 128,64 --reps 1 2>&1 | grep group
echo "== unroll 4"; LDPC_TOOLBOX_UNROLL=4 python tools/perf_probe.py --groups 4096 --reps 1 2>&1 | grep group
echo "== unroll vn 4"; LDPC_TOOLBOX_UNROLL_VN=4 python tools/perf_probe.py --groups 4096 --reps 1 2>&1 | grep group
echo "== vec 2"; LDPC_TOOLBOX_VEC=2 python tools/perf_probe.py --groups 4096 --reps 1 2>&1 | grep group
echo "== vec 1"; LDPC_TOOLBOX_VEC=1 python tools/perf_probe.py --groups 4096 --reps 1 2>&1 | grep group
for w in 2048 4096 16384 32768 131072; do echo "== waves $w"; LDPC_TOOLBOX_WAVES=$w python tools/perf_probe.py --groups 4096 --reps 1 2>&1 | grep group; done
