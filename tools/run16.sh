for s in 0.7943 0.8318 0.8610; do echo "== sigma $s"; python tools/perf_probe.py --groups 4096,1024 --reps 1 --sigma $s 2>&1 | grep group | cut -c1-220; done
