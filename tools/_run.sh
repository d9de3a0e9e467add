python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for impl in HLMinsumf32 HLTanhf32 HLMinstarapproxi8 HLAminstarf32; do for set in lanes=1 lanes=2; do echo "== $impl $set"; python tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.8 --set $set 2>&1 | grep group | cut -c1-200; done; done
for set in lanes=1 lanes=2; do echo "== ar4ja HLMinsumf32 $set"; python tools/perf_probe.py --spec ar4ja:1/2:1024 --impl HLMinsumf32 --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.8 --set $set 2>&1 | grep group | cut -c1-200; done
python tools/host_path_probe.py dvbs2:R1_2 Minsumf32 8192 50 2>&1 | grep path
python tools/host_path_probe.py nr5g:1:384 HLMinsumf32 16384 10 2>&1 | grep path
