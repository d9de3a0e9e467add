python -m pytest tests -m gpu -q -x -k "sweep" 2>&1 | tail -3
python -m ldpc_toolbox_amd.ber --code dvbs2:R1_2 --decoder Minsumf32 --min-ebn0 1.2 --max-ebn0 2.0 --step-ebn0 0.2 --max-iter 50 --frame-errors 100 --max-frames 65536 --seed 1 2>&1 | grep -v amdgpu
