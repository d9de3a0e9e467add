python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py 2>&1 | grep -v amdgpu | tail -1
bash tools/profile_r01.sh 2>&1 | tail -32
