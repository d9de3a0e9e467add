python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python tools/ab_tune.py --rounds 5 --iters 10 --configs "tile=256;tile=256,nt=1;tile=512;tile=512,nt=1;tile=1024;tile=1024,nt=1;tile=2048,nt=1;tile=4096,nt=1;tile=512,vec=2,nt=1;tile=256,vec=2,nt=1" 2>&1 | grep -v amdgpu
