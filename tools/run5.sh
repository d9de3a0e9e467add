python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python tools/ab_tune.py --rounds 5 --configs "waves=8192;waves=65536;waves=131072;waves=262144;waves=524288;waves=262144,unroll_vn=4;waves=262144,unroll_cn=4;waves=262144,block=128;waves=262144,block=64;waves=262144,vec=2;waves=262144,group_size=2048;waves=262144,group_size=1024" 2>&1 | grep -v amdgpu
