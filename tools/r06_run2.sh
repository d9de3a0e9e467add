#!/bin/bash
# round 6, second GPU call: the full GPU suite on the split library, the launcher tests printed
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
python -m pytest tests/test_gpu_launcher.py -x -q -m gpu -s > gpurun_out/r06/t_launcher2.log 2>&1; echo "launcher rc=$?" | tee -a gpurun_out/r06/summary2.txt
tail -25 gpurun_out/r06/t_launcher2.log | cut -c1-300
python -m pytest tests -q -m gpu > gpurun_out/r06/t_all2.log 2>&1; echo "full gpu suite rc=$?" | tee -a gpurun_out/r06/summary2.txt
tail -8 gpurun_out/r06/t_all2.log | cut -c1-300
