#!/bin/bash
# builds (here) or runs (GPU box) the variants of tools/mb/slice_bench:  slice_ab.sh build | run [args of slice_bench after the graph]
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/tools/mb
if [ "$1" = build ]; then
  for v in ${VARIANTS:-0 1 2 3 4 5 7}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -DSLICE_EXP=$v $EXTRA slice_bench.hip -o slice_bench_$v || exit 1
  done
else
  shift
  for v in ${VARIANTS:-0 1 2 3 4 5 7}; do echo -n "exp $v: "; timeout 120 ./slice_bench_$v graph_bg1_384.bin "$@"; done
fi
