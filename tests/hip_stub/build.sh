#!/bin/bash
# builds tests/tsan_driver against the library's host code + the HIP stub under ThreadSanitizer:  build.sh <out-dir>
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$1; mkdir -p $OUT
CXX=/opt/rocm/lib/llvm/bin/clang++
HIPHOST="-x hip --cuda-host-only --offload-arch=gfx950 -Wno-option-ignored -I/opt/rocm/include"
FLAGS="-std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=thread -ffp-contract=off -fPIC"
C=$R/ldpc_toolbox_amd/csrc
HIPTU="device_decoder run_group_f32 run_group_f64 run_group_i8 latency_paths simulator"   # the library's HIP translation units (csrc/Makefile, HIP_SRC)
for f in $HIPTU; do $CXX $HIPHOST $FLAGS -c $C/$f.hip -o $OUT/$f.o & done
$CXX $HIPHOST $FLAGS -c $R/tests/hip_stub/hip_stub.hip -o $OUT/hip_stub.o &
for f in sparse codes encoder implementation c_api; do $CXX $FLAGS -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c $C/$f.cpp -o $OUT/$f.o & done
$CXX $FLAGS -c $R/tests/tsan_driver.cpp -o $OUT/tsan_driver.o &
wait
# the host objects of a host-only HIP compile refer to the (absent) device image by a hashed symbol: define them empty
syms=$(nm $(for f in $HIPTU; do echo $OUT/$f.o; done) $OUT/hip_stub.o | grep " U __hip_fatbin" | awk '{print $2}' | sort -u)
: > $OUT/fatbin.c
for s in $syms; do echo "const char $s[16] = {0};" >> $OUT/fatbin.c; done
gcc -c $OUT/fatbin.c -o $OUT/fatbin.o
$CXX -fsanitize=thread -pthread $OUT/*.o -o $OUT/tsan_driver
