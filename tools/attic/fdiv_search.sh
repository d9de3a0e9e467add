#!/bin/bash
# Which reciprocal-refinement sequence (exact_math.h, fdiv_v<V>) does each division site need?  Builds the
# exhaustive device check once per (site, variant) and runs it on the function that contains the site.
#   tools/fdiv_search.sh build     (here: hipcc cross-compiles)
#   tools/fdiv_search.sh run       (GPU box)
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/tools/mb/fdiv
declare -A FN=([EXPM1]=expm1f [TANH]=tanhf [L1P_C]=log1pf [L1P_S]=log1pf [ATANH]=atanh)
if [ "$1" = build ]; then
  mkdir -p $B
  for site in EXPM1 TANH L1P_C L1P_S ATANH; do for v in 1 2 3 4; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -pthread -DEM_FDIV_$site=$v \
      $R/tools/check_exact_math_device.hip -o $B/check_${site}_$v 2>/dev/null &
  done; wait; done
  # every site at variant 2 together (the sites interact inside tanhf: expm1f's quotient feeds tanhf's)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -pthread -DEM_FDIV_EXPM1=2 -DEM_FDIV_TANH=2 \
    -DEM_FDIV_L1P_C=2 -DEM_FDIV_L1P_S=2 -DEM_FDIV_ATANH=2 $R/tools/check_exact_math_device.hip -o $B/check_ALL_2 2>/dev/null
  ls $B
else
  for site in EXPM1 TANH L1P_C L1P_S ATANH; do for v in 4 2 3 1; do
    echo "== site $site variant $v"; $B/check_${site}_$v ${FN[$site]}
  done; done
  echo "== all sites at variant 2"; $B/check_ALL_2
fi
