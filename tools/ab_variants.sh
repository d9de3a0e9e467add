#!/bin/bash
# A/B builds of the library for kernel experiments (never the product): each variant is the whole library
# compiled with extra -D flags into ldpc_toolbox_amd/lib/libldpc_toolbox_<tag>.so; LDPC_TOOLBOX_LIB selects it.
#   tools/ab_variants.sh build <tag> [-Dflags...]     (here)
#   tools/ab_variants.sh run <tag> [<tag> ...]        (GPU box: config 3 + DVB-S2 Tanhf32 timing, VALU counters)
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  tag=$2; shift 2
  make -C $R/ldpc_toolbox_amd/csrc BUILD=build_$tag OUT=../lib/libldpc_toolbox_$tag.so EXTRA_HIPFLAGS="$*" ALLOW_HAZARD=1 2>&1 | grep -E "error|Error|store\(s\) of more" ; ls -la $R/ldpc_toolbox_amd/lib/libldpc_toolbox_$tag.so
else
  shift
  cd /tmp && export TMPDIR=/tmp
  OUT=$R/gpurun_out/ab; mkdir -p $OUT
  for rep in 1 2; do for tag in "$@"; do
    export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox_$tag.so
    echo "== $tag (rep $rep)"
    for impl in ${AB_IMPLS_HL:-HLTanhf32}; do echo -n "$impl  "
    python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl $impl --batch 8192 --iters 10 --groups 8192 --reps 2 --sigma 1.565 2>&1 | grep -E "group|Error" | cut -c1-150; done
    for impl in ${AB_IMPLS_FL:-Tanhf32}; do echo -n "$impl  "
    python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl $impl --batch 4096 --iters 10 --groups 4096 --reps 2 2>&1 | grep -E "group|Error" | cut -c1-170; done
  done; done
  [ -n "$AB_NO_PMC" ] && exit 0
  for tag in "$@"; do
    export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox_$tag.so
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_$tag -- python3 $R/tools/perf_probe.py --spec nr5g:1:384 --impl HLTanhf32 --batch 8192 --iters 4 --groups 8192 --reps 1 --sigma 1.565 > $OUT/pmc_$tag.log 2>&1
    echo "== $tag counters"; python3 $R/tools/parse_pmc.py $OUT/pmc_$tag --match hl_level | grep -v JSON
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $OUT/pmcf_$tag -- python3 $R/tools/perf_probe.py --spec dvbs2:R1_2 --impl Tanhf32 --batch 4096 --iters 4 --groups 4096 --reps 1 > $OUT/pmcf_$tag.log 2>&1
    python3 $R/tools/parse_pmc.py $OUT/pmcf_$tag --match cn_staged | grep -v JSON
  done
  rm -rf $OUT/pmc_*/ $OUT/pmcf_*/
fi
