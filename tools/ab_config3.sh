#!/bin/bash
# A/B of decoder options on BASELINE config 3 (tools/bench_config3.py): tools/ab_config3.sh "pairs=1" "pairs=0" ...
# IMPL=HLTanhf32 by default.  Prints codewords/s (fixed work), codewords/s (2 dB, early termination), level launch us.
for v in "$@"; do
  args=""
  for kv in $v; do args="$args --set $kv"; done
  python3 tools/bench_config3.py --no-cpu-baseline --impl "${IMPL:-HLTanhf32}" $args 2>&1 | tail -1 > /tmp/ab_c3.json
  python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_c3.json").read())
print(f"{sys.argv[1]:24s} fixed {d['value']:8.0f} cw/s   realistic {d['realistic']['codewords_per_s']:8.0f} cw/s   level launch {d['roofline']['avg_launch_us']:.1f} us   whole-job frac {d['whole_job_frac']:.3f}")
PY
done
