R=$PWD
for rep in 1 2 3; do for tag in "$@"; do
  export LDPC_TOOLBOX_LIB=$R/ldpc_toolbox_amd/lib/libldpc_toolbox_$tag.so
  echo -n "== $tag  config3: "
  timeout 300 python3 tools/bench_config3.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']), round(j['realistic']['codewords_per_s']), end='')"
  echo -n "   headline: "
  timeout 300 python3 bench.py --steps 3 --warmup 1 --no-live-traffic --no-cpu-baseline --no-config3 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(round(j['value']), round(j['realistic']['codewords_per_s']))"
done; done
