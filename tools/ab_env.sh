#!/bin/bash
# A/B of an environment switch on ONE box. usage: tools/ab_env.sh VAR=a VAR=b [bench args]
A="$1"; B="$2"; shift 2
for i in 1 2 3; do
  for t in "$A" "$B"; do
    env $t timeout 200 python bench.py --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('$t', r['value'], r['ms_per_step'])"
  done
done
