#!/bin/bash
# A/B of the default bench under env settings: tools/ab.sh "<envA>" "<envB>" [runs] ; prints ms_per_step of each run
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  for e in "$a" "$b"; do
    r=$(env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "[$e] $r"
  done
done
