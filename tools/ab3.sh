#!/bin/bash
# same-lease A/B/C...: tools/ab3.sh <runs> <steps> "<env1>" "<env2>" ...   (ms_per_step of each 'steps'-step run, interleaved)
n=$1; steps=$2; shift 2
for i in $(seq $n); do
  for e in "$@"; do
    r=$(env $e python bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "[$e] $r"
  done
done
