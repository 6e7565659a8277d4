#!/bin/bash
# one bench run per env setting: tools/sweep.sh "A=1" "A=2 B=3" ...
for e in "$@"; do
  r=$(env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "[$e] $r"
done
