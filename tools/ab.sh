#!/bin/bash
# A/B on ONE box: tree B = this snapshot, tree A = a copy with the files under .ab_old/ laid over it. usage: tools/ab.sh [bench args]
set -e
rm -rf /tmp/repoA && cp -r "$GRAFT_REPO_ROOT" /tmp/repoA && cp -r /tmp/repoA/.ab_old/. /tmp/repoA/
for i in 1 2 3; do
  for t in A B; do
    d=/tmp/repoA; [ $t = B ] && d="$GRAFT_REPO_ROOT"
    (cd $d && timeout 200 python bench.py --steps 30 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('$t', r['value'], r['ms_per_step'])")
  done
done
