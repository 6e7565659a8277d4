#!/bin/bash
# HBM-side traffic of the final build: FETCH_SIZE and WRITE_SIZE in separate passes over eager launches (tools/profile_round.sh, step 2)
root=$(pwd); out=$root/gpurun_out/r05h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
EAGER="env MAGGIE_HIP_GRAPHS=0 python $root/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pr_$c
  timeout 105 rocprofv3 --pmc $c --output-format csv -d /tmp/pr_$c -- $EAGER > /tmp/pr_$c.log 2>&1
done
ff=$(find /tmp/pr_FETCH_SIZE -name '*counter_collection.csv' | head -1)
wf=$(find /tmp/pr_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python $root/tools/pmc_traffic.py $ff $wf 4 $out/pmc_traffic.json > $out/pmc_traffic.txt
tail -3 /tmp/pr_FETCH_SIZE.log; head -5 $out/pmc_traffic.txt
