#!/bin/bash
# usage: tools/trace_nb.sh <tag> [bench args]  -> gpurun_out/<tag>_neighbours.txt (+ the family summary)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mg_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/mg_tr -- python $root/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline "$@" > /tmp/mg_tr.log 2>&1
line=$(grep '^{"metric"' /tmp/mg_tr.log | tail -1)
ms=$(python -c "import json,sys; print(json.loads(sys.argv[1])['ms_per_step'])" "$line")
f=$(find /tmp/mg_tr -name '*kernel_trace.csv' | head -1)
head -1 $f > $root/gpurun_out/${tag}_csv_header.txt
python $root/tools/trace_summary.py $f 10 $ms 80 --torch > $root/gpurun_out/${tag}_trace_summary.txt
python $root/tools/trace_neighbours.py $f $ms > $root/gpurun_out/${tag}_neighbours.txt
echo "ms_per_step (traced) $ms"
python $root/tools/step_kernels.py $f $ms > $root/gpurun_out/${tag}_step_kernels.txt
