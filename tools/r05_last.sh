#!/bin/bash
# the final binary once more: the default line without the CPU leg, and one kernel trace
out=gpurun_out/r05i; mkdir -p $out
timeout 100 python bench.py --no-cpu-baseline > $out/bench_default_final.json 2>/dev/null
tail -1 $out/bench_default_final.json | cut -c1-260
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_stats
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- python $root/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $root/$out/bench_default_traced.json 2> /tmp/pr_stats.err
f=$(find /tmp/pr_stats -name '*kernel_stats.csv' | head -1)
cp $f $root/$out/bench_default_kernel_stats.csv
python $root/tools/kstats.py $f 60 > $root/$out/bench_default_kernel_stats_top.txt
tr=$(find /tmp/pr_stats -name '*kernel_trace.csv' | head -1)
ms=$(python -c "import json,sys; print(json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric\"')][-1])['ms_per_step'])" $root/$out/bench_default_traced.json)
python $root/tools/trace_summary.py $tr 10 $ms 90 --torch > $root/$out/trace_summary.txt
head -1 $root/$out/trace_summary.txt
