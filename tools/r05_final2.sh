#!/bin/bash
# Final GPU call of round 5 (second pass): full GPU suite, the driver's default bench command, one kernel trace. usage (through gpurun): bash tools/r05_final2.sh
out=gpurun_out/r05f
mkdir -p $out
timeout 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 1500 2>&1 | tail -12 > $out/pytest_full.txt
grep -h "passed\|failed\|error" $out/pytest_full.txt
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -1 $out/bench_default.json | cut -c1-400
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- python $root/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $root/$out/bench_default_traced.json 2> /tmp/pr_stats.err
f=$(find /tmp/pr_stats -name '*kernel_stats.csv' | head -1)
cp $f $root/$out/bench_default_kernel_stats.csv
python $root/tools/kstats.py $f 60 > $root/$out/bench_default_kernel_stats_top.txt
tr=$(find /tmp/pr_stats -name '*kernel_trace.csv' | head -1)
ms=$(python -c "import json,sys; print(json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric\"')][-1])['ms_per_step'])" $root/$out/bench_default_traced.json)
python $root/tools/trace_summary.py $tr 10 $ms 90 --torch > $root/$out/trace_summary.txt
head -2 $root/$out/trace_summary.txt
cd $root
timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --video > $out/bench_video.json 2>/dev/null; tail -1 $out/bench_video.json | cut -c1-200
