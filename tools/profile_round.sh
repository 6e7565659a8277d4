#!/bin/bash
# All per-round profile artefacts in one GPU call; results land in gpurun_out/<tag>_* (copy the summaries into profiles/).
# usage: tools/profile_round.sh r02
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
BENCH="python $root/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline"

# 1. kernel trace + stats of the default command (graph replay)
rm -rf /tmp/pr_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- $BENCH > $out/${tag}_bench_default_traced.json 2> /tmp/pr_stats.err
f=$(find /tmp/pr_stats -name '*kernel_stats.csv' | head -1)
cp $f $out/${tag}_bench_default_kernel_stats.csv
python $root/tools/kstats.py $f 60 > $out/${tag}_bench_default_kernel_stats_top.txt
tr=$(find /tmp/pr_stats -name '*kernel_trace.csv' | head -1)
ms=$(python -c "import json,sys; print(json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric\"')][-1])['ms_per_step'])" $out/${tag}_bench_default_traced.json)
python $root/tools/trace_summary.py $tr 10 $ms 80 --torch > $out/${tag}_trace_summary.txt

# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, eager launches (4 steps)
EAGER="env MAGGIE_HIP_GRAPHS=0 python $root/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pr_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pr_$c -- $EAGER > /tmp/pr_$c.log 2>&1
done
ff=$(find /tmp/pr_FETCH_SIZE -name '*counter_collection.csv' | head -1)
wf=$(find /tmp/pr_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python $root/tools/pmc_traffic.py $ff $wf 4 $out/${tag}_pmc_traffic.json > $out/${tag}_pmc_traffic.txt

# 3. MFMA utilisation
rm -rf /tmp/pr_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pr_mfma -- $EAGER > /tmp/pr_mfma.log 2>&1
mf=$(find /tmp/pr_mfma -name '*counter_collection.csv' | head -1)
python $root/tools/pmc_mfma.py $mf > $out/${tag}_pmc_mfma.txt
ls -la $out/${tag}_*
