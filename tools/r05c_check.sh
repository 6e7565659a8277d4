#!/bin/bash
# a change of one or two kernels: their parity tests, the train fixtures + determinism, a same-lease A/B against the library before it (MAGGIE_LIB_PATH), per-kernel stats
out=gpurun_out/r05c
mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "matting_losses or three_scales or os8_weight" 2>&1 | tail -4 > $out/pytest_kernels.txt

timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_graphs.py -m gpu -x -q -p no:cacheprovider -k "train_step_matches or train_step_is_bit_reproducible or graphed_step_matches" 2>&1 | tail -4 > $out/pytest_model.txt
grep -h "passed\|failed\|error" $out/pytest_kernels.txt $out/pytest_model.txt
B="--steps 60 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1; do
  for lib in new prev; do
    if [ $lib = prev ]; then export MAGGIE_LIB_PATH=maggie_amd/_variants/lib_prev.so; else unset MAGGIE_LIB_PATH; fi
    timeout 200 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('$lib', r['value'], r['ms_per_step'])"
  done
done | tee $out/ab_prev_new.txt
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_stats
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- python $root/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > /tmp/traced.json 2> /tmp/pr_stats.err
f=$(find /tmp/pr_stats -name '*kernel_stats.csv' | head -1)
grep -i "pyr_\|point_bwd" $f | awk -F, '{print $1, $(NF-6), $(NF-4), $(NF-2), $(NF-1)}' | cut -c1-160 > $root/$out/pyr_stats.txt
cat $root/$out/pyr_stats.txt
