#!/bin/bash
# Round 5, second pass (batched-load stencils, one-launch BatchNorm, mask-embedding backward, imd_prep planes, dilation rows): the parity tests of the
# touched kernels, then a same-lease A/B of the step against the library built from the sources before the change
# (maggie_amd/_variants/lib_old.so, MAGGIE_LIB_PATH) and a kernel trace of the new build. usage (through gpurun): bash tools/r05b_check.sh
out=gpurun_out/r05b
mkdir -p $out
K="token_self_attention or spatial_mean or bn_fold_and_pool or upsample_tanh or plane_flags or three_scales or gather_scatter or gather_tables or video_region or matting_losses or batch_norm_act_one_call or bn_train_forward_backward or reformed_from_the_raw_input or mask_embed or compute_unknown or active_pyramid or os8_weight or atten_guidance or bits_ or bn_fold"
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "$K" 2>&1 | tail -15 > $out/pytest_kernels.txt
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py -m gpu -x -q -p no:cacheprovider -k "train_step_matches or eval_forward_matches or train_step_is_bit_reproducible or former_atomic or video_consecutive" 2>&1 | tail -15 > $out/pytest_model.txt
grep -h "passed\|failed\|error" $out/pytest_kernels.txt $out/pytest_model.txt
B="--steps 60 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2; do
  for lib in new old; do
    if [ $lib = old ]; then export MAGGIE_LIB_PATH=maggie_amd/_variants/lib_old.so; else unset MAGGIE_LIB_PATH; fi
    timeout 300 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('$lib', r['value'], r['ms_per_step'])"
  done
done | tee $out/ab_old_new.txt
unset MAGGIE_LIB_PATH
# kernel trace of the new build
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_stats -- python $root/bench.py --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $root/$out/bench_traced.json 2> /tmp/pr_stats.err
tr=$(find /tmp/pr_stats -name '*kernel_trace.csv' | head -1)
ms=$(python -c "import json,sys; print(json.loads([l for l in open(sys.argv[1]) if l.startswith('{\"metric\"')][-1])['ms_per_step'])" $root/$out/bench_traced.json)
python $root/tools/trace_summary.py $tr 10 $ms 90 --torch > $root/$out/trace_summary.txt
head -3 $root/$out/trace_summary.txt
grep -n "pyr_\|point_bwd\|bn_small\|mask_embed\|imd_prep\|dilate" $root/$out/trace_summary.txt
