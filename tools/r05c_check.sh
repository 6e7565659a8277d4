#!/bin/bash
# token-linear forward with its epilogue operands requested up front: its parity tests, the train fixtures, a same-lease A/B against the library before the batched-load pass
out=gpurun_out/r05c
mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "copy_k or token_linear" 2>&1 | tail -4 > $out/pytest_kernels.txt
timeout 300 python -m pytest tests/test_gpu_conv.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 > $out/pytest_conv.txt
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_determinism.py tests/test_gpu_graphs.py -m gpu -x -q -p no:cacheprovider -k "train_step_matches or train_step_is_bit_reproducible or parked_slab or outputs_of_a_replayed or gradient_sink" 2>&1 | tail -4 > $out/pytest_model.txt
grep -h "passed\|failed\|error" $out/pytest_kernels.txt $out/pytest_conv.txt $out/pytest_model.txt
B="--steps 60 --warmup 10 --no-cpu-baseline --no-roofline"
for i in 1 2; do
  for lib in new prev; do
    if [ $lib = prev ]; then export MAGGIE_LIB_PATH=maggie_amd/_variants/lib_prev.so; else unset MAGGIE_LIB_PATH; fi
    timeout 200 python bench.py $B 2>/dev/null | tail -1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print('$lib', r['value'], r['ms_per_step'])"
  done
done | tee $out/ab_prev_new.txt
