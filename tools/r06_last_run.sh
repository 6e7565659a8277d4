mkdir -p gpurun_out/r06b
python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 1500 > gpurun_out/r06b/pytest_full.txt 2>&1; grep -v '^  File' gpurun_out/r06b/pytest_full.txt | tail -60 > gpurun_out/r06b/pytest_x.txt; grep -n "passed\|failed" gpurun_out/r06b/pytest_x.txt
python bench.py --cpu-baseline-full > gpurun_out/r06b/bench_default.json 2> gpurun_out/r06b/bench_default.err
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r06b/bench_default.json') if l.startswith('{')][-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['graph_replay'])"
MAGGIE_DIST_BACKEND=gloo MAGGIE_ONE_GPU=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r06b/bench_2ranks_one_gpu.json 2>/dev/null; tail -c 400 gpurun_out/r06b/bench_2ranks_one_gpu.json
(echo "=== shape 4 32 32 512 3 : conv_halo3_slab_kernel"; bash tools/pmc_one.sh conv_halo3_slab 4 32 32 512 3; echo "=== same shape, per-tile single-role form (MG_H3_SLAB_MIN=100000000)"; MG_H3_SLAB_MIN=100000000 bash tools/pmc_one.sh conv_halo3_kernel 4 32 32 512 3) > gpurun_out/r06b/pmc_slab.txt 2>&1; tail -30 gpurun_out/r06b/pmc_slab.txt
