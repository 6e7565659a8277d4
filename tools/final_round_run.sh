mkdir -p gpurun_out/r04r
python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 1500 2>&1 | tail -40 > gpurun_out/r04r/pytest_x_9.txt
python bench.py > gpurun_out/r04r/bench_default.json 2> gpurun_out/r04r/bench_default.err
B="--steps 60 --warmup 10 --no-cpu-baseline"
python bench.py $B --video > gpurun_out/r04r/bench_video.json 2>/dev/null
python bench.py $B --instances 4 > gpurun_out/r04r/bench_4inst.json 2>/dev/null
python bench.py $B --dtype fp16 > gpurun_out/r04r/bench_fp16.json 2>/dev/null
python bench.py $B --batch 12 > gpurun_out/r04r/bench_batch12.json 2>/dev/null
MAGGIE_FORCE_DDP=1 python bench.py $B > gpurun_out/r04r/bench_force_ddp.json 2>/dev/null
MAGGIE_SYNCBN_WORLD1=1 MAGGIE_FORCE_DDP=1 python bench.py $B --sync-bn > gpurun_out/r04r/bench_syncbn_default.json 2>/dev/null
MAGGIE_DETERMINISTIC=0 python bench.py $B > gpurun_out/r04r/bench_nondet.json 2>/dev/null
bash tools/profile_round.sh r04r > gpurun_out/r04r/profile.log 2>&1
grep -n "passed\|failed" gpurun_out/r04r/pytest_x_9.txt
for f in default video 4inst fp16 batch12 force_ddp syncbn_default nondet; do python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r04r/bench_$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
