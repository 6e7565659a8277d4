#!/bin/bash
# the other bench lines of the final build, most important first (each writes its own file: a call cut short keeps what finished)
out=gpurun_out/r05g
mkdir -p $out
B="--steps 60 --warmup 10 --no-cpu-baseline"
MAGGIE_MEM_FRACTION=0.92 timeout 200 python bench.py $B --batch 12 > $out/bench_batch12_full.json 2>/dev/null
MAGGIE_MEM_FRACTION=0.92 MAGGIE_SPARSE_CAPACITY=0.4 timeout 200 python bench.py $B --video --frames 8 --clips 4 > $out/bench_video_t8.json 2>/dev/null
MAGGIE_SYNCBN_WORLD1=1 MAGGIE_FORCE_DDP=1 timeout 150 python bench.py $B --sync-bn > $out/bench_syncbn_default.json 2>/dev/null
timeout 150 python bench.py $B --workload pred > $out/bench_pred.json 2>/dev/null
timeout 150 python bench.py $B --instances 4 > $out/bench_4inst.json 2>/dev/null
timeout 150 python bench.py $B --dtype fp16 > $out/bench_fp16.json 2>/dev/null
MAGGIE_FORCE_DDP=1 timeout 150 python bench.py $B > $out/bench_force_ddp.json 2>/dev/null
for f in batch12_full video_t8 syncbn_default pred 4inst fp16 force_ddp; do python -c "
import json,sys
try:
    d=json.loads([l for l in open('$out/bench_$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))
except Exception as e: print('$f', 'missing')"; done
