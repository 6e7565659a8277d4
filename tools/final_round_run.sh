#!/bin/bash
# One GPU call at the end of a round: the full -x GPU suite, every bench line and the profile passes of the same build on one lease.
# usage (through gpurun): bash tools/final_round_run.sh r05  -> gpurun_out/r05/*, gpurun_out/r05_*
TAG=${1:-rXX}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 1500 > gpurun_out/$TAG/pytest_full.txt 2>&1; grep -v '^  File' gpurun_out/$TAG/pytest_full.txt | tail -60 > gpurun_out/$TAG/pytest_x.txt
python bench.py --cpu-baseline-full > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
B="--steps 60 --warmup 10 --no-cpu-baseline"
python bench.py $B --video > gpurun_out/$TAG/bench_video.json 2>/dev/null
python bench.py $B --instances 4 > gpurun_out/$TAG/bench_4inst.json 2>/dev/null
python bench.py $B --dtype fp16 > gpurun_out/$TAG/bench_fp16.json 2>/dev/null
MAGGIE_MEM_FRACTION=0.92 python bench.py $B --batch 12 > gpurun_out/$TAG/bench_batch12_full.json 2>/dev/null                      # maggie_image.yaml:83
MAGGIE_MEM_FRACTION=0.92 python bench.py $B --video --frames 8 --clips 4 > gpurun_out/$TAG/bench_video_t8.json 2>/dev/null   # maggie_video.yaml:32,89
python bench.py $B --workload pred > gpurun_out/$TAG/bench_pred.json 2>/dev/null                                                   # iter = 10000
MAGGIE_LAZY_BN=0 python bench.py $B > gpurun_out/$TAG/bench_stored_bn.json 2>/dev/null                                              # the stored BatchNorm form (A/B of the operand path)
MAGGIE_FORCE_DDP=1 python bench.py $B > gpurun_out/$TAG/bench_force_ddp.json 2>/dev/null
MAGGIE_SYNCBN_WORLD1=1 MAGGIE_FORCE_DDP=1 python bench.py $B --sync-bn > gpurun_out/$TAG/bench_syncbn_default.json 2>/dev/null
MAGGIE_DETERMINISTIC=0 python bench.py $B > gpurun_out/$TAG/bench_nondet.json 2>/dev/null
bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile.log 2>&1
# per-layer conv tables (eager HIP events per entry-point call: they come out of the roofline pass, so no --no-roofline here) at the three shapes the judge's targets name, and the kernel-level check + timing log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --layers > /dev/null 2> gpurun_out/${TAG}_conv_layers_b4.txt
MAGGIE_MEM_FRACTION=0.92 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --layers --batch 12 > /dev/null 2> gpurun_out/${TAG}_conv_layers_b12.txt
MAGGIE_MEM_FRACTION=0.92 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --layers --video --frames 8 --clips 4 > /dev/null 2> gpurun_out/${TAG}_conv_layers_vt8.txt
(python tools/h3_check.py check | tail -3; H3_CHECK_CFG=8,32,201 python tools/h3_check.py check | tail -2; python tools/h3_check.py time 8,32,1 8,32,201) > gpurun_out/${TAG}_h3_check.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$TAG/smoke.txt 2>&1; tail -1 gpurun_out/$TAG/smoke.txt
grep -n "passed\|failed" gpurun_out/$TAG/pytest_x.txt
for f in default video 4inst fp16 batch12_full video_t8 pred stored_bn force_ddp syncbn_default nondet; do python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/$TAG/bench_$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
