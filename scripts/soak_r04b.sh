#!/bin/bash
# Round-4 soak after the graph-replay hazard fix: long graph-served runs, prints (eager kernels + read-back) every 10 steps.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/soak_r04b; mkdir -p $O
line() { python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', d['value'], 'img/s', d['ms_per_step_median'], 'ms median', d['ms_per_step_max'], 'max; loss', d['loss_mean_timed_steps'], 'replayed', d['config']['step_graphs']['timed_steps_replayed'], 'host min', d['host_issue_ms_min'])
except Exception as e: print('$2: no line', e)
"; }
timeout -k 10 300 python bench.py --steps 600 --warmup 8 --no_cpu_baseline --print_freq 10 > $O/a.json 2> $O/a.err; line $O/a.json "configs[1] 600 steps"
timeout -k 10 300 python bench.py --model vit_small_patch16_224 --head None --num_heads 8 --learning_rate 0.005 --steps 300 --warmup 5 --no_cpu_baseline --print_freq 10 > $O/b.json 2> $O/b.err; line $O/b.json "configs[2] 300 steps"
timeout -k 10 300 python bench.py --batch_size 64 --steps 800 --warmup 8 --no_cpu_baseline --print_freq 10 > $O/c.json 2> $O/c.err; line $O/c.json "B=64 800 steps"
MOMA_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29672 bench.py --gpus 1 --steps 400 --warmup 8 --no_cpu_baseline --print_freq 10 > $O/d.json 2> $O/d.err; line $O/d.json "one RCCL rank 400 steps"
timeout -k 10 300 python bench.py --head None --steps 300 --warmup 8 --no_cpu_baseline --print_freq 10 > $O/e.json 2> $O/e.err; line $O/e.json "head None 300 steps"
grep -il "nan" $O/*.err; echo done
