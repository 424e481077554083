#!/bin/bash
# long graph-served runs on the final tree (prints every 25 steps)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/soak_r04c; mkdir -p $O
line() { python3 -c "
import json
try:
    d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', d['value'], 'img/s', d['ms_per_step_median'], 'ms median', d['ms_per_step_max'], 'max; loss', d['loss_mean_timed_steps'], 'replayed', d['config']['step_graphs']['timed_steps_replayed'])
except Exception as e: print('$2: no line', e)
"; }
timeout -k 10 400 python bench.py --steps 2000 --warmup 8 --no_cpu_baseline --print_freq 25 > $O/a.json 2> $O/a.err; line $O/a.json "configs[1] 2000 steps"
timeout -k 10 400 python bench.py --model vit_small_patch16_224 --head None --num_heads 8 --learning_rate 0.005 --steps 1200 --warmup 5 --no_cpu_baseline --print_freq 25 > $O/b.json 2> $O/b.err; line $O/b.json "configs[2] 1200 steps"
grep -il "nan" $O/*.err; echo done
