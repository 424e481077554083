#!/bin/bash
# long graph-served runs on the round-6 tree (restructured K2 waits, K3 inside the combine launch, unconditional wide-row requests)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/soak_r06; mkdir -p $O
line() { python3 -c "
import json
try:
    d=json.loads([l for l in open('$1') if l.startswith('{')][-1]); print('$2', d['value'], 'img/s', d['ms_per_step_median'], 'ms median', d['ms_per_step_max'], 'max; loss', d['loss_mean_timed_steps'], 'replayed', d['config']['step_graphs']['timed_steps_replayed'], 'host cpu ms', d['host_cpu_ms_median'])
except Exception as e: print('$2: no line', e)
"; }
timeout -k 10 400 python bench.py --steps 1500 --warmup 8 --no_cpu_baseline --print_freq 25 > $O/a.json 2> $O/a.err; line $O/a.json "configs[1] 1500 steps"
timeout -k 10 400 python bench.py --model vit_small_patch16_224 --head None --num_heads 8 --learning_rate 0.005 --steps 800 --warmup 5 --no_cpu_baseline --print_freq 25 > $O/b.json 2> $O/b.err; line $O/b.json "configs[2] 800 steps"
timeout -k 10 500 python bench.py --model ResNet50 --model_t vit_base_patch16_224 --image_size 512 --batch_size 64 --amp fp16 --learning_rate 0.0005 --steps 300 --warmup 8 --no_cpu_baseline --print_freq 25 > $O/c.json 2> $O/c.err; line $O/c.json "configs[4] form, fp16, 300 steps"
timeout -k 10 400 python bench.py --batch_size 64 --steps 1500 --warmup 8 --no_cpu_baseline --print_freq 50 > $O/d.json 2> $O/d.err; line $O/d.json "configs[1] at B = 64, 1500 steps"
timeout -k 10 400 python bench.py --head None --steps 800 --warmup 8 --no_cpu_baseline --print_freq 50 > $O/e.json 2> $O/e.err; line $O/e.json "configs[1] with --head None (d = 1280: wide-row K2), 800 steps"
grep -il "nan" $O/*.err; echo done
