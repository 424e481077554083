#!/bin/bash
# bench.py flag matrix at a small shape: every case must print a JSON line with a finite loss (rc 0).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/bench_matrix; mkdir -p $O
BASE="--image_size 64 --batch_size 32 --nce_k 4096 --steps 8 --warmup 5 --no_cpu_baseline --learning_rate 0.01"
fail=0; i=0
run() {
  i=$((i+1))
  if timeout -k 10 240 python bench.py $BASE "$@" > $O/case_$i.json 2> $O/case_$i.err && python3 -c "
import json,math,sys; d=json.loads([l for l in open('$O/case_$i.json') if l.startswith('{')][-1]); assert math.isfinite(d['loss_mean_timed_steps']) and d['value'] > 0; print('ok   %-70s %8.1f img/s  replayed %s' % ('$*', d['value'], d['config']['step_graphs']['timed_steps_replayed']))"; then :; else echo "FAIL $*  ($(tail -n 1 $O/case_$i.err | cut -c1-160))"; fail=$((fail+1)); fi
}
run
run --amp none
run --amp fp16
run --channels_last
run --queue_dtype fp32
run --moma_prec fp32 --queue_dtype fp32
run --moma_prec fp32 --queue_dtype bf16
run --head linear
run --head None
run --head None --moma_prec fp32 --queue_dtype fp32
run --num_heads 8
run --batch_size 37 --nce_k 1000
run --batch_size 130 --nce_k 5000 --feat_dim 256
run --feat_dim 128
run --feat_dim 384 --num_heads 8
run --no_graph_student
run --no_overlap_teacher
run --prefetch_queue
run --print_freq 1
run --model resnet8x4
run --model ResNet18 --image_size 64
run --model vit_tiny_patch16_224 --head None --num_heads 3
run --model ResNet18 --model_t vit_tiny_patch16_224 --amp fp16
echo "$i cases, $fail failed"; exit $fail
