#!/bin/bash
# A matrix of train_student_moma.py invocations (2 epochs x 6 steps on synthetic data each): every --mem / --attn / --head family, both
# precision policies, both Shuffle-BN modes, graphs on / off.  Prints one line per case; exit code = number of failures.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/cli_matrix; mkdir -p $O; S=/tmp/cli_matrix_save   # (checkpoints stay off gpurun_out)
BASE="--distill moma --model_s resnet8x4 --model_t resnet8x4 --dataset cifar100 --n_cls 3 --batch_size 32 --epochs 2 --steps_per_epoch 6 --nce_k 1024 --feat_dim 128 -c 1 -d 1 -b 1 --print_freq 2 --miopen_find off"
fail=0; i=0
run() {
  i=$((i+1)); rm -rf $S
  if timeout -k 10 240 python train_student_moma.py $BASE --save_root $S "$@" > $O/case_$i.log 2>&1 && grep -q "best accuracy" $O/case_$i.log && ! grep -qi "nan" $O/case_$i.log; then
    echo "ok   $*"; else echo "FAIL $*  ($(tail -n 1 $O/case_$i.log | cut -c1-160))"; fail=$((fail+1)); fi
}
run --head mlp
run --head linear
run --head None
run --head mlp --moma_prec bf16 --queue_dtype bf16
run --head mlp --moma_prec bf16 --queue_dtype fp32
run --head mlp --moma_prec bf16 --queue_dtype bf16 --amp bf16
run --head mlp --amp fp16
expect_refusal() {      # configurations the reference's own loop cannot run: a clear NotImplementedError, no crash mid-step
  i=$((i+1)); rm -rf $S
  timeout -k 10 240 python train_student_moma.py $BASE --save_root $S "$@" > $O/case_$i.log 2>&1
  if grep -q "NotImplementedError" $O/case_$i.log; then echo "ok   (refused) $*"; else echo "FAIL (no refusal) $*"; fail=$((fail+1)); fi
}
expect_refusal --head mlp --mem MoCoST
expect_refusal --head mlp --mem MoCoSSTT
expect_refusal --head mlp --mem MoCoAtt --attn dual2
for a in qk dual self_qk all self; do run --head mlp --mem MoCoAtt --attn $a; done
run --head mlp --mem MoCoAtt --attn qk --moma_prec bf16 --queue_dtype bf16
run --head mlp --attn self_mix
run --head mlp --attn self_nomix
run --head mlp --shuffle_bn gather
run --head mlp --no_graph_student --no_graph_teacher --no_overlap_teacher
run --head mlp --no_fused
run --head mlp --moma_prec bf16 --queue_dtype bf16 --batch_size 100 --nce_k 1000
run --distill kd
run --head mlp --model_s effiB0 --model_t effiB0 --moma_prec bf16 --queue_dtype bf16 --amp bf16 --batch_size 16
run --head None --model_s vit_tiny_patch16_224 --model_t vit_tiny_patch16_224 --moma_prec bf16 --queue_dtype bf16 --amp bf16 --batch_size 16 --num_heads 3
echo "$i cases, $fail failed"; exit $fail
