#!/bin/bash
# third CLI matrix: degenerate queue sizes (K < B: every step overwrites the whole ring, last writer wins; K = 1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/cli_matrix3; mkdir -p $O; S=/tmp/cli_matrix3_save
BASE="--distill moma --model_s resnet8x4 --model_t resnet8x4 --dataset cifar100 --n_cls 3 --batch_size 32 --epochs 2 --steps_per_epoch 6 --feat_dim 128 --head mlp -c 1 -d 1 -b 1 --print_freq 2 --miopen_find off"
fail=0; i=0
run() {
  i=$((i+1)); rm -rf $S
  if timeout -k 10 240 python train_student_moma.py $BASE --save_root $S "$@" > $O/case_$i.log 2>&1 && grep -q "best accuracy" $O/case_$i.log && ! grep -qi "nan" $O/case_$i.log; then
    echo "ok   $*"; else echo "FAIL $*  ($(tail -n 1 $O/case_$i.log | cut -c1-160))"; fail=$((fail+1)); fi
}
for k in 1 16 33 100; do
  run --nce_k $k
  run --nce_k $k --moma_prec bf16 --queue_dtype bf16
  run --nce_k $k --moma_prec bf16 --queue_dtype fp32
done
run --nce_k 16 --mem MoCoAtt --attn all
run --nce_k 16 --shuffle_bn gather
run --nce_k 7 --batch_size 1 --head None
run --nce_k 7 --batch_size 2 --moma_prec bf16 --queue_dtype bf16
echo "$i cases, $fail failed"; exit $fail
