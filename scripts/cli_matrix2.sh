#!/bin/bash
# second CLI matrix: schedule / bookkeeping flags of the reference's parser
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/cli_matrix2; mkdir -p $O; S=/tmp/cli_matrix2_save
BASE="--distill moma --model_s resnet8x4 --model_t resnet8x4 --dataset cifar100 --n_cls 3 --batch_size 32 --epochs 3 --steps_per_epoch 5 --nce_k 1024 --feat_dim 128 --head mlp -c 1 -d 1 -b 1 --print_freq 2 --miopen_find off"
fail=0; i=0
run() {
  i=$((i+1)); rm -rf $S
  if timeout -k 10 240 python train_student_moma.py $BASE --save_root $S "$@" > $O/case_$i.log 2>&1 && grep -q "best accuracy" $O/case_$i.log && ! grep -qi "nan" $O/case_$i.log; then
    echo "ok   $*"; else echo "FAIL $*  ($(tail -n 1 $O/case_$i.log | cut -c1-160))"; fail=$((fail+1)); fi
}
run --cosine
run --deterministic
run --skip_validation
run --lr_decay_epochs 1,2 --lr_decay_rate 0.5
run --trial 3 --seed 7
run --std_strict --tec_strict
run --alpha 0.9 --nce_t 0.07 --nce_m 0.5 --kd_T 2
run -c 0 -d 0 -b 1
run -c 1 -d 0 -b 0
run --dp ddp
run --dp flat
run --image_size 48
run --dataset prostate_hv --image_size 64 --n_cls 4
run --model_s resnet8 --model_t resnet32x4
run --aug_train RA
run --dali gpu
run --weight_decay 0 --momentum 0 --learning_rate 0.001
echo "$i cases, $fail failed"; exit $fail
