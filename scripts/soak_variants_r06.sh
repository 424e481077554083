#!/bin/bash
# long CLI runs of the loop variants that became graph-served in round 6 (--attn self_mix / self_nomix, --mem MoCoAtt), EfficientNet-B0 pair
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/soak_variants_r06; mkdir -p $O
run() { tag=$1; shift
  timeout -k 10 500 python train_student_moma.py --distill moma --model_s effiB0 --model_t effiB0 --dataset synthetic --image_size 128 --n_cls 4 \
    --batch_size 64 --epochs 3 --steps_per_epoch 250 --nce_k 16384 --head mlp --feat_dim 512 -c 1 -d 1 -b 1 --amp bf16 --print_freq 125 \
    --miopen_find off --save_root /tmp/soak_$tag "$@" > $O/$tag.log 2>&1
  echo "$tag rc=$? $(grep -c -i 'nan' $O/$tag.log) nan-lines; $(grep ' \* Epoch' $O/$tag.log | tail -1)"; }
run self_mix --attn self_mix
run self_nomix --attn self_nomix
run mocoatt_qk --mem MoCoAtt --attn qk
run mocoatt_all_fp32 --mem MoCoAtt --attn all --moma_prec fp32 --nce_k 4096
run mocoatt_dual --mem MoCoAtt --attn dual --nce_k 4096
echo done
