#!/bin/bash
# --attn self_mix went NaN in the second epoch of a 750-step CLI soak (EfficientNet-B0, bf16 autocast): graph-served path or training?
# usage: bash scripts/diag_selfmix_nan.sh [runs per variant]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/selfmix_nan; mkdir -p $O; N=${1:-4}
run() { tag=$1; shift
  timeout -k 10 500 python train_student_moma.py --distill moma --model_s effiB0 --model_t effiB0 --dataset synthetic --image_size 128 --n_cls 4 \
    --batch_size 64 --epochs 2 --steps_per_epoch 250 --nce_k 16384 --head mlp --feat_dim 512 -c 1 -d 1 -b 1 --amp bf16 --print_freq 5 \
    --miopen_find off --save_root /tmp/nan_$tag "$@" > $O/$tag.log 2>&1
  first=$(grep -n "Loss nan" $O/$tag.log | head -1 | cut -d: -f2- | cut -c1-22)
  echo "$tag first nan: ${first:-none}; $(grep ' \* Epoch' $O/$tag.log | tail -1 | cut -c1-60)"; }
for i in $(seq 1 $N); do run graphs_mix_$i --attn self_mix; done
for i in $(seq 1 $N); do run eager_mix_$i --attn self_mix --no_graph_student; done
for i in $(seq 1 $N); do run graphs_self_$i --attn self; done
echo done
