#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  port=$((20000 + RANDOM % 20000))
  MOMA_BENCH_SAME_DEVICE=1 MOMA_BENCH_BACKEND=gloo MOMA_BENCH_FORCE_OVERLAP=1 MOMA_DP=$1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port $R/bench.py --gpus 2 --steps $3 --warmup 5 --batch_size 32 --image_size 64 --nce_k 4096 --no_cpu_baseline --learning_rate $4 $2 > $R/gpurun_out/soak.out 2> $R/gpurun_out/soak.err
  rc=$?
  echo "dp=$1 extra='$2' steps=$3 lr=$4 rc=$rc $(python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$R/gpurun_out/soak.out') if l.startswith('{')][-1]); print('loss', d['loss_mean_timed_steps'], 'spread', d['dist']['replica_checksum_spread'], 'replayed', d['config']['step_graphs']['timed_steps_replayed'])
except Exception as e: print('no line', e)
")"
}
run flat "" 300 0.01; run flat "" 300 0.01; run flat "" 300 0.05; run flat "--no_graph_student" 300 0.05
