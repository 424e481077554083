#!/bin/bash
# the N > 1 step on RCCL itself with the one rank a one-GPU box allows, for many steps: every step issues eager collectives (flat buffer
# broadcast, flat gradient all-reduce on RCCL's own stream) between graph replays, with the host running ahead.  usage: bash scripts/soak_rccl_one_rank.sh [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/soak_rccl; mkdir -p $O; N=${1:-600}
for dp in flat ddp; do
  MOMA_BENCH_FORCE_DIST=1 MOMA_DP=$dp timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 200)) \
    bench.py --gpus 1 --steps $N --warmup 8 --no_cpu_baseline > $O/$dp.json 2> $O/$dp.err
  echo "dp=$dp rc=$? $(python3 -c "
import json
try:
    d=json.loads([l for l in open('$O/$dp.json') if l.startswith('{')][-1]); print(d['value'], 'img/s', d['ms_per_step_median'], 'ms median', d['ms_per_step_max'], 'max; replayed', d['config']['step_graphs']['timed_steps_replayed'], 'allreduce launches', d['dist']['criterion_allreduce_launches'], d['dist']['backend'])
except Exception as e: print('no line', e)
")"
done
