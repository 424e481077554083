#!/bin/bash
# Round 6: what an N > 1 run leaves behind -- rehearsed with several ranks on ONE GPU (gloo carries the collectives: RCCL refuses two
# ranks on a device).  (1) a normal launcher-less run: the phase lines of every rank; (2) the parent's deadline on a job that cannot
# finish in time: every rank stopped, per-rank phase report + stderr tails, exit 124; (3) the rank-side watchdog under
# torch.distributed.run (how the driver starts its N > 1 runs): the stuck rank names its phase and dumps its stacks.
# usage (GPU box, repo root): bash scripts/rehearse_first_contact.sh <outfile>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
SMALL="--steps 6 --warmup 5 --batch_size 32 --image_size 64 --nce_k 4096 --no_cpu_baseline"
export MOMA_BENCH_SAME_DEVICE=1 MOMA_BENCH_BACKEND=gloo MOMA_BENCH_FORCE_OVERLAP=1
{
echo "=== (1) python bench.py --gpus 4 $SMALL   (no launcher: the parent starts the ranks)"
python $R/bench.py --gpus 4 $SMALL 2>&1 >/tmp/line1.json | grep -E "phase|starting|deadline|exited" ; echo "exit code ${PIPESTATUS[0]}"; python3 -c "
import json; d=json.load(open('/tmp/line1.json')); print('line:', d['value'], d['unit'], 'n_gpus', d['n_gpus'], 'spread', d['dist']['replica_checksum_spread'])"
echo
echo "=== (2) the same at full size with --launch_timeout 8 (cannot finish: model build + warm-up of four ranks on one GPU take longer)"
python $R/bench.py --gpus 4 --steps 20 --warmup 5 --no_cpu_baseline --launch_timeout 8 2>&1 >/dev/null | grep -vE "^\[bench\] \.\.\. still|amdgpu.ids" | tail -40; echo "exit code ${PIPESTATUS[0]}"
echo
echo "=== (3) MOMA_BENCH_RANK_DEADLINE=4 python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (full size): the ranks' own watchdog"
MOMA_BENCH_RANK_DEADLINE=4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 $R/bench.py --gpus 2 --steps 20 --warmup 5 --no_cpu_baseline 2>&1 >/dev/null | grep -E "\[bench\] rank|deadline|phase|File |Thread|most recent|exitcode|ChildFailed" | head -40; echo "exit code ${PIPESTATUS[0]}"
} > $OUT 2>&1
tail -5 $OUT
