# The closing GPU job of a round: full GPU suite, smoke, profile collection and the bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r03}
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/full_suite.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/full_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scripts/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -1 gpurun_out/bench_$TAG.json | cut -c1-160
python bench.py --head None --no_cpu_baseline > gpurun_out/bench_${TAG}_head_none.json 2> gpurun_out/bench_${TAG}_head_none.err; tail -1 gpurun_out/bench_${TAG}_head_none.json | cut -c1-160
python bench.py --model ResNet50 --model_t vit_base_patch16_224 --image_size 512 --batch_size 64 --amp fp16 --steps 10 --warmup 5 --no_cpu_baseline > gpurun_out/bench_${TAG}_config5.json 2> gpurun_out/bench_${TAG}_config5.err; tail -1 gpurun_out/bench_${TAG}_config5.json | cut -c1-160
# the N > 1 code path on RCCL itself with the one rank a one-GPU box allows: the default flat wrap, then the stock DDP reducer
MOMA_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 1 --no_cpu_baseline > gpurun_out/bench_${TAG}_rccl_w1.json 2> gpurun_out/bench_${TAG}_rccl_w1.err; tail -1 gpurun_out/bench_${TAG}_rccl_w1.json | cut -c1-160
MOMA_DP=ddp MOMA_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29673 bench.py --gpus 1 --no_cpu_baseline > gpurun_out/bench_${TAG}_rccl_w1_ddp.json 2> gpurun_out/bench_${TAG}_rccl_w1_ddp.err; tail -1 gpurun_out/bench_${TAG}_rccl_w1_ddp.json | cut -c1-160
