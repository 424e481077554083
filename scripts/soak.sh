# Long runs of the bench on the GPU box: 400 steps as one rank of an RCCL group (collectives + graphs live), 300 steps with --head None.
cd $GRAFT_REPO_ROOT
MOMA_BENCH_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29672 bench.py --gpus 1 --steps 400 --warmup 8 --no_cpu_baseline 2>gpurun_out/soak1.err | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('rccl_w1 400 steps', l['value'], l['ms_per_step_median'], l['ms_per_step_max'], l['dist']['criterion_allreduce_launches'])"
grep -c "nan" gpurun_out/soak1.err
timeout -k 10 300 python bench.py --head None --steps 300 --warmup 8 --no_cpu_baseline 2>gpurun_out/soak2.err | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('head none 300 steps', l['value'], l['ms_per_step_median'], l['ms_per_step_max'])"
grep -c "nan" gpurun_out/soak2.err
tail -2 gpurun_out/soak2.err | cut -c1-200
