cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_t21.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r2_t21.log
