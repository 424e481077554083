cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "overflow_repass" > gpurun_out/r2_tw19.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r2_tw19.log
