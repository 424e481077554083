cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sweeps.py -x -q -m gpu > gpurun_out/r2_t23.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2_t23.log
