cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "infonce" > gpurun_out/r2_tw7.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r2_tw7.log
