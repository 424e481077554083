cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/full_suite.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/full_suite.log
