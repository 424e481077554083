cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "wide_queue" -s > gpurun_out/r2_t25.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/r2_t25.log | tail -14
