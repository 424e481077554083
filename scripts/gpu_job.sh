set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "mha or mocoatt or abi or infonce_golden" > gpurun_out/r2_k1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_k1.log
tail -25 gpurun_out/r2_k1.log
python -m pytest tests/test_gpu_step.py -q -m gpu > gpurun_out/r2_step.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_step.log
tail -25 gpurun_out/r2_step.log
python scripts/bench_k1.py > gpurun_out/r2_k1bench.log 2>&1; grep -v amdgpu gpurun_out/r2_k1bench.log
