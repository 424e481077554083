cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_k1e.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sweeps.py tests/test_gpu_step.py -x -q -m gpu -k "mha or attention or moco or sweep or loop" > gpurun_out/r2_tk1e.log 2>&1; echo "pytest rc=$?" > $L
tail -3 gpurun_out/r2_tk1e.log >> $L
timeout -k 10 300 python scripts/sweep_k1.py 200 21 | tail -2 >> $L
timeout -k 10 300 python scripts/bench_k1.py 256 1280 4 >> $L 2>&1
MOMA_HIP_LIB= timeout -k 10 300 python scripts/bench_k1.py 256 512 4 >> $L 2>&1
grep -v amdgpu.ids $L
