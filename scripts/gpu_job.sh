cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_wide18.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_tw18.log 2>&1; echo "pytest rc=$?" > $L
tail -3 gpurun_out/r2_tw18.log >> $L
for t in 1 2 4 8; do
echo "tpb=$t" >> $L
MOMA_K2_COMBINE_TPB=$t timeout -k 10 120 python scripts/bench_k2.py 256 1280 65536 bf16 bf16 30 >> $L 2>&1
done
grep -v amdgpu.ids $L | grep -v "dq=False"
