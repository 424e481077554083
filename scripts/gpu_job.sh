cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_abl21.log
MOMA_HIP_LIB=moma_amd/lib/variants/lib_sm1.so timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_t24.log 2>&1; echo "pytest(sm1) rc=$?" > $L
for i in 1 2 3; do
python scripts/ablate_k2.py >> $L 2>&1
MOMA_HIP_LIB=moma_amd/lib/variants/lib_sm1.so python scripts/ablate_k2.py >> $L 2>&1
done
grep -v amdgpu.ids $L | grep "rc=\|dq=True"
