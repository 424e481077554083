cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_abl14.log
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_t14.log 2>&1; echo "pytest rc=$?" > $L
python scripts/ablate_k2.py >> $L 2>&1
MOMA_HIP_LIB=moma_amd/lib/variants/lib_head.so python scripts/ablate_k2.py >> $L 2>&1
MOMA_HIP_LIB=moma_amd/lib/variants/lib_rot4.so python scripts/ablate_k2.py >> $L 2>&1
python scripts/ablate_k2.py >> $L 2>&1
grep -v amdgpu.ids $L | grep "dq=\|rc="; tail -2 gpurun_out/r2_t14.log
