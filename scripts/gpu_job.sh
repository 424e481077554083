set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_abl8.log
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_t8.log 2>&1; echo "pytest rc=$?" > $L
MOMA_HIP_LIB=$PWD/moma_amd/lib/variants/lib_pf2.so python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" >> gpurun_out/r2_t8.log 2>&1; echo "pf2 pytest rc=$?" >> $L
for v in base pf2 rd6pf2; do MOMA_HIP_LIB=moma_amd/lib/variants/lib_$v.so python scripts/ablate_k2.py >> $L 2>&1; done
python scripts/ablate_k2.py >> $L 2>&1
MOMA_HIP_LIB=moma_amd/lib/variants/lib_st1.so python scripts/stamps_k2.py >> $L 2>&1
grep -v amdgpu.ids $L; tail -3 gpurun_out/r2_t8.log
