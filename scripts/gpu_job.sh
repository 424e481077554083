set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_abl4.log
MOMA_HIP_LIB=$PWD/moma_amd/lib/variants/lib_zm.so python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_t3.log 2>&1; echo "zm pytest rc=$?" > $L
MOMA_HIP_LIB=$PWD/moma_amd/lib/variants/lib_pro2.so python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" >> gpurun_out/r2_t3.log 2>&1; echo "pro2 pytest rc=$?" >> $L
for v in zm pro1 pro2 pro3 pf4 pf2 rd10 rd6 zm; do MOMA_HIP_LIB=moma_amd/lib/variants/lib_$v.so python scripts/ablate_k2.py >> $L 2>&1; done
for v in zm_st1 pro1_st1 pro2_st1; do echo $v >> $L; MOMA_HIP_LIB=moma_amd/lib/variants/lib_$v.so python scripts/stamps_k2.py >> $L 2>&1; done
grep -v amdgpu.ids $L
