cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_abl22.log
: > $L
for i in 1 2 3; do
python scripts/ablate_k2.py >> $L 2>&1
MOMA_HIP_LIB=moma_amd/lib/variants/lib_nt.so python scripts/ablate_k2.py >> $L 2>&1
done
python scripts/bench_k2.py 256 512 65536 bf16 bf16 30 >> $L 2>&1
MOMA_HIP_LIB=moma_amd/lib/variants/lib_nt.so python scripts/bench_k2.py 256 512 65536 bf16 bf16 30 >> $L 2>&1
grep -v amdgpu.ids $L | grep "dq=True"
