cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_wide20.log
: > $L
for v in "" sd5 sd7 "" sd5 sd7; do
if [ -z "$v" ]; then echo "sd3" >> $L; timeout -k 10 120 python scripts/bench_k2.py 256 1280 65536 bf16 bf16 30 >> $L 2>&1
else echo $v >> $L; MOMA_HIP_LIB=moma_amd/lib/variants/lib_$v.so timeout -k 10 120 python scripts/bench_k2.py 256 1280 65536 bf16 bf16 30 >> $L 2>&1; fi
done
MOMA_HIP_LIB=moma_amd/lib/variants/lib_sd5.so timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce" > gpurun_out/r2_tw20.log 2>&1; echo "pytest(sd5) rc=$?" >> $L
grep -v amdgpu.ids $L | grep -v "dq=False"
