cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python scripts/sweep_k1.py 150 3 > gpurun_out/sweep_k1.log 2>&1; echo "rc=$?"
grep -v amdgpu.ids gpurun_out/sweep_k1.log | grep -v "^ok" | tail -12
