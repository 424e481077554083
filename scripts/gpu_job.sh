cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_t22.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r2_t22.log
for i in 1 2; do
python bench.py --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('mlp ', j['value'], j['ms_per_step'], r['frac'], r['ms_per_launch'], r['whole_call_ms'])"
python bench.py --head None --no_cpu_baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('none', j['value'], j['ms_per_step'], r['frac'], r['ms_per_launch'], r['whole_call_ms'])"
done
