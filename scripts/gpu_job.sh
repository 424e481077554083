cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -x -q -m gpu -k "infonce or moco or loop or step" > gpurun_out/r2_tw15.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2_tw15.log
python bench.py --head None --no_cpu_baseline > gpurun_out/bench_r02_head_none.json 2> gpurun_out/bench_r02_head_none.err; tail -1 gpurun_out/bench_r02_head_none.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], j['ms_per_step'], r['frac'], r['ms_per_launch'], r['whole_call_ms'])"
