cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --head None --no_cpu_baseline > gpurun_out/bench_r02_head_none.json 2> gpurun_out/bench_r02_head_none.err; tail -1 gpurun_out/bench_r02_head_none.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('none', j['value'], j['ms_per_step'], r['frac'], r['ms_per_launch'], r['whole_call_ms'], r['other_ms'])"
