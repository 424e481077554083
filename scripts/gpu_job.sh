cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1
tail -12 gpurun_out/collect_r02.log
python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; tail -1 gpurun_out/bench_r02.json | cut -c1-400
python bench.py --head None --no_cpu_baseline > gpurun_out/bench_r02_head_none.json 2> gpurun_out/bench_r02_head_none.err; tail -1 gpurun_out/bench_r02_head_none.json | cut -c1-300
