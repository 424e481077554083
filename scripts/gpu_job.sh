# The last GPU job of the round: full GPU suite, smoke, profile collection and the two bench lines.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/full_suite.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/full_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scripts/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1
python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; tail -1 gpurun_out/bench_r02.json | cut -c1-160
python bench.py --head None --no_cpu_baseline > gpurun_out/bench_r02_head_none.json 2> gpurun_out/bench_r02_head_none.err; tail -1 gpurun_out/bench_r02_head_none.json | cut -c1-160
