cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; tail -1 gpurun_out/bench_r02.json | cut -c1-160
python bench.py --head None --no_cpu_baseline > gpurun_out/bench_r02_head_none.json 2> gpurun_out/bench_r02_head_none.err; tail -1 gpurun_out/bench_r02_head_none.json | cut -c1-160
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_step2
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_step2 --output-format csv -- python3 $R/bench.py --steps 10 --warmup 4 --no_cpu_baseline > $R/gpurun_out/prof_step2.log 2>&1
cd $R
python scripts/summarise_profiles.py r02_step gpurun_out/prof_step2
cp profiles/r02_step_kernel_stats.csv gpurun_out/
grep "ms_per_launch" gpurun_out/prof_step2.log | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('under rocprof: bench ms_per_launch', j['roofline']['ms_per_launch'])"
grep "infonce_flash" profiles/r02_step_kernel_stats.csv | cut -c1-40,150-
