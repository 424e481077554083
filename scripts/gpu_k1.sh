# K1 development job: MHA parity tests, then the K1 micro-benchmark under rocprofv3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mha or mocoatt" > gpurun_out/k1_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -15 gpurun_out/k1_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
python scripts/bench_k1.py 256 512 4 > gpurun_out/k1_bench.log 2>&1; cat gpurun_out/k1_bench.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_k1 --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_k1.py 256 512 4 > $GRAFT_REPO_ROOT/gpurun_out/k1_prof.log 2>&1
cd $GRAFT_REPO_ROOT && python scripts/summarise_profiles.py dev_k1 gpurun_out/prof_k1 && head -20 profiles/dev_k1_kernel_stats.csv
bash scripts/gpu_k1b.sh
