cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_wide6.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_tw6.log 2>&1; echo "pytest rc=$?" > $L
tail -15 gpurun_out/r2_tw6.log >> $L
timeout -k 10 120 python scripts/bench_k2.py 256 1280 65536 bf16 bf16 20 >> $L 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_w6
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_w6 -o w --output-format csv -- python3 $R/scripts/bench_k2.py 256 1280 65536 bf16 bf16 20 > $R/gpurun_out/prof_w6.log 2>&1
cd $R
python - >> $L <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_w6/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'moma' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
grep -v amdgpu.ids $L
