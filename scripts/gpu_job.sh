cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_k1d.log
timeout -k 10 300 python scripts/diag_mha.py 2>&1 | grep -v amdgpu.ids > $L
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -x -q -m gpu -k "mha or attention or moco or loop or step" > gpurun_out/r2_tk1d.log 2>&1; echo "pytest rc=$?" >> $L
tail -4 gpurun_out/r2_tk1d.log >> $L
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_k1d
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k1d -o w --output-format csv -- python3 $R/scripts/bench_k1.py 256 512 4 > $R/gpurun_out/prof_k1d.log 2>&1
cd $R
python - >> $L <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_k1d/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'mha_core' in r['Name']: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'])
PY
cat $L
