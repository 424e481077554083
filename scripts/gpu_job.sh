cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_wide11.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce" > gpurun_out/r2_tw11.log 2>&1; echo "pytest rc=$?" > $L
tail -4 gpurun_out/r2_tw11.log >> $L
timeout -k 10 120 python scripts/bench_k2.py 256 1280 65536 bf16 bf16 20 >> $L 2>&1
timeout -k 10 120 python scripts/bench_k2.py 256 768 65536 bf16 bf16 20 >> $L 2>&1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_w4
rm -rf $O $R/gpurun_out/prof_w11
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d $O/a -o a --output-format csv -- python3 $R/scripts/bench_k2.py 256 1280 65536 bf16 bf16 6 > $O.a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_w11 -o w --output-format csv -- python3 $R/scripts/bench_k2.py 256 1280 65536 bf16 bf16 20 > $R/gpurun_out/prof_w11.log 2>&1
cd $R
python scripts/summarise_pmc.py gpurun_out/pmc_w4.csv $O/a >> $L
grep "wide" gpurun_out/pmc_w4.csv >> $L
python - >> $L <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_w11/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'moma' in r['Name']: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
grep -v amdgpu.ids $L
