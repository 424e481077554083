cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_k1f.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sweeps.py tests/test_gpu_step.py -x -q -m gpu -k "mha or attention or moco or sweep or loop" > gpurun_out/r2_tk1f.log 2>&1; echo "pytest rc=$?" > $L
tail -3 gpurun_out/r2_tk1f.log >> $L
timeout -k 10 300 python scripts/sweep_k1.py 200 31 | tail -1 >> $L
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_k1f
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_k1f -o w --output-format csv -- python3 $R/scripts/bench_k1.py 256 512 4 > $R/gpurun_out/prof_k1f.log 2>&1
cd $R
python - >> $L <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_k1f/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'ksplit' in n: d[(n.split('(')[0][-36:], r['Grid_Size_X'], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in sorted(d.items()):
    v.sort(); print(k, len(v), 'min %.1f med %.1f'%(v[0], v[len(v)//2]))
PY
grep -v amdgpu.ids $L
