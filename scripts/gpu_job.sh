cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r2_cmb.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "infonce or moco" > gpurun_out/r2_tcmb.log 2>&1; echo "pytest rc=$?" > $L
tail -4 gpurun_out/r2_tcmb.log >> $L
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_cmb
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cmb -o w --output-format csv -- python3 $R/scripts/bench_k2.py 256 512 65536 bf16 bf16 30 > $R/gpurun_out/prof_cmb.log 2>&1
cd $R
grep "K2 B" gpurun_out/prof_cmb.log >> $L
python - >> $L <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_cmb/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'moma' in n: d[(n.split('(')[0][-46:], r['Grid_Size_Y'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in d.items():
    v.sort(); print(k, len(v), 'min %.1f med %.1f'%(v[0], v[len(v)//2]))
PY
grep -v amdgpu.ids $L
