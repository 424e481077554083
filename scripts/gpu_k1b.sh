cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_k1p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_k1_parts.py > $GRAFT_REPO_ROOT/gpurun_out/k1p.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_k1p/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ks=[(r['Kernel_Name'],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r['Grid_Size_X']) for r in rows if 'k1_' in r['Kernel_Name']]
import collections
agg=collections.defaultdict(list)
for n,t,g in ks: agg[(n.split('(')[0][-22:],g)].append(t)
for k,v in agg.items():
    v=sorted(v); print(k, len(v), "median %.2f min %.2f"%(v[len(v)//2], v[0]))
PY
