#!/bin/bash
# kernel time of every ablation variant built by scripts/build_k2_variants.py (rocprofv3 kernel trace of scripts/bench_k2.py)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/k2var; mkdir -p $O
for lib in $R/moma_amd/lib/variants/libmoma_*.so; do
  n=$(basename $lib .so); n=${n#libmoma_}
  rm -rf $O/$n
  MOMA_HIP_LIB=$lib timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/$n --output-format csv -- python3 $R/scripts/bench_k2.py ${1:-256} ${2:-512} 65536 bf16 bf16 20 > $O/$n.log 2>&1
  python3 - $O/$n $n <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_stats.csv")
if not f: print(sys.argv[2],"no stats"); raise SystemExit
for r in csv.DictReader(open(f[0])):
    if "flash_kernel" in r["Name"] and "true" in r["Name"] or "small_kernel" in r["Name"] or "wide_" in r["Name"]:
        print("%-14s %s avg %.2f us  min %.2f"%(sys.argv[2], r["Name"].split("(")[0][-28:], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
t=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")
if t:
    d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(t[0])) if "combine_kernel" in r["Kernel_Name"])
    up=d[len(d)//2:]
    if up: print("%-14s combine (dq launches) avg %.2f us  min %.2f  max %.2f"%(sys.argv[2], sum(up)/len(up), up[0], up[-1]))
PY
done
