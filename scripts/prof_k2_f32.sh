#!/bin/bash
# rocprofv3 kernel-trace of scripts/bench_k2_f32.py at one shape.  usage: bash scripts/prof_k2_f32.sh B d K
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_k2f_$1_$2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/scripts/bench_k2_f32.py $1 $2 $3 > $O/run.log 2>&1 || echo FAILED
grep "K2 fp32" $O/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "moma" in r["Name"]:
        print("   %-70s calls %4s avg %9.2f us  min %9.2f" % (r["Name"].replace("void moma::(anonymous namespace)::", "").split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
