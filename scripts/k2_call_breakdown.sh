#!/bin/bash
# Per-kernel device time of the K2 call at a list of shapes (rocprofv3 --kernel-trace --stats of scripts/bench_k2.py ... dq_only).
# usage (GPU box, repo root): bash scripts/k2_call_breakdown.sh "B d K [queue_dtype prec]" ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/k2calls; mkdir -p $O
for shape in "$@"; do
  set -- $shape; B=$1; d=$2; K=$3; qd=${4:-bf16}; pr=${5:-bf16}
  n="${B}_${d}_${K}_${qd}_${pr}"; rm -rf $O/$n
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/$n --output-format csv -- python3 $R/scripts/bench_k2.py $B $d $K $qd $pr 20 dq_only > $O/$n.log 2>&1 || { echo "$n FAILED"; continue; }
  python3 - $O/$n "$shape" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
rows = [r for r in csv.DictReader(open(f[0])) if "moma::" in r["Name"]]
tot = 0.0
print("B d K = %s" % sys.argv[2])
for r in rows:
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    per_call = avg * calls / 25.0            # 5 warm-up + 20 timed calls
    tot += per_call
    print("    %-60s calls %3d  avg %8.2f us  per K2 call %8.2f us" % (r["Name"].replace("void ", "").replace("moma::(anonymous namespace)::", "moma::").split("(")[0][:60], calls, avg, per_call))
print("    sum of kernel time per K2 call: %.2f us" % tot)
PY
done
