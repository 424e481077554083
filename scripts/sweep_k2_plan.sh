#!/bin/bash
# Kernel times (rocprofv3 --kernel-trace: device time per launch, not the host-bound loop of bench_k2.py) of the K2 call at a list of
# shapes, for several numbers of workgroups the passes over the queue are cut into (MOMA_K2_TARGET_WG; the product default is 256).
# usage (GPU box, repo root): bash scripts/sweep_k2_plan.sh "B d K" ["B d K" ...]        env WGS="128 256 512" selects the targets
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/k2plan; mkdir -p $O
WGS=${WGS:-"64 128 256 512"}
for shape in "$@"; do
  for wg in $WGS; do
    n=$(echo "${shape}_wg$wg" | tr ' ' '_')
    rm -rf $O/$n
    MOMA_K2_TARGET_WG=$wg timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $O/$n --output-format csv -- python3 $R/scripts/bench_k2.py $shape bf16 bf16 20 > $O/$n.log 2>&1 || { echo "$n FAILED"; continue; }
    python3 - $O/$n "$shape" $wg <<'PY'
import csv, glob, sys
t = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
if not t:
    print(sys.argv[2], "wg", sys.argv[3], "no trace"); raise SystemExit
rows = list(csv.DictReader(open(t[0])))
def stat(pred):
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pred(r["Kernel_Name"]))
    return d
main = stat(lambda n: ("flash_kernel" in n and "true" in n) or "small_kernel" in n)
comb = stat(lambda n: "combine_kernel" in n)
comb = comb[len(comb) // 2:]                      # (the launches with dq: the upper half)
grid = [r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?") for r in rows if "flash_kernel" in r["Kernel_Name"] or "small_kernel" in r["Kernel_Name"]][:1]
if main:
    print("B d K = %-16s target %4s  grid %s  one-pass avg %6.2f min %6.2f | combine avg %6.2f | sum %6.2f us" % (
        sys.argv[2], sys.argv[3], grid, sum(main) / len(main), main[0], sum(comb) / max(1, len(comb)), sum(main) / len(main) + sum(comb) / max(1, len(comb))))
PY
  done
done
