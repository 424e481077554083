cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wcomb; mkdir -p $O
for lib in base wcomb_tpb8 wcomb_tpb10; do
 for shape in "64 1280 16384" "256 1280 65536"; do
  n=$(echo "${lib}_$shape" | tr ' ' '_'); rm -rf $O/$n
  MOMA_HIP_LIB=$R/moma_amd/lib/variants/libmoma_$lib.so timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $O/$n --output-format csv -- python3 $R/scripts/bench_k2.py $shape bf16 bf16 20 dq_only > $O/$n.log 2>&1
  python3 - $O/$n "$lib $shape" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
for r in csv.DictReader(open(f[0])):
    if "combine" in r["Name"]: print("%-32s combine avg %.2f us min %.2f" % (sys.argv[2], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
 done
done
