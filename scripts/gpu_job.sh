set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/prof_r02b
rm -rf $R; mkdir -p $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py -q -m gpu -x -k "mha or mocoatt or infonce_golden or infonce_vs or loop or shuffle" > gpurun_out/r2_k1b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_k1b.log
tail -5 gpurun_out/r2_k1b.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/k2_trace -- python3 scripts/bench_k2.py 256 512 65536 bf16 bf16 40 > $R/k2_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/k1_trace -- python3 scripts/bench_k1.py 256 512 4 > $R/k1_trace.log 2>&1
python scripts/bench_k2.py 256 512 65536 bf16 bf16 50 > $R/k2_plain.log 2>&1
python scripts/bench_k4.py 200 > $R/k4_plain.log 2>&1
grep -h "K2 B=\|K1 N=\|K4 alone" $R/*.log
python3 - <<'PY'
import csv,glob,os
R=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/prof_r02b"
for d in ("k2_trace","k1_trace"):
    f=glob.glob(R+"/"+d+"/*/*kernel_stats.csv")[0]
    print(d)
    for r in csv.DictReader(open(f)):
        if "moma" in r["Name"]:
            print("  %-80s calls %5s avg %8.2f us min %8.2f max %8.2f"%(r["Name"].replace("void moma::(anonymous namespace)::","").replace("moma::(anonymous namespace)::","")[:80], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
