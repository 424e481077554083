cd /tmp && export TMPDIR=/tmp
for v in "" 1; do
export K1_X_BF16=$v
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_k1x$v --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/bench_k1.py 256 512 4 bf16 > $GRAFT_REPO_ROOT/gpurun_out/k1x$v.log 2>&1
echo "x bf16=$v"; python3 $GRAFT_REPO_ROOT/scripts/k1_trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_k1x$v
done
