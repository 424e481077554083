cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python scripts/diag_wcache.py > gpurun_out/diag_wcache.log 2>&1; tail -16 gpurun_out/diag_wcache.log
