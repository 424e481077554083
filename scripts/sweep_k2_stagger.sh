#!/bin/bash
# Experiment (round 4): row-block stagger of the one-pass K2 kernel.  usage on the GPU box: bash scripts/sweep_k2_stagger.sh "0 1 2 4" [B d K]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stagger; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in $1; do
  export MOMA_K2_STAGGER=$s
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s$s --output-format csv -- python3 $R/scripts/bench_k2.py ${2:-256} ${3:-512} ${4:-65536} bf16 bf16 30 > $O/s$s.log 2>&1 || echo "FAILED $s"
  f=$(find $O/s$s -name "*kernel_stats.csv" | head -1)
  echo "stagger $s: $(grep -h "K2 B=" $O/s$s.log | head -1 | cut -c1-90)"
  grep -h "infonce_flash_kernel\|infonce_combine" $f | awk -F, '{print "    " $1 " calls " $2 " avg_ns " $4 " min " $6 " max " $7}' | cut -c1-200
done
