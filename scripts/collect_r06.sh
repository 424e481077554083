#!/bin/bash
# Round 6: kernel traces of K2 (bench shape, wide rows, small batch, slab fallback widths) and of K1's wide-head paths.
# usage (GPU box): bash scripts/collect_r06.sh <outdir>      -> <outdir>/k2_<B>_<d>_<K>_kernel_stats.csv, k1_<N>_<d>_<H>_kernel_stats.csv, *.log
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(cd "$1" && pwd)
cd /tmp && export TMPDIR=/tmp
keep() {  # keep the per-kernel summary only (the raw trace databases are tens of MB)
    f=$(find "$1" -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$2"
    rm -rf "$1"
}
for shape in "256 512 65536" "256 1280 65536" "256 2048 65536" "256 768 65536" "64 512 65536" "64 512 16384" "256 640 65536" "256 896 65536"; do
    tag=${shape// /_}
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k2_$tag -- python3 $R/scripts/bench_k2.py $shape bf16 bf16 30 dq_only >> $OUT/k2.log 2>&1
    keep /tmp/p_k2_$tag $OUT/k2_${tag}_kernel_stats.csv
done
for shape in "512 1280 4" "768 1280 4" "1024 1280 4" "256 512 4"; do
    tag=${shape// /_}
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k1_$tag -- python3 $R/scripts/bench_k1.py $shape bf16 >> $OUT/k1.log 2>&1
    keep /tmp/p_k1_$tag $OUT/k1_${tag}_kernel_stats.csv
done
grep "^K[12]" $OUT/k2.log $OUT/k1.log
