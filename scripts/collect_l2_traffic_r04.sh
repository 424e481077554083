#!/bin/bash
# Round 4: L2 -> CU traffic of the K2 kernels (TCC read requests / sectors), next to the HBM-side FETCH_SIZE of collect_profiles_r04.sh:
# the per-CU ingest figures of DESIGN section 4.   usage (GPU box, repo root): bash scripts/collect_l2_traffic_r04.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_r04_l2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in "256 512" "256 1280" "64 512"; do
  set -- $S; n=b$1_d$2
  # (one counter per pass: several TCC counters at once exceed the hardware's slots and rocprofv3 then aborts -- and hangs until killed)
  for c in TCC_READ_sum TCC_READ_SECTORS_sum; do
    rm -rf $O/${n}_$c; timeout -k 5 75 rocprofv3 --pmc $c -d $O/${n}_$c --output-format csv -- python3 $R/scripts/bench_k2.py $1 $2 65536 bf16 bf16 8 > $O/${n}_$c.log 2>&1 || { echo "FAILED $n $c"; exit 1; }
  done
done
cd $R; python scripts/summarise_pmc.py profiles/r04_k2_l2_traffic.csv $O/b*_d*_TCC_*; cat profiles/r04_k2_l2_traffic.csv; cp profiles/r04_k2_l2_traffic.csv gpurun_out/
