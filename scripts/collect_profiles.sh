#!/bin/bash
# Collect the rocprofv3 evidence behind DESIGN.md's numbers on the GPU box and condense it into profiles/<tag>_*.
#   usage (on the GPU box, from the repo root):  bash scripts/collect_profiles.sh r02
# One rocprofv3 run per counter group (FETCH_SIZE / WRITE_SIZE / SQ): larger groups exceed the hardware's counter slots.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
prof() { d=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$d.log 2>&1 || echo "FAILED: $d"; echo "done $d"; }
K2="python3 $R/scripts/bench_k2.py 256 512 65536 bf16 bf16"
KW="python3 $R/scripts/bench_k2.py 256 1280 65536 bf16 bf16"
prof k2 --kernel-trace --stats -d $O/k2 --output-format csv -- $K2 20
prof k2_fetch --pmc FETCH_SIZE -d $O/k2_fetch --output-format csv -- $K2 8
prof k2_write --pmc WRITE_SIZE -d $O/k2_write --output-format csv -- $K2 8
prof k2_sq --pmc $SQ -d $O/k2_sq --output-format csv -- $K2 8
prof k2w --kernel-trace --stats -d $O/k2w --output-format csv -- $KW 20
prof k2w_fetch --pmc FETCH_SIZE -d $O/k2w_fetch --output-format csv -- $KW 8
prof k2w_write --pmc WRITE_SIZE -d $O/k2w_write --output-format csv -- $KW 8
prof k2w_sq --pmc $SQ -d $O/k2w_sq --output-format csv -- $KW 8
prof k1 --kernel-trace --stats -d $O/k1 --output-format csv -- python3 $R/scripts/bench_k1.py 256 512 4
prof k1_sq --pmc $SQ -d $O/k1_sq --output-format csv -- python3 $R/scripts/bench_k1.py 256 512 4
prof k4 --kernel-trace --stats -d $O/k4 --output-format csv -- python3 $R/scripts/bench_k4.py
prof step --kernel-trace --stats -d $O/step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 4 --no_cpu_baseline
cd $R
python scripts/summarise_profiles.py ${TAG}_k2 $O/k2
python scripts/summarise_profiles.py ${TAG}_k2_d1280 $O/k2w
python scripts/summarise_profiles.py ${TAG}_k1 $O/k1
python scripts/summarise_profiles.py ${TAG}_k4 $O/k4
python scripts/summarise_profiles.py ${TAG}_step $O/step
python scripts/summarise_pmc.py profiles/${TAG}_k2_pmc.csv $O/k2_fetch $O/k2_write $O/k2_sq
python scripts/summarise_pmc.py profiles/${TAG}_k2_d1280_pmc.csv $O/k2w_fetch $O/k2w_write $O/k2w_sq
python scripts/summarise_pmc.py profiles/${TAG}_k1_pmc.csv $O/k1_sq
cp profiles/${TAG}_*.csv gpurun_out/ 2>/dev/null
