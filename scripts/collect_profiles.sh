#!/bin/bash
# Collect the rocprofv3 evidence behind DESIGN.md's numbers on the GPU box and condense it into profiles/<tag>_*.
#   usage (on the GPU box, from the repo root):  bash scripts/collect_profiles.sh r03
# One rocprofv3 run per counter group (FETCH_SIZE / WRITE_SIZE / SQ): larger groups exceed the hardware's counter slots.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
prof() { d=$1; shift; timeout -k 10 400 rocprofv3 "$@" > $O/$d.log 2>&1 || echo "FAILED: $d"; echo "done $d"; }
K2="python3 $R/scripts/bench_k2.py 256 512 65536 bf16 bf16"
KW="python3 $R/scripts/bench_k2.py 256 1280 65536 bf16 bf16"
KX="python3 $R/scripts/bench_k2.py 256 2048 65536 bf16 bf16"
KF="python3 $R/scripts/bench_k2_f32.py"
K1="python3 $R/scripts/bench_k1.py 256 512 4 bf16"
STEP="python3 $R/bench.py --steps 6 --warmup 4 --no_cpu_baseline"
prof k2 --kernel-trace --stats -d $O/k2 --output-format csv -- $K2 20
prof k2_fetch --pmc FETCH_SIZE -d $O/k2_fetch --output-format csv -- $K2 8
prof k2_write --pmc WRITE_SIZE -d $O/k2_write --output-format csv -- $K2 8
prof k2_sq --pmc $SQ -d $O/k2_sq --output-format csv -- $K2 8
prof k2w --kernel-trace --stats -d $O/k2w --output-format csv -- $KW 20
prof k2x --kernel-trace --stats -d $O/k2x --output-format csv -- $KX 20
prof k2f --kernel-trace --stats -d $O/k2f --output-format csv -- $KF
prof k2f_sq --pmc $SQ -d $O/k2f_sq --output-format csv -- $KF
prof k2f_fetch --pmc FETCH_SIZE -d $O/k2f_fetch --output-format csv -- $KF
prof k1 --kernel-trace --stats -d $O/k1 --output-format csv -- $K1
prof k1_sq --pmc $SQ -d $O/k1_sq --output-format csv -- $K1
prof k4 --kernel-trace --stats -d $O/k4 --output-format csv -- python3 $R/scripts/bench_k4.py
prof step --kernel-trace --stats -d $O/step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 4 --no_cpu_baseline
# HBM traffic of the one-pass K2 kernel INSIDE the training step: PMC passes of the bench command itself, raw per-launch rows kept
prof step_fetch --pmc FETCH_SIZE -d $O/step_fetch --output-format csv -- $STEP
prof step_write --pmc WRITE_SIZE -d $O/step_write --output-format csv -- $STEP
cd $R
python scripts/summarise_profiles.py ${TAG}_k2 $O/k2
python scripts/summarise_profiles.py ${TAG}_k2_d1280 $O/k2w
python scripts/summarise_profiles.py ${TAG}_k2_d2048 $O/k2x
python scripts/summarise_profiles.py ${TAG}_k2_f32 $O/k2f
python scripts/summarise_profiles.py ${TAG}_k1 $O/k1
python scripts/summarise_profiles.py ${TAG}_k4 $O/k4
python scripts/summarise_profiles.py ${TAG}_step $O/step
python scripts/summarise_pmc.py profiles/${TAG}_k2_pmc.csv $O/k2_fetch $O/k2_write $O/k2_sq
python scripts/summarise_pmc.py profiles/${TAG}_k2_f32_pmc.csv $O/k2f_sq $O/k2f_fetch
python scripts/summarise_pmc.py profiles/${TAG}_k1_pmc.csv $O/k1_sq
python scripts/k1_trace_summary.py $O/k1 > profiles/${TAG}_k1_launch_sequence.txt
python scripts/pmc_rows.py profiles/${TAG}_step_k2_traffic_rows.csv infonce_flash_kernel $O/step_fetch $O/step_write
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
