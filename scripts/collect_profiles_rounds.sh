#!/bin/bash
# Round-4 and later additions to the rocprofv3 evidence (the round-3 set is scripts/collect_profiles.sh):
#   K1: HBM / fabric traffic and L2 hit counters next to the SQ counters (the "cold L2" model of DESIGN.md, VERDICT r3 #6a)
#   K2 wide rows: SQ + FETCH / WRITE passes for d = 1280 and d = 2048 (none existed for 2048; 1280 was round 2's)
#   K2 small batch (B = 64: the reference's default --batch_size): kernel stats + SQ + FETCH / WRITE
#   K2 exact-fp32 at d = 1280 (the segment-streamed one-pass kernel); the training step under the step graphs (B = 256 and 64)
# usage (GPU box, repo root):  bash scripts/collect_profiles_rounds.sh [sections]   (TAG=rNN in the environment names the round; default r05)      sections: k1 k2 k2w k2x k2s k2f step step64   (default: all)
TAG=${TAG:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
SECT=${1:-"k1 k2 k2w k2x k2s k2f step step64"}
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
prof() { d=$1; shift; rm -rf $O/$d; timeout -k 10 300 rocprofv3 "$@" > $O/$d.log 2>&1 || echo "FAILED: $d"; echo "done $d"; }
pmc3() {   # pmc3 <name> <program...> : SQ, FETCH_SIZE, WRITE_SIZE, L2 hit / miss passes of one command
  n=$1; shift
  prof ${n}_sq --pmc $SQ -d $O/${n}_sq --output-format csv -- "$@"
  prof ${n}_fetch --pmc FETCH_SIZE -d $O/${n}_fetch --output-format csv -- "$@"
  prof ${n}_write --pmc WRITE_SIZE -d $O/${n}_write --output-format csv -- "$@"
  prof ${n}_tcc --pmc TCC_HIT_sum TCC_MISS_sum -d $O/${n}_tcc --output-format csv -- "$@"
}
for s in $SECT; do
case $s in
k1)
  K1="python3 $R/scripts/bench_k1.py 256 512 4 bf16"
  prof k1 --kernel-trace --stats -d $O/k1 --output-format csv -- $K1
  pmc3 k1 $K1
  cd $R; python scripts/summarise_profiles.py ${TAG}_k1 $O/k1
  python scripts/summarise_pmc.py profiles/${TAG}_k1_pmc.csv $O/k1_sq $O/k1_fetch $O/k1_write $O/k1_tcc
  python scripts/k1_trace_summary.py $O/k1 > profiles/${TAG}_k1_launch_sequence.txt; cd /tmp ;;
k2w)
  KW="python3 $R/scripts/bench_k2.py 256 1280 65536 bf16 bf16 8"
  prof k2w --kernel-trace --stats -d $O/k2w --output-format csv -- python3 $R/scripts/bench_k2.py 256 1280 65536 bf16 bf16 20
  pmc3 k2w $KW
  cd $R; python scripts/summarise_profiles.py ${TAG}_k2_d1280 $O/k2w
  python scripts/summarise_pmc.py profiles/${TAG}_k2_d1280_pmc.csv $O/k2w_sq $O/k2w_fetch $O/k2w_write $O/k2w_tcc; cd /tmp ;;
k2x)
  KX="python3 $R/scripts/bench_k2.py 256 2048 65536 bf16 bf16 8"
  prof k2x --kernel-trace --stats -d $O/k2x --output-format csv -- python3 $R/scripts/bench_k2.py 256 2048 65536 bf16 bf16 20
  pmc3 k2x $KX
  cd $R; python scripts/summarise_profiles.py ${TAG}_k2_d2048 $O/k2x
  python scripts/summarise_pmc.py profiles/${TAG}_k2_d2048_pmc.csv $O/k2x_sq $O/k2x_fetch $O/k2x_write $O/k2x_tcc; cd /tmp ;;
k2s)
  KS="python3 $R/scripts/bench_k2.py 64 512 65536 bf16 bf16 8"
  prof k2s --kernel-trace --stats -d $O/k2s --output-format csv -- python3 $R/scripts/bench_k2.py 64 512 65536 bf16 bf16 30
  pmc3 k2s $KS
  cd $R; python scripts/summarise_profiles.py ${TAG}_k2_b64 $O/k2s
  python scripts/summarise_pmc.py profiles/${TAG}_k2_b64_pmc.csv $O/k2s_sq $O/k2s_fetch $O/k2s_write $O/k2s_tcc; cd /tmp ;;
k2)
  K2="python3 $R/scripts/bench_k2.py 256 512 65536 bf16 bf16 12 dq_only"     # (the step's call: flash kernel with dq + its combine)
  prof k2 --kernel-trace --stats -d $O/k2 --output-format csv -- python3 $R/scripts/bench_k2.py 256 512 65536 bf16 bf16 20
  pmc3 k2 $K2
  cd $R; python scripts/summarise_profiles.py ${TAG}_k2 $O/k2
  python scripts/summarise_pmc.py profiles/${TAG}_k2_pmc.csv $O/k2_sq $O/k2_fetch $O/k2_write $O/k2_tcc; cd /tmp ;;
k2f)
  KF="python3 $R/scripts/bench_k2_f32.py 256 1280 65536"
  prof k2f --kernel-trace --stats -d $O/k2f --output-format csv -- $KF
  prof k2f_sq --pmc $SQ -d $O/k2f_sq --output-format csv -- $KF
  prof k2f_fetch --pmc FETCH_SIZE -d $O/k2f_fetch --output-format csv -- $KF
  cd $R; python scripts/summarise_profiles.py ${TAG}_k2_f32_d1280 $O/k2f
  python scripts/summarise_pmc.py profiles/${TAG}_k2_f32_d1280_pmc.csv $O/k2f_sq $O/k2f_fetch; cd /tmp ;;
step64)
  prof step64 --kernel-trace --stats -d $O/step64 --output-format csv -- python3 $R/bench.py --steps 10 --warmup 5 --no_cpu_baseline --batch_size 64
  cd $R; python scripts/summarise_profiles.py ${TAG}_step_b64 $O/step64; cd /tmp ;;
step)
  prof step --kernel-trace --stats -d $O/step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 5 --no_cpu_baseline
  cd $R; python scripts/summarise_profiles.py ${TAG}_step $O/step; cd /tmp ;;
esac
done
cd $R
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
ls profiles/${TAG}_*
