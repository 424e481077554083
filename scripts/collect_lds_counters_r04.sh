#!/bin/bash
# Round 4: LDS-side counters of the one-pass K2 kernel (what do the score product's row reads cost?), one small group per pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof_r04_lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS"; do
  i=$((i+1)); rm -rf $O/g$i
  timeout -k 5 75 rocprofv3 --pmc $grp -d $O/g$i --output-format csv -- python3 $R/scripts/bench_k2.py 256 512 65536 bf16 bf16 8 > $O/g$i.log 2>&1 || { echo "FAILED group $i: $grp"; tail -n 3 $O/g$i.log | cut -c1-200; }
done
cd $R; python scripts/summarise_pmc.py profiles/r04_k2_lds_counters.csv $O/g*; grep "flash_kernel<512, true>" profiles/r04_k2_lds_counters.csv; cp profiles/r04_k2_lds_counters.csv gpurun_out/
