#!/bin/bash
# PMC passes on the 7x7 head kernels (scripts/head_bench.py runs both forms)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SMEM"; do
  i=$((i+1)); out=gpurun_out/pmc_head/p$i; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/head_bench.py > $out/log.txt 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py ${f%_counter_collection.csv} conv_head7x7 2>&1 | tee gpurun_out/pmc_head_p$i.txt
  tail -3 $out/log.txt
done
rm -rf gpurun_out/pmc_head
