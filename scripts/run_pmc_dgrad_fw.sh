#!/bin/bash
# PMC passes on the data-gradient GEMM of a 1024 -> 1024 ResnetBlock conv at 64x64 in both operand forms
# (scripts/dgrad_fw_bench.py: wino_gemm_sk_kernel<.., false> on the transposed copy, <.., true> on the forward packing)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); out=gpurun_out/pmc_dgrad_fw/p$i; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/dgrad_fw_bench.py 20 > $out/log.txt 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py ${f%_counter_collection.csv} wino_gemm_sk_kernel 2>&1 | tee gpurun_out/pmc_dgrad_fw_p$i.txt
done
rm -rf gpurun_out/pmc_dgrad_fw
