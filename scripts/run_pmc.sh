#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) on one kernel_bench.py configuration.
# usage: run_pmc.sh <tag> <kernel_bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  out=gpurun_out/pmc_${tag}/p$i
  mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/kernel_bench.py "$@" --iters 20 --warmup 60 > $out/log.txt 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py ${f%_counter_collection.csv} ${PMC_PAT:-conv_igemm} 2>&1 | tee $out/summary.txt
done
