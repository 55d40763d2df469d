#!/bin/bash
# PMC passes on the weight-gradient kernel (one counter group per run, kernel-trace only).
# usage: run_pmc_wgrad.sh <shape> [images per launch]     (a key of scripts/kernel_bench_shapes.py, default rb1024; default 1 image)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
shape=${1:-rb1024}
batch=${2:-1}
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  out=gpurun_out/pmc_wgrad_$shape/p$i
  mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -o pmc -- python scripts/wgrad_bench.py --shapes $shape --batch $batch --iters 10 --warmup 10 > $out/log.txt 2>&1
  f=$(find $out -name "*counter_collection.csv" | head -1)
  python scripts/pmc_summary.py ${f%_counter_collection.csv} conv_wgrad 2>&1 | tee $out/summary.txt
done
