#!/bin/bash
# the train step's two queues under rocprofv3 --kernel-trace: how much of the side queue's weight gradients runs beside the main
# queue's kernels, with the fixed-grid kernels on two blocks per CU (default) and on one
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/overlap
for v in 2 1; do
  rm -rf gpurun_out/prof_train
  T2V_SK_BLOCKS_PER_CU=$v bash scripts/prof_train.sh > gpurun_out/overlap/summary_percu$v.txt 2>&1
  tr=$(find gpurun_out/prof_train -name "*kernel_trace.csv" | head -1)
  head -1 $tr > gpurun_out/overlap/header.txt
  python scripts/trace_overlap.py $tr 4 | tee gpurun_out/overlap/overlap_percu$v.txt
done
rm -rf gpurun_out/prof_train
