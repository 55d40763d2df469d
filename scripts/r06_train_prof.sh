#!/bin/bash
# Round 6: the train step's per-shape kernel tables on ONE stream, batched discriminator passes (default) and pass by pass
# (T2V_D_BATCHED=0), on the same box; then alternating un-profiled timings of both.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
for mode in 1 0; do
  T2V_D_BATCHED=$mode T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 bash scripts/prof_train.sh > $O/train_step_kernel_summary_1stream_b$mode.txt 2>&1
  tr=$(find gpurun_out/prof_train -name "*kernel_trace.csv" | head -1)
  python scripts/trace_shapes.py $tr 4 > $O/train_step_shapes_1stream_b$mode.txt
  rm -rf gpurun_out/prof_train
done
for i in 1 2 3; do for mode in 1 0; do
  echo -n "D_BATCHED=$mode " >> $O/train_ab.txt
  T2V_D_BATCHED=$mode python scripts/train_bench.py --iters 16 2>&1 | tail -1 | cut -c1-110 >> $O/train_ab.txt
done; done
cat $O/train_ab.txt
