#!/bin/bash
# train step: the fixed-grid kernels on one block per CU while the side stream runs weight gradients (overlap hint 2, default)
# against two per CU throughout (T2V_TRAIN_SK_HINT=0), alternating on one box; then both under rocprofv3 for the two-queue picture
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/skhint; mkdir -p $O
for i in 1 2 3 4; do for v in 0 1; do
  echo -n "T2V_TRAIN_SK_HINT=$v "; T2V_TRAIN_SK_HINT=$v python scripts/train_bench.py --iters 16 2>/dev/null | grep "ms/step" | cut -c1-150
done; done | tee $O/ab_train.txt
for v in 0 1; do
  rm -rf gpurun_out/prof_train
  T2V_TRAIN_SK_HINT=$v bash scripts/prof_train.sh > $O/summary_hint$v.txt 2>&1
  tr=$(find gpurun_out/prof_train -name "*kernel_trace.csv" | head -1)
  python scripts/trace_overlap.py $tr 4 60-61.3 > $O/two_queues_hint$v.txt 2>&1
  grep -E "^last step|^  queue|^  idle|^  side|^  in that" $O/two_queues_hint$v.txt
done
rm -rf gpurun_out/prof_train
