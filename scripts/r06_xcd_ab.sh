#!/bin/bash
# T2V_XCD_SLICES A/B: per-frame kernel tables on one stream (rocprofv3), un-profiled frames and train steps alternating on one box
mkdir -p gpurun_out/xcd
for v in 0 1; do
  T2V_XCD_SLICES=$v T2V_STREAMS=1 bash scripts/prof_frames.sh xcd$v --frames 50 > gpurun_out/xcd/frames_1stream_xcd$v.txt 2>&1
  grep -E "FRAMES|winograd4_input|kernel time" gpurun_out/xcd/frames_1stream_xcd$v.txt | cut -c1-150
done
bash scripts/ab_frames.sh T2V_XCD_SLICES=0 T2V_XCD_SLICES=1 3 2>&1 | tee gpurun_out/xcd/ab_frames.txt
for i in 1 2 3; do for v in 0 1; do
  echo -n "T2V_XCD_SLICES=$v "; T2V_XCD_SLICES=$v python scripts/train_bench.py --iters 12 2>/dev/null | grep "ms/step" | cut -c1-110
done; done | tee gpurun_out/xcd/ab_train.txt
