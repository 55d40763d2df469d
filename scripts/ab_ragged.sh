#!/bin/bash
# Ragged M tiles of the fixed-grid Winograd GEMM against whole (padded) tiles inside whole frames, unprofiled, alternating runs on
# one box: the reference's 512x680 frames (352 tile rows), two 512x320 sequences in lock-step (320), the 16:9 512x448 (224).
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for cfg in "--width 680" "--width 320 --batch 2" "--width 448"; do
    for r in 1 0; do
      echo -n "ragged=$r two-stream $cfg: "; T2V_WINO_GEMM_SK_RAGGED=$r python scripts/frame_prof.py --frames 60 $cfg 2>/dev/null | grep FRAMES
      echo -n "ragged=$r one-stream $cfg: "; T2V_STREAMS=1 T2V_WINO_GEMM_SK_RAGGED=$r python scripts/frame_prof.py --frames 60 $cfg 2>/dev/null | grep FRAMES
    done
  done
done
