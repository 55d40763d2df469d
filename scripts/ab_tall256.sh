#!/bin/bash
# 256x128 tiles on one block per CU (T2V_WINO_GEMM_SK_TALL=2) against the default 128x128 on two, at 512x512: two-stream frames
# (1 and 2 sequences, flow / no flow) and the bench headline, alternating runs on one box.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for cfg in "" "--batch 2" "--noflow"; do
    for t in 1 2; do
      echo -n "tall=$t two-stream 512x512 $cfg: "; T2V_WINO_GEMM_SK_TALL=$t python scripts/frame_prof.py --frames 80 $cfg 2>/dev/null | grep FRAMES
    done
  done
done
for rep in 1 2; do
  for t in 1 2; do
    echo -n "tall=$t bench: "; T2V_WINO_GEMM_SK_TALL=$t python bench.py --cpu-frames 0 --e2e-frames 0 --hires-frames 0 --train-steps 0 --no-cold-start 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); v=d['config']['variants']; print(d['value'], v['noflow_fps'], v['batch2_fps'], v['batch4_fps'], d['roofline']['frac'])"
  done
done
