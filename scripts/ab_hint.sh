#!/bin/bash
# the overlap hint (256x128 tiles on one block per CU inside the generator's two-stream frames) against T2V_OVERLAP_HINT=0
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "overlap_hint or one_block_per_cu" 2>&1 | tail -2
for rep in 1 2 3; do
  for cfg in "" "--batch 2" "--noflow" "--size 1024 --scales 2 --frames 20"; do
    for h in 0 1; do
      echo -n "hint=$h two-stream $cfg: "; T2V_OVERLAP_HINT=$h python scripts/frame_prof.py --frames 80 $cfg 2>/dev/null | grep FRAMES
    done
  done
done
