#!/bin/bash
# direct weight gradient of the 512<->1024 stride-2 / transposed layers against the pixel-range split count (in-kernel combine up to 8)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for s in 0 2 3 4 5 6 7 8; do
  echo -n "splits=$s: "; T2V_WGRAD_SPLITS=$s T2V_WGRAD_COMBINE_MAX=8 python scripts/wgrad_bench.py --shapes down512,up1024 --iters 100 --warmup 300 2>/dev/null | tr '\n' ' '; echo
done
done
