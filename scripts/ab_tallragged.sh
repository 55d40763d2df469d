#!/bin/bash
# One-block-per-CU balanced ragged tiles (T2V_WINO_GEMM_SK_RAGGED=2, wino_gemm_skt_kernel) against the two-per-CU ragged form (=1):
# bit equality, the GEMM stage alone, and whole frames unprofiled, alternating runs on one box.
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "fixed_grid_winograd_gemm_equals and tall_ragged" 2>&1 | tail -3
for r in 1 2; do echo "== sk_probe ragged=$r"; SK_PROBE_RAGGED=$r timeout 600 python scripts/sk_probe.py ragged 2>&1 | grep "tile rows"; done
for rep in 1 2 3; do
  for cfg in "--width 680" "--width 320 --batch 2" "--width 448" "--width 912"; do
    for r in 1 2; do
      echo -n "ragged=$r two-stream $cfg: "; T2V_WINO_GEMM_SK_RAGGED=$r python scripts/frame_prof.py --frames 60 $cfg 2>/dev/null | grep FRAMES
      echo -n "ragged=$r one-stream $cfg: "; T2V_STREAMS=1 T2V_WINO_GEMM_SK_RAGGED=$r python scripts/frame_prof.py --frames 60 $cfg 2>/dev/null | grep FRAMES
    done
  done
done
