#!/bin/bash
# Tile / ring sweep of the stride-2 and transposed layers at the reference's own frame geometries (512x320, 512x680) and at
# 512x512: the library's own rule (auto) against forced 128x128 / 64x64 tiles and ring depths 2 / 3.
#   gpurun -- 'bash scripts/ab_s2_tiles.sh > gpurun_out/ab_s2_tiles.txt'
cd "$(dirname "$0")/.."
S320=down128_320,down256_320,down512_320,up1024_320,up512_320,up256_320
S680=down128_680,down256_680,down512_680,up1024_680,up512_680,up256_680
S512=down128,down256,down512,up1024,up512,up256
for tile in -1 0 2; do
  for ring in 0 2 3; do
    echo "== T2V_CONV_TILE=$tile T2V_CONV_RING=$ring"
    T2V_CONV_TILE=$tile T2V_CONV_RING=$ring python scripts/kernel_bench.py --shapes $S320,$S680,$S512 --iters 60 --warmup 120
  done
done
