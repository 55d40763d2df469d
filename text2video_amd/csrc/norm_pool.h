// norm_pool.h -- device-side pieces shared by the norm finalize kernel (elementwise.hip: the geometry of the statistics
// partials) and the weight-gradient kernel's in-launch combine (conv_wgrad.hip: the arrival ticket).
#pragma once
#include <hip/hip_runtime.h>

namespace t2v {

constexpr int kFinSlices = 64;       // slices of inorm_finalize_kernel: fixes the summation order of the pooling

// pixels the partial `part` was computed over.  Conv-kernel partials (wm == 0) cover BM consecutive GEMM rows of a
// phase; the Winograd output transform's partials (wm = 2 | 4) cover 128/wm^2 consecutive wm x wm tiles of the
// ceil(H/wm) x ceil(W/wm) tile grid, ragged at the bottom / right edge and padded with empty tiles at the end;
// wm < 0: square pixel tiles of edge -wm (conv_stem.hip), ragged at the right / bottom.
__device__ __forceinline__ int partial_pixels(int part, int mtiles, int BM, int M, int wm, int H, int W) {
    if (wm < 0) {
        const int e = -wm, TW = (W + e - 1) / e;
        const int pi = part % mtiles, ty = pi / TW, tx = pi - ty * TW;
        return min(e, H - e * ty) * min(e, W - e * tx);
    }
    if (wm == 0) {
        const int mt = part % mtiles;
        return min(BM, M - mt * BM);
    }
    const int TW = (W + wm - 1) / wm, T = ((H + wm - 1) / wm) * TW, tpb = 128 / (wm * wm);
    const int pi = part % mtiles;   // partial index inside its image (mtiles = partials per image)
    if (H % wm == 0 && W % wm == 0)   // no ragged tiles: every tile of the grid is whole
        return wm * wm * max(0, min(tpb, T - pi * tpb));
    int nb = 0;
    for (int t = pi * tpb; t < min((pi + 1) * tpb, T); ++t) {
        const int ty = t / TW, tx = t - ty * TW;
        nb += min(wm, H - wm * ty) * min(wm, W - wm * tx);
    }
    return nb;
}

// After every wave of the block has drained its stores (s_waitcnt vmcnt(0)) and the block has met at a barrier:
// draw a ticket; true in the block that drew the last one of `count`.  `flag` is one int of LDS.
__device__ __forceinline__ bool last_arriver(int* ticket, int count, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == count - 1;
        if (last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

}  // namespace t2v
