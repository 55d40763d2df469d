// norm_pool.h -- device-side pieces of the instance-norm statistics shared by the finalize kernel (elementwise.hip)
// and the producers that finalize in their own last block (winograd.hip, conv_igemm.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace t2v {

constexpr int kFinSlices = 64;       // slices of inorm_finalize_kernel: fixes the summation order of the pooling
constexpr int kTicketMaxParts = 128; // producers pool their own partials up to this many per channel

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

// Publishing a (mean, M2) partial to the block that will pool it inside the SAME launch: an 8-byte write-through
// (sc1) store -- it leaves the writer's L2 for memory, and the reader's sc1 loads below bypass its L1, so no
// agent-scope fence is needed on either side (cdna_hip_programming.md section 6 G16, the sc1 / sc1 pair).
__device__ __forceinline__ void publish_partial(float2* dst, float2 v) {
    unsigned long long bits;
    __builtin_memcpy(&bits, &v, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 read_partial(const float2* src) {
    const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
    float2 v;
    __builtin_memcpy(&v, &bits, 8);
    return v;
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

// The pooling of inorm_finalize_kernel for ONE channel by ONE thread, in that kernel's summation order: slice sl of
// kFinSlices accumulates partials sl, sl + 64, ... in fp64 (count-weighted power sums of mean_b - ref, ref = the first
// partial's mean), then the slices are added up in the order 0, 1, ..., 63.  Bit-identical to the kernel for any
// nparts <= kTicketMaxParts; the loads of 16 slices are issued together.  Returns (mean, rstd).
__device__ __forceinline__ float2 pool_partials_ordered(const float2* stats, int nparts, int C, int c, int mtiles, int BM,
                                                        int M, int wm, int H, int W, float eps) {
    const float ref = read_partial(stats + c).x;
    const bool uniform = wm == 0 && M % BM == 0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, sm2 = 0.0;
    for (int sl0 = 0; sl0 < kFinSlices; sl0 += 16) {
        float2 v[2][16];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int part = sl0 + j + kFinSlices * r;
                v[r][j] = part < nparts ? read_partial(stats + (size_t)part * C + c) : make_float2(0.f, 0.f);
            }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int part = sl0 + j + kFinSlices * r;
                if (part >= nparts) continue;
                const int nb = uniform ? BM : partial_pixels(part, mtiles, BM, M, wm, H, W);
                if (nb == 0) continue;
                const double d = (double)(v[r][j].x - ref), n = (double)nb;
                a0 += n;
                a1 += n * d;
                a2 += n * d * d;
                a3 += (double)v[r][j].y;
            }
            if (sl0 + j == 0) {
                s0 = a0; s1 = a1; s2 = a2; sm2 = a3;
            } else {
                s0 += a0; s1 += a1; s2 += a2; sm2 += a3;
            }
        }
    }
    const double mean_d = s1 / s0;
    double m2 = sm2 + s2 - s1 * mean_d;
    m2 = m2 > 0.0 ? m2 : 0.0;
    const float var = (float)(m2 / s0);
    return make_float2(ref + (float)mean_d, 1.0f / sqrtf(var + eps));
}

}  // namespace t2v
