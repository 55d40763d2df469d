// conv_stem.hip -- the generator's 7x7 stems (ReflectionPad2d(3) + Conv2d(9 | 6 -> 64 | 128, 7) with fused
// instance-norm statistics) on gfx950.  SURVEY.md section 8a row a5.
//
// Why not the implicit-GEMM kernel: with 12 (8) stored input channels a 32-float K stage of the im2col row spans
// 2.7 (4) taps, so its loader decodes a tap per 16-byte chunk (MODE 1: VALU work per stage) and every input pixel
// is re-fetched through L2 49 times; K = 588 is 19 stages only, so the per-tile fixed cost shows: 75 TFLOP/s.
// This kernel is shaped by the data instead (like conv_head.hip, but on the matrix cores):
//   * one block = a 16x16 pixel tile x ALL output channels; its 22x22 input halo is staged ONCE in LDS
//     ([pixel][Cin_s], lane-linear LDS-DMA with the reflection resolved per lane) and read 49 times from there;
//   * for a fixed kernel row kh the 7 taps x Cin_s channels of an output pixel are ONE contiguous run of
//     7*Cin_s floats in that LDS image (NHWC), so an MFMA A fragment (4 consecutive k of one pixel) is a single
//     ds_read_b128 at pixel stride Cin_s*4 bytes = 3 (2) quads: an odd quad stride -> conflict-free for Cin_s 12;
//   * the weights of one kernel row ([Cout][7*Cin_s] slice of the SAME packed [Cout_p][Kp] matrix) are staged per
//     kh with an odd row stride (23 | 15 quads); lanes past the run write zeros (out-of-range DMA lanes), which
//     also absorbs the run's padding to a multiple of 8 floats (the k permutation inside a quad pair is the
//     implicit-GEMM kernel's: identical for A and B);
//   * 8 waves, all MFMA: wave w owns pixel rows 2w, 2w+1 of the tile (32 pixels) x all channels (<= 64
//     accumulator VGPRs); epilogue = bias + two-pass tree-summed (mean, M2) per 256-pixel tile + NHWC store.
// 2 blocks per CU (70 KiB LDS): one block's per-kh weight staging is hidden by the other's MFMAs.
#include <stdlib.h>

#include "t2v_internal.h"

namespace t2v {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x16s __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void st_dma16(const float* base, int nbytes, char* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#endif
}

// CR = real input channels where the kernel knows how to skip the storage padding (9 of 12, 6 of 8: `DENSE`), else CS
template <int CS, int NT, int CR = CS>
struct StemCfg {
    static constexpr int TILE = 16, HALO = TILE + 6, NPIX = HALO * HALO;      // 484 halo pixels
    static constexpr int CQ = CS / 4;                                          // quads per pixel
    static constexpr int RUNQ = 7 * CQ;                                        // quads of one kernel row's run (21 | 14)
    static constexpr int QS = (RUNQ + 1) / 2;                                  // k steps of 8 floats per kernel row (11 | 7)
    static constexpr bool DENSE = (CS == 12 && CR == 9) || (CS == 8 && CR == 6);
    // weight row stride in quads, odd (23 | 15); the dense k order reads one quad pair past the run (tap "7": zeros) -> 25 | 17
    static constexpr int BQ = DENSE ? (CS == 12 ? 25 : 17) : ((2 * QS) | 1);
    static constexpr int N = NT * 32;
    static constexpr int HALO_INSTR = (NPIX * CQ + 63) / 64;                   // DMA wave-instructions for the halo
    static constexpr int HALO_BYTES = HALO_INSTR * 1024 + 256;                 // + slack read by the padded run
    static constexpr int W_INSTR = (N * BQ + 63) / 64;
    static constexpr int W_BYTES = W_INSTR * 1024;
    static constexpr int RED_BYTES = 8 * N * 4;
    static constexpr int LDS = HALO_BYTES + W_BYTES + RED_BYTES;
};

template <int CS, int NT, int CR>
__global__ __launch_bounds__(512) void conv_stem7x7_kernel(const StemParams p) {
    using C = StemCfg<CS, NT, CR>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_halo = smem;
    char* s_w = smem + C::HALO_BYTES;
    float* s_red = reinterpret_cast<float*>(smem + C::HALO_BYTES + C::W_BYTES);   // [8][N]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7

    // XCD-banded tile order (block b runs on XCD b % 8: a band of tile rows per XCD shares halos in its L2)
    int tile;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, idx = b >> 3;
        const int q = nb >> 3, r = nb & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_x = (p.W + C::TILE - 1) / C::TILE;   // ragged right / bottom tiles: clamped gather, masked epilogue
    const int x0 = (tile % tiles_x) * C::TILE, y0 = (tile / tiles_x) * C::TILE;

    // ---- stage the halo (once) and zero its slack ----
    const int x_bytes = p.H * p.W * CS * 4;
#pragma unroll
    for (int i = 0; i < (C::HALO_INSTR + 7) / 8; ++i) {
        const int instr = wave + 8 * i;
        if (instr < C::HALO_INSTR) {
            const int L = instr * 64 + lane;              // chunk index = pixel * CQ + j
            const int pix = L / C::CQ, j = L - pix * C::CQ;
            const int hy = pix / C::HALO, hx = pix - hy * C::HALO;
            int gy = y0 - 3 + hy, gx = x0 - 3 + hx;
            gy = gy < 0 ? -gy : gy;
            gx = gx < 0 ? -gx : gx;
            gy = max(min(gy, 2 * p.H - 2 - gy), 0);   // (past the reflected border: only feeds masked outputs)
            gx = max(min(gx, 2 * p.W - 2 - gx), 0);
            const int voff = pix < C::NPIX ? ((gy * p.W + gx) * CS + 4 * j) * 4 : 0x7fff0000;
            st_dma16(p.x, x_bytes, s_halo + instr * 1024, voff, 0);
        }
    }
    if (tid < 64) reinterpret_cast<float*>(s_halo + C::HALO_INSTR * 1024)[tid] = 0.f;
    // weights of kernel row kh: LDS chunk L = n * BQ + j  <-  packed row n, floats kh*7*CS + 4j .. +3  (j < RUNQ)
    const int w_bytes = C::N * p.Kp * 4;
    int w_voff[(C::W_INSTR + 7) / 8];
#pragma unroll
    for (int i = 0; i < (C::W_INSTR + 7) / 8; ++i) {
        const int L = (wave + 8 * i) * 64 + lane;
        const int n = L / C::BQ, j = L - n * C::BQ;
        w_voff[i] = (n < C::N && j < C::RUNQ) ? (n * p.Kp + 4 * j) * 4 : 0x7fff0000;
    }
    auto stage_w = [&](int kh) {
#pragma unroll
        for (int i = 0; i < (C::W_INSTR + 7) / 8; ++i) {
            const int instr = wave + 8 * i;
            if (instr < C::W_INSTR) st_dma16(p.w, w_bytes, s_w + instr * 1024, w_voff[i], kh * (7 * CS * 4));
        }
    };
    stage_w(0);

    // ---- fragment geometry ----
    const int r = lane & 31, g = lane >> 5;
    const int py = 2 * wave + (r >> 4), px = r & 15;
    const char* a_base = s_halo + ((py * C::HALO + px) * CS) * 4 + g * 16;   // + kh * HALO*CS*4 + q * 32
    const char* b_base = s_w + (r * C::BQ + g) * 16;                          // + t * 32*BQ*16 + q * 32
    f32x16s acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

    for (int kh = 0; kh < 7; ++kh) {
        __syncthreads();   // halo (kh == 0) and this kernel row's weights have landed (the barrier's fence drains vmcnt)
        const char* ar = a_base + kh * (C::HALO * CS * 4);
        if constexpr (CS == 12 && CR == 9) {
            // 9 real channels in 12: a pixel's quads are (c0..3) (c4..7) (c8, 0, 0, 0).  k order of a kernel row: 7 steps
            // "tap kw: quad 0 | quad 1" with 4 MFMAs each, then 4 steps "c8 of tap 2s | c8 of tap 2s+1" with ONE MFMA each
            // (tap 7 = the pixel after the run against zero weights): 32 MFMAs per row and channel tile instead of 44
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const f32x4s a = *reinterpret_cast<const f32x4s*>(ar + kw * 48);
                f32x4s b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const f32x4s*>(b_base + t * (32 * C::BQ * 16) + kw * 48);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[t][e], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const float a1 = *reinterpret_cast<const float*>(ar + g * 32 + s2 * 96 + 32);      // pixel + 2 s2 + g, quad 2
                float b1[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    b1[t] = *reinterpret_cast<const float*>(b_base + g * 32 + t * (32 * C::BQ * 16) + s2 * 96 + 32);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[t], acc[t], 0, 0, 0);
            }
        } else if constexpr (CS == 8 && CR == 6) {
            // 6 real channels in 8: quads (c0..3) (c4, c5, 0, 0).  4 steps "quad 0 of tap 2s | of tap 2s+1" with 4 MFMAs,
            // 4 steps "quad 1 of tap 2s | of tap 2s+1" with 2 (tap 7: zero weights): 24 MFMAs per row instead of 28
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const f32x4s a = *reinterpret_cast<const f32x4s*>(ar + g * 16 + s2 * 64);
                f32x4s b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    b[t] = *reinterpret_cast<const f32x4s*>(b_base + g * 16 + t * (32 * C::BQ * 16) + s2 * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[t][e], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const float2 a2 = *reinterpret_cast<const float2*>(ar + g * 16 + s2 * 64 + 16);
                float2 b2[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    b2[t] = *reinterpret_cast<const float2*>(b_base + g * 16 + t * (32 * C::BQ * 16) + s2 * 64 + 16);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2[t].x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2[t].y, acc[t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < C::QS; ++q) {
                const f32x4s a = *reinterpret_cast<const f32x4s*>(ar + q * 32);
                f32x4s b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const f32x4s*>(b_base + t * (32 * C::BQ * 16) + q * 32);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[t][e], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();   // every wave is done with these weights
        if (kh + 1 < 7) stage_w(kh + 1);
    }

    // ---- epilogue: bias, (mean, M2) of the tile per channel, NHWC store ----
    // accumulator layout: channel = lane & 31 (+ 32 t), pixel row of the wave = (k & 3) + 8 (k >> 2) + 4 g
    float bias_v[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bias_v[t] = p.bias ? p.bias[t * 32 + r] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] += bias_v[t];
    }
    const int vh = min(C::TILE, p.H - y0), vw = min(C::TILE, p.W - x0);   // valid part of a ragged tile
    const float inv_cnt = 1.f / (float)(vh * vw);
    unsigned okmask = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int row = (k & 3) + 8 * (k >> 2) + 4 * g;
        if (2 * wave + (row >> 4) < vh && (row & 15) < vw) okmask |= 1u << k;
    }
    float mean_b[NT];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float d = acc[t][k] - (pass ? mean_b[t] : 0.f);
                v[k] = ((okmask >> k) & 1u) ? (pass ? d * d : d) : 0.f;
            }
#pragma unroll
            for (int w2 = 8; w2 >= 1; w2 >>= 1)
#pragma unroll
                for (int k = 0; k < w2; ++k) v[k] += v[k + w2];
            float sm = v[0];
            sm += __shfl_xor(sm, 32);
            if (g == 0) s_red[wave * C::N + t * 32 + r] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = t * 32 + r;
            const float tot = ((s_red[c] + s_red[C::N + c]) + (s_red[2 * C::N + c] + s_red[3 * C::N + c])) +
                              ((s_red[4 * C::N + c] + s_red[5 * C::N + c]) + (s_red[6 * C::N + c] + s_red[7 * C::N + c]));
            if (pass == 0) {
                mean_b[t] = tot * inv_cnt;
            } else if (wave == 0 && g == 0) {
                reinterpret_cast<float2*>(p.stats)[(size_t)tile * p.Cout + c] = make_float2(mean_b[t], tot);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int row = (k & 3) + 8 * (k >> 2) + 4 * g;          // pixel of this wave, 0..31
        if (!((okmask >> k) & 1u)) continue;
        const int oy = y0 + 2 * wave + (row >> 4), ox = x0 + (row & 15);
        float* dst = p.y + ((size_t)oy * p.W + ox) * p.Cout_s;
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t * 32 + r] = acc[t][k];
    }
}

template <int CS, int NT, int CR>
static int launch_stem(hipStream_t s, const StemParams& p) {
    using C = StemCfg<CS, NT, CR>;
    auto kern = conv_stem7x7_kernel<CS, NT, CR>;
    static bool attr_done = false;
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(((p.H + C::TILE - 1) / C::TILE) * ((p.W + C::TILE - 1) / C::TILE)), dim3(512), C::LDS, s, p);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

bool conv_stem7x7_supported(int H, int W, int Cin_s, int Cout) {
    return H >= 16 && W >= 16 && (Cin_s == 8 || Cin_s == 12) && (Cout == 64 || Cout == 128);
}

int launch_conv_stem7x7(hipStream_t s, const StemParams& p) {
    T2V_REQUIRE(conv_stem7x7_supported(p.H, p.W, p.Cin_s, p.Cout) && p.stats && p.Cout_s >= p.Cout, "stem kernel: unsupported launch");
    // the generator's own stems (9 pose channels in 12, 6 previous-frame channels in 8) skip the storage padding in K
    if (p.Cin_s == 12 && p.Cin == 9) return p.Cout == 128 ? launch_stem<12, 4, 9>(s, p) : launch_stem<12, 2, 9>(s, p);
    if (p.Cin_s == 8 && p.Cin == 6) return p.Cout == 128 ? launch_stem<8, 4, 6>(s, p) : launch_stem<8, 2, 6>(s, p);
    if (p.Cin_s == 12) return p.Cout == 128 ? launch_stem<12, 4, 12>(s, p) : launch_stem<12, 2, 12>(s, p);
    return p.Cout == 128 ? launch_stem<8, 4, 8>(s, p) : launch_stem<8, 2, 8>(s, p);
}

}  // namespace t2v
