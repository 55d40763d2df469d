// polyphase.hip -- polyphase Winograd F(4,2) for the generator's stride-2 3x3 convolutions ("down") and their transposed
// counterparts ConvTranspose2d(3, stride 2, pad 1, output_padding 1) ("up") on gfx950.  SURVEY.md section 8a rows a6 / a8.
//
// Per dimension a stride-2 3x3 conv is a 2-tap correlation on the odd input samples (taps 0 and 2) plus a 1-tap one on the
// even samples (tap 1); the transposed conv is a 2-tap correlation that makes the odd outputs (taps 2 and 0) plus a 1-tap one
// that makes the even outputs.  The 2-tap parts run as Winograd F(4,2) -- 5 products per 4 outputs --, the 1-tap parts as they
// are: 9 "positions" per dimension, 81 per tile (4x4 outputs of a down conv = a 9x9 input patch at stride 8; 8x8 outputs of an
// up conv = a 5x5 input patch at stride 4) against the direct form's 144 multiply-adds per tile and channel pair: 0.5625x the
// MFMA work, and milder in fp32 than F(4x4,3x3) (2.0e-6 against the direct conv's 1.4e-6 on unit-variance data).
//   V[pos][tile][c] = (B d B^T)[pr][pc]        input transform (this file)
//   M[pos][tile][n] = sum_c V[pos][tile][c] U[pos][n][c]     81 batched GEMMs [T x Cin] x [Cin x Cout]: the fixed-grid kernel
//                                                             of the ResnetBlock convs (conv_igemm.hip: wino_gemm_sk_kernel)
//   y = A M A^T + bias, norm statistics partials              output transform (this file)
// with pos = pr * 9 + pc; pr, pc in 0..4 = the F(4,2) positions, 5..8 = the four 1-tap samples.  Matrices: polyphase_consts.h
// (scripts/gen_polyphase_consts.py; checked against torch in fp64 there).  It pays where the transforms (V is 81/64 of a down
// conv's input and M 81/16 of its output; 81/16 and 81/64 for an up conv) are small against the GEMM: the 512<->1024 and
// 256<->512 layers (csrc/capi.hip: polyphase_supported); the wider, shallower maps stay on the implicit-GEMM kernel.
#include "t2v_internal.h"
#include "norm_pool.h"
#include "polyphase_consts.h"

namespace t2v {

namespace {

inline int pp_grid(long n, int block) {
    long g = (n + block - 1) / block;
    if (g > 65535L * 16) g = 65535L * 16;
    return (int)(g < 1 ? 1 : g);
}

template <int K>
__device__ __forceinline__ float pdot(const double (&row)[K], const float (&v)[K]) {
    float acc = 0.f;
    bool first = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (row[k] != 0.0) {   // compile-time after unrolling: x * 0 is not foldable under IEEE rules
            const float t = (float)row[k] * v[k];
            acc = first ? t : acc + t;
            first = false;
        }
    }
    return acc;
}

// B^T of F(4,2) (5 x 5) = rows 0..4 of kBU;  A^T (4 x 5) = columns 0..4 of kAD
struct PpBT {
    double m[5][5];
    constexpr PpBT() : m{} {
        for (int r = 0; r < 5; ++r)
            for (int c = 0; c < 5; ++c) m[r][c] = pp::kBU[r][c];
    }
};
struct PpAT {
    double m[4][5];
    constexpr PpAT() : m{} {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 5; ++c) m[r][c] = pp::kAD[r][c];
    }
};
constexpr PpBT kBT5{};
constexpr PpAT kAT4{};

// the 128-pixel statistics partial of an output-transform block (64 channels x 4 lanes x 32 values): same arithmetic and the
// same partial layout as winograd.hip's block_stats_128, so inorm_finalize pools it with the geometry of wm x wm tiles
__device__ __forceinline__ void pp_block_stats_128(const float (&val)[32], unsigned mask, float (*sh)[64], int tl, int cl, bool ok,
                                                   float2* __restrict__ stats, int N, int n) {
    if (stats == nullptr) return;
    sh[tl][cl] = (float)__popc(mask);
    __syncthreads();
    const float cnt = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
    __syncthreads();
    const float inv_cnt = cnt > 0.f ? 1.f / cnt : 0.f;
    float mean_b = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float dlt = val[i] - mean_b;
            v[i] = ((mask >> i) & 1u) ? (pass ? dlt * dlt : val[i]) : 0.f;
        }
#pragma unroll
        for (int w = 16; w >= 1; w >>= 1)
#pragma unroll
            for (int i = 0; i < w; ++i) v[i] += v[i + w];
        sh[tl][cl] = v[0];
        __syncthreads();
        const float tot = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
        __syncthreads();
        if (pass == 0) {
            mean_b = tot * inv_cnt;
        } else if (tl == 0 && ok) {
            stats[(size_t)blockIdx.x * N + n] = make_float2(mean_b, tot);
        }
    }
}

}  // namespace

// ---- weights: U[pr*9+pc][n][c] = sum_{a,b} G[pr][a] G[pc][b] g[a][b], in fp64, rounded once ---------------------------------------
// UP = false: w is Conv2d's [Cout][Cin][3][3]; UP = true: ConvTranspose2d's [Cin][Cout][3][3]
template <bool UP>
__global__ void polyphase_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Cout_p,
                                        int Cin_s) {
    const long total = (long)Cout_p * Cin_s;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / Cin_s), c = (int)(i - (long)n * Cin_s);
        double g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const size_t at = UP ? (((size_t)c * Cout + n) * 3 + a) * 3 + b : (((size_t)n * Cin + c) * 3 + a) * 3 + b;
                g[a][b] = (n < Cout && c < Cin) ? (double)w[at] : 0.0;
            }
        double t[9][3];
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const double(&G)[9][3] = UP ? pp::kGU : pp::kGD;
                t[p][b] = G[p][0] * g[0][b] + G[p][1] * g[1][b] + G[p][2] * g[2][b];
            }
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const double(&G)[9][3] = UP ? pp::kGU : pp::kGD;
                const double u = t[p][0] * G[q][0] + t[p][1] * G[q][1] + t[p][2] * G[q][2];
                U[(size_t)(p * 9 + q) * Cout_p * Cin_s + (size_t)n * Cin_s + c] = (float)u;
            }
    }
}
int launch_polyphase_weight(hipStream_t s, const float* w, float* U, int Cout, int Cin, int Cout_p, int Cin_s, int up) {
    const int grid = pp_grid((long)Cout_p * Cin_s, 256);
    if (up)
        hipLaunchKernelGGL(polyphase_weight_kernel<true>, dim3(grid), dim3(256), 0, s, w, U, Cout, Cin, Cout_p, Cin_s);
    else
        hipLaunchKernelGGL(polyphase_weight_kernel<false>, dim3(grid), dim3(256), 0, s, w, U, Cout, Cin, Cout_p, Cin_s);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---- input transform --------------------------------------------------------------------------------------------------------------
// one thread = one tile x 2 channels.  DOWN: the tile's 9 x 9 patch starts at input (8 ty - 1, 8 tx - 1) (zero padding 1); its
// even patch indices 0, 2, .., 8 are the odd-phase samples (F(4,2) input, 5 of them), the odd indices 1, 3, 5, 7 the even-phase
// samples (taken as they are).  UP: the 5 x 5 patch starts at input (4 ty, 4 tx) (zeros past the map); all five samples feed
// F(4,2), the first four are also the 1-tap samples.  The four (transformed | plain) x (transformed | plain) sub-blocks are
// done one after the other, so that at most 25 values per channel are live.
// NORM: x is the previous layer's conv output that has not gone through its norm layer yet; the transform applies
// relu((x - mean) * rstd [* gamma + beta]) on the fly -- inorm_apply_kernel's arithmetic in the same order, so the result is
// bit-identical to apply-then-transform -- and that layer's apply pass (a read and a write of the map) is dropped.  The zero
// padding pads the NORMALISED map: samples outside stay exact zeros.
template <bool UP, bool NORM>
__global__ __launch_bounds__(256) void polyphase_input_kernel(const float2* __restrict__ x, float2* __restrict__ V, int H, int W,
                                                             int C2, int TW, int T, int Tt,
                                                             const float2* __restrict__ mean_rstd,
                                                             const float2* __restrict__ gamma,
                                                             const float2* __restrict__ beta, int relu) {
    const long total = (long)Tt * C2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long tile = i / C2;
        const int c2 = (int)(i - tile * C2);
        if (tile >= T) {   // padding tiles: zeros
#pragma unroll 1
            for (int pos = 0; pos < 81; ++pos) V[((long)pos * Tt + tile) * C2 + c2] = make_float2(0.f, 0.f);
            continue;
        }
        const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
        // sample k of the F(4,2) set / of the plain set, per dimension -> input index
        auto wrow = [&](int k, int t) { return UP ? 4 * t + k : 8 * t - 1 + 2 * k; };       // k = 0..4
        auto prow = [&](int k, int t) { return UP ? 4 * t + k : 8 * t + 2 * k; };           // k = 0..3
        float2 mr0 = make_float2(0.f, 1.f), mr1 = make_float2(0.f, 1.f), gm = make_float2(1.f, 1.f), bt = make_float2(0.f, 0.f);
        if (NORM) {
            mr0 = mean_rstd[2 * c2];
            mr1 = mean_rstd[2 * c2 + 1];
            if (gamma) {
                gm = gamma[c2];
                bt = beta[c2];
            }
        }
        auto load = [&](int yy, int xx) {
            if (!((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)) return make_float2(0.f, 0.f);
            float2 v = x[((long)yy * W + xx) * C2 + c2];
            if (NORM) {
                v.x = (v.x - mr0.x) * mr0.y;
                v.y = (v.y - mr1.x) * mr1.y;
                if (gamma) {
                    v.x = v.x * gm.x + bt.x;
                    v.y = v.y * gm.y + bt.y;
                }
                if (relu == 1) {
                    v.x = fmaxf(v.x, 0.f);
                    v.y = fmaxf(v.y, 0.f);
                }
            }
            return v;
        };
        auto store = [&](int pr, int pc, float vx, float vy) { V[((long)(pr * 9 + pc) * Tt + tile) * C2 + c2] = make_float2(vx, vy); };
        // (1) transformed rows x transformed columns: 5 x 5 -> 5 x 5
        {
            float rx[5][5], ry[5][5];
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                float dx[5], dy[5];
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    const float2 v = load(wrow(a, ty), wrow(b, tx));
                    dx[b] = v.x;
                    dy[b] = v.y;
                }
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    rx[a][q] = pdot<5>(kBT5.m[q], dx);
                    ry[a][q] = pdot<5>(kBT5.m[q], dy);
                }
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                float cx[5], cy[5];
#pragma unroll
                for (int a = 0; a < 5; ++a) {
                    cx[a] = rx[a][q];
                    cy[a] = ry[a][q];
                }
#pragma unroll
                for (int p = 0; p < 5; ++p) store(p, q, pdot<5>(kBT5.m[p], cx), pdot<5>(kBT5.m[p], cy));
            }
        }
        // (2) transformed rows x plain columns: per plain column a 5-vector down the rows
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float cx[5], cy[5];
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const float2 v = load(wrow(a, ty), prow(b, tx));
                cx[a] = v.x;
                cy[a] = v.y;
            }
#pragma unroll
            for (int p = 0; p < 5; ++p) store(p, 5 + b, pdot<5>(kBT5.m[p], cx), pdot<5>(kBT5.m[p], cy));
        }
        // (3) plain rows x transformed columns
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float dx[5], dy[5];
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const float2 v = load(prow(a, ty), wrow(b, tx));
                dx[b] = v.x;
                dy[b] = v.y;
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) store(5 + a, q, pdot<5>(kBT5.m[q], dx), pdot<5>(kBT5.m[q], dy));
        }
        // (4) plain x plain: copies
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float2 v = load(prow(a, ty), prow(b, tx));
                store(5 + a, 5 + b, v.x, v.y);
            }
    }
}
// H, W: the INPUT map; tiles: 4x4 outputs of the H/2 x W/2 map (down) | 4x4 inputs (up); Tt = padded tile rows of V.
// mean_rstd != null: x still has to go through its norm layer (relu: 0 | 1 after it)
int launch_polyphase_input(hipStream_t s, const float* x, float* V, int H, int W, int C, int up, int Tt, const float* mean_rstd,
                           const float* gamma, const float* beta, int relu) {
    T2V_REQUIRE((gamma == nullptr) == (beta == nullptr) && (relu == 0 || relu == 1), "polyphase_input: bad norm arguments");
    const int TH = up ? (H + 3) / 4 : (H / 2 + 3) / 4, TW = up ? (W + 3) / 4 : (W / 2 + 3) / 4;
    const int T = TH * TW;
    const int grid = pp_grid((long)Tt * (C / 2), 256);
    auto kern = up ? (mean_rstd ? polyphase_input_kernel<true, true> : polyphase_input_kernel<true, false>)
                   : (mean_rstd ? polyphase_input_kernel<false, true> : polyphase_input_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float2*>(x), reinterpret_cast<float2*>(V), H, W,
                       C / 2, TW, T, Tt, reinterpret_cast<const float2*>(mean_rstd), reinterpret_cast<const float2*>(gamma),
                       reinterpret_cast<const float2*>(beta), relu);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---- output transform, down: y (4 x 4 per tile) = A M A^T + bias; block = 64 channels x 4 tile lanes, 8 tiles = 128 pixels ------
__global__ __launch_bounds__(256) void polyphase_output_down_kernel(const float* __restrict__ Mm, const float* __restrict__ bias,
                                                                    float* __restrict__ y, float2* __restrict__ stats, int Ho,
                                                                    int Wo, int N, int TW, int T, int Tt) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + cl;
    const bool ok = n < N;
    const float bv = (ok && bias) ? bias[n] : 0.f;
    float out[32];
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long tile = (long)blockIdx.x * 8 + tl + 4 * i;
        const bool tv = tile < T;
        float r[4][9];   // r[i2][pc] = sum_pr A[i2][pr] m[pr][pc]
#pragma unroll
        for (int pc = 0; pc < 9; ++pc) {
            float m[9];
#pragma unroll
            for (int pr = 0; pr < 9; ++pr) m[pr] = (ok && tv) ? Mm[((long)(pr * 9 + pc) * Tt + tile) * N + n] : 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) r[i2][pc] = pdot<9>(pp::kAD[i2], m);
        }
        const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const float v = pdot<9>(pp::kAD[j2], r[i2]) + bv;
                out[i * 16 + i2 * 4 + j2] = v;
                const int oy = 4 * ty + i2, ox = 4 * tx + j2;
                if (tv && oy < Ho && ox < Wo) {
                    mask |= 1u << (i * 16 + i2 * 4 + j2);
                    if (ok) y[((long)oy * Wo + ox) * N + n] = v;
                }
            }
    }
    pp_block_stats_128(out, mask, sh, tl, cl, ok, stats, N, n);
}

// ---- output transform, up: y (8 x 8 per tile) = A M A^T + bias; a thread makes HALF a tile (4 output rows x 8 columns = 32
// values), block = 64 channels x 4 lanes = 2 tiles = 128 pixels: the statistics partial of 8 x 8 tiles (wm = 8) ------------------
__global__ __launch_bounds__(256) void polyphase_output_up_kernel(const float* __restrict__ Mm, const float* __restrict__ bias,
                                                                  float* __restrict__ y, float2* __restrict__ stats, int Ho,
                                                                  int Wo, int N, int TW, int T, int Tt) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + cl;
    const bool ok = n < N;
    const float bv = (ok && bias) ? bias[n] : 0.f;
    const long tile = (long)blockIdx.x * 2 + (tl >> 1);
    const int half = tl & 1;                     // output rows 4*half .. 4*half + 3 of the tile
    const bool tv = tile < T;
    // the four output rows of this half: 8h (plain 2h), 8h+1 (F(4,2) output 2h), 8h+2 (plain 2h+1), 8h+3 (F(4,2) output 2h+1)
    float r[4][9];
#pragma unroll
    for (int pc = 0; pc < 9; ++pc) {
        float m[5];
#pragma unroll
        for (int pr = 0; pr < 5; ++pr) m[pr] = (ok && tv) ? Mm[((long)(pr * 9 + pc) * Tt + tile) * N + n] : 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = 2 * half + k;                                  // plain sample / F(4,2) output index 0..3
            r[2 * k][pc] = (ok && tv) ? Mm[((long)((5 + q) * 9 + pc) * Tt + tile) * N + n] : 0.f;
            r[2 * k + 1][pc] = half ? (k ? pdot<5>(kAT4.m[3], m) : pdot<5>(kAT4.m[2], m)) : (k ? pdot<5>(kAT4.m[1], m) : pdot<5>(kAT4.m[0], m));
        }
    }
    const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
    float out[32];
    unsigned mask = 0;
#pragma unroll
    for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) {
            const float v = pdot<9>(pp::kAU[j2], r[i2]) + bv;
            out[i2 * 8 + j2] = v;
            const int oy = 8 * ty + 4 * half + i2, ox = 8 * tx + j2;
            if (tv && oy < Ho && ox < Wo) {
                mask |= 1u << (i2 * 8 + j2);
                if (ok) y[((long)oy * Wo + ox) * N + n] = v;
            }
        }
    pp_block_stats_128(out, mask, sh, tl, cl, ok, stats, N, n);
}
// Ho, Wo: the OUTPUT map
int launch_polyphase_output(hipStream_t s, const float* Mm, const float* bias, float* y, float* stats, int Ho, int Wo, int N,
                            int up, int Tt) {
    const int e = up ? 8 : 4;
    const int TW = (Wo + e - 1) / e, T = ((Ho + e - 1) / e) * TW, Tp = wino_pad_tiles(T);
    if (up)
        hipLaunchKernelGGL(polyphase_output_up_kernel, dim3(Tp / 2, (N + 63) / 64), dim3(256), 0, s, Mm, bias, y,
                           reinterpret_cast<float2*>(stats), Ho, Wo, N, TW, T, Tt);
    else
        hipLaunchKernelGGL(polyphase_output_down_kernel, dim3(Tp / 8, (N + 63) / 64), dim3(256), 0, s, Mm, bias, y,
                           reinterpret_cast<float2*>(stats), Ho, Wo, N, TW, T, Tt);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

}  // namespace t2v
