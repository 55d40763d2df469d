// winograd.hip -- Winograd F(2x2,3x3) transforms for the ResnetBlock convolutions (3x3, stride 1,
// ReflectionPad2d(1), C -> C at the bottleneck: 84 % of the generator's FLOPs).  fp32 throughout.
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d
//
// 16 element-wise products per (tile, cin, cout) instead of 36 multiply-adds: 2.25x less MFMA work.
// The contraction over cin for each of the 16 transform positions xi is a plain GEMM
//   M[xi][tile][n] = sum_c V[xi][tile][c] * U[xi][n][c]
// which runs on the implicit-GEMM kernel unchanged (a 1x1 "conv" over a 16 x T "image" whose weight
// matrix is selected per group of M tiles, conv_igemm.hip `group_*`).  This file holds the three
// memory-bound pieces around it:
//   winograd_weight_kernel : U = G g G^T, packed [16][Cout_p][Cin_s]           (once, at load)
//   winograd_input_kernel  : V = B^T d B with the reflection padding folded into the patch gather
//   winograd_output_kernel : y = A^T M A + bias, plus the instance-norm partial statistics
//                            (mean, M2 per 128 output pixels) that the direct kernel's epilogue emits
// Layouts: V [16][Tp][C], M [16][Tp][N] (channels contiguous) -- both are NHWC tensors of 16*Tp "pixels", so the
// GEMM kernel's loader / epilogue need nothing new.  Any H, W >= 2: the tile grid is ceil(H/2) x ceil(W/2) = T tiles
// (ragged last row / column: the transforms read clamped indices and mask their writes), padded with zero tiles to
// Tp = a multiple of 128 so that every transform position owns whole GEMM tiles.
#include "t2v_internal.h"
#include "norm_pool.h"
#include "winograd_f4_consts.h"

namespace t2v {

static inline int wg_grid(long n, int block) {
    long g = (n + block - 1) / block;
    if (g > 4096) g = 4096;
    return g < 1 ? 1 : (int)g;
}

// U[xi][n][c] = (G g G^T)[xi],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
// adjoint: w is the forward layer's weight [Cin][Cout][3][3]; the tap read is w[c][n][2-a][2-b] (the data gradient's filter)
__global__ void winograd_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Cout_p,
                                       int Cin_s, int adjoint) {
    const long total = (long)Cout_p * Cin_s;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / Cin_s), c = (int)(i - (long)n * Cin_s);
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                g[a][b] = (n < Cout && c < Cin) ? (adjoint ? w[(((size_t)c * Cout + n) * 3 + (2 - a)) * 3 + (2 - b)]
                                                           : w[(((size_t)n * Cin + c) * 3 + a) * 3 + b]) : 0.f;
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]),
                        u3 = t[a][2];
            U[((size_t)(a * 4 + 0) * Cout_p + n) * Cin_s + c] = u0;
            U[((size_t)(a * 4 + 1) * Cout_p + n) * Cin_s + c] = u1;
            U[((size_t)(a * 4 + 2) * Cout_p + n) * Cin_s + c] = u2;
            U[((size_t)(a * 4 + 3) * Cout_p + n) * Cin_s + c] = u3;
        }
    }
}
int launch_winograd_weight(hipStream_t s, const float* w, float* U, int Cout, int Cin, int Cout_p, int Cin_s, int adjoint) {
    hipLaunchKernelGGL(winograd_weight_kernel, dim3(wg_grid((long)Cout_p * Cin_s, 256)), dim3(256), 0, s, w, U, Cout, Cin,
                       Cout_p, Cin_s, adjoint);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// V[xi][tile][c] = (B^T d B)[xi],  d = 4x4 patch at rows 2ty-1.., cols 2tx-1.. (reflection pad 1)
// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].  One thread = one tile x 4 channels (float4).
// pad / reflect: ReflectionPad2d(1) (the forward ResnetBlock conv) or zero padding `pad` in {0,1,2} (pad 2 = the
// data gradient of that conv: a full correlation producing (H+2) x (W+2)).  The tile grid covers the OUTPUT.
__global__ __launch_bounds__(256) void winograd_input_kernel(const float4* __restrict__ x, float4* __restrict__ V, int H,
                                                             int W, int C4, int TW, int T, int Tp, int pad, int reflect) {
    const long total = (long)Tp * C4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long tile = i / C4;
        const int c4 = (int)(i - tile * C4);
        if (tile >= T) {   // padding tiles: zeros (their GEMM rows are never read back)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) V[((long)xi * Tp + tile) * C4 + c4] = z;
            continue;
        }
        const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
        int ry[4], rx[4];
        bool oky[4], okx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int yy = 2 * ty - pad + k, xx = 2 * tx - pad + k;
            oky[k] = reflect || ((unsigned)yy < (unsigned)H);
            okx[k] = reflect || ((unsigned)xx < (unsigned)W);
            yy = yy < 0 ? -yy : yy;
            xx = xx < 0 ? -xx : xx;
            // rows / columns past the reflected border only feed outputs of a ragged tile that are masked
            ry[k] = max(min(yy, 2 * H - 2 - yy), 0);
            rx[k] = max(min(xx, 2 * W - 2 - xx), 0);
        }
        float4 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                d[a][b] = (oky[a] && okx[b]) ? x[((long)ry[a] * W + rx[b]) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        // rows: t = B^T d
        float4 t[4][4];
#define T2V_SUB(o, p, q) o.x = p.x - q.x; o.y = p.y - q.y; o.z = p.z - q.z; o.w = p.w - q.w;
#define T2V_ADD(o, p, q) o.x = p.x + q.x; o.y = p.y + q.y; o.z = p.z + q.z; o.w = p.w + q.w;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            T2V_SUB(t[0][b], d[0][b], d[2][b])
            T2V_ADD(t[1][b], d[1][b], d[2][b])
            T2V_SUB(t[2][b], d[2][b], d[1][b])
            T2V_SUB(t[3][b], d[1][b], d[3][b])
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 v0, v1, v2, v3;
            T2V_SUB(v0, t[a][0], t[a][2])
            T2V_ADD(v1, t[a][1], t[a][2])
            T2V_SUB(v2, t[a][2], t[a][1])
            T2V_SUB(v3, t[a][1], t[a][3])
            V[((long)(a * 4 + 0) * Tp + tile) * C4 + c4] = v0;
            V[((long)(a * 4 + 1) * Tp + tile) * C4 + c4] = v1;
            V[((long)(a * 4 + 2) * Tp + tile) * C4 + c4] = v2;
            V[((long)(a * 4 + 3) * Tp + tile) * C4 + c4] = v3;
        }
    }
}
int launch_winograd_input(hipStream_t s, const float* x, float* V, int H, int W, int C, int pad, int reflect) {
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    const int TW = (Wo + 1) / 2, T = ((Ho + 1) / 2) * TW, Tp = wino_pad_tiles(T);
    hipLaunchKernelGGL(winograd_input_kernel, dim3(wg_grid((long)Tp * (C / 4), 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(V), H, W, C / 4, TW, T, Tp, pad,
                       reflect);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// (mean, M2) of a block's <= 128 valid output pixels per channel: each of the 4 tile lanes holds 32 pixel
// slots, `mask` marks the ones inside the image (all of them except in ragged / padding tiles).  Two passes,
// tree-summed (exact for constant maps over power-of-two counts), written as the partial inorm_finalize
// merges; the partial's pixel count is recomputed there from the geometry.
__device__ __forceinline__ void block_stats_128(const float (&val)[32], unsigned mask, float (*sh)[64], int tl, int cl,
                                                bool ok, float2* __restrict__ stats, int N, int n) {
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

// y = A^T M A + bias,  A^T = [[1,1,1,0],[0,1,-1,-1]];  block = 64 channels x 4 tile lanes, 32 tiles = 128
// output pixels per block => exactly the (mean_b, M2_b) partial per 128 pixels that inorm_finalize merges.
__global__ __launch_bounds__(256) void winograd_output_kernel(const float* __restrict__ Mm, const float* __restrict__ bias,
                                                              float* __restrict__ y, float2* __restrict__ stats, int H,
                                                              int W, int N, int TW, int T, int Tp) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + cl;
    const bool ok = n < N;
    const float bv = (ok && bias) ? bias[n] : 0.f;
    float out[32];
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long tile = (long)blockIdx.x * 32 + tl + 4 * i;
        const bool tv = tile < T;
        float m[4][4];
#pragma unroll
        for (int a = 0; a < 16; ++a) m[a >> 2][a & 3] = (ok && tv) ? Mm[((long)a * Tp + tile) * N + n] : 0.f;
        float r[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            r[0][b] = m[0][b] + m[1][b] + m[2][b];
            r[1][b] = m[1][b] - m[2][b] - m[3][b];
        }
        const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float v = (b == 0 ? r[a][0] + r[a][1] + r[a][2] : r[a][1] - r[a][2] - r[a][3]) + bv;
                out[i * 4 + a * 2 + b] = v;
                const int oy = 2 * ty + a, ox = 2 * tx + b;
                if (tv && oy < H && ox < W) {
                    mask |= 1u << (i * 4 + a * 2 + b);
                    if (ok) y[((long)oy * W + ox) * N + n] = v;
                }
            }
    }
    block_stats_128(out, mask, sh, tl, cl, ok, stats, N, n);
}
int launch_winograd_output(hipStream_t s, const float* Mm, const float* bias, float* y, float* stats, int H, int W, int N) {
    const int TW = (W + 1) / 2, T = ((H + 1) / 2) * TW, Tp = wino_pad_tiles(T);
    hipLaunchKernelGGL(winograd_output_kernel, dim3(Tp / 32, (N + 63) / 64), dim3(256), 0, s, Mm, bias, y,
                       reinterpret_cast<float2*>(stats), H, W, N, TW, T, Tp);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ------------------------------------------------------------------------------------------------
// F(4x4,3x3): 36 products per 16 outputs (2.25 per output; F(2x2) needs 4, the direct conv 9).  6x6 input
// patches at stride 4, interpolation points {0, +-3/4, +-3/2, inf} (winograd_f4_consts.h: the textbook
// {0, +-1, +-2} cost ~2x the fp32 rounding error).  V / M shrink to 36/16 = 2.25x the activation (F(2x2): 4x),
// so the memory-bound transforms get cheaper as well.  Same layouts: U [36][Cout_p][Cin_s], V [36][Tp][C],
// M [36][Tp][N], T = ceil(H/4) * ceil(W/4) tiles padded to Tp (multiple of 128).
template <int K>
__device__ __forceinline__ float cdot(const double (&row)[K], const float (&v)[K]) {
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

__device__ __forceinline__ void winograd4_weight_store(const double (&g)[3][3], float* __restrict__ U, size_t at,
                                                        size_t pos_stride) {
    double t[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) t[a][b] = f4::kG[a][0] * g[0][b] + f4::kG[a][1] * g[1][b] + f4::kG[a][2] * g[2][b];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const double u = t[a][0] * f4::kG[b][0] + t[a][1] * f4::kG[b][1] + t[a][2] * f4::kG[b][2];
            U[(size_t)(a * 6 + b) * pos_stride + at] = (float)u;   // transformed in fp64, rounded once
        }
}
__global__ void winograd4_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin, int Cout_p,
                                        int Cin_s) {
    const long total = (long)Cout_p * Cin_s;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / Cin_s), c = (int)(i - (long)n * Cin_s);
        double g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = (n < Cout && c < Cin) ? (double)w[(((size_t)n * Cin + c) * 3 + a) * 3 + b] : 0.0;
        winograd4_weight_store(g, U, (size_t)n * Cin_s + c, (size_t)Cout_p * Cin_s);
    }
}
// The data-gradient conv's weights from the FORWARD layer's tensor: element (n, c, a, b) = w[c][n][2-a][2-b], w laid out
// [Cin of this conv = forward Cout][Cout of this conv = forward Cin][3][3].  A thread per (n, c) with c fastest would
// gather 36-byte pieces 36 KB apart; instead a block stages the 8 (n) x 32 (c) tile through LDS: per c a contiguous
// run of 8 x 9 floats is read, and the transformed values leave in 128-byte rows.
// FLIP = false: transposition only -- U^T of the forward layer, [36][Cin_f as rows][Cout_f as K], for the data gradient by the
// TRANSPOSED Winograd algorithm (winograd4_dgrad_output_kernel below)
template <bool FLIP>
__global__ __launch_bounds__(256) void winograd4_weight_adjoint_kernel(const float* __restrict__ w, float* __restrict__ U,
                                                                       int Cout, int Cin, int Cout_p, int Cin_s) {
    __shared__ float sh[32][8 * 9 + 1];
    const int n0 = blockIdx.y * 8, c0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * 72; i += 256) {
        const int cl = i / 72, r = i - cl * 72;        // r = nl * 9 + tap
        const int c = c0 + cl, n = n0 + r / 9;
        sh[cl][r] = (c < Cin && n < Cout) ? w[((size_t)c * Cout + n0) * 9 + r] : 0.f;
    }
    __syncthreads();
    const int cl = threadIdx.x & 31, nl = threadIdx.x >> 5;
    const int n = n0 + nl, c = c0 + cl;
    if (n >= Cout_p || c >= Cin_s) return;
    double g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = (double)sh[cl][nl * 9 + (FLIP ? (2 - a) * 3 + (2 - b) : a * 3 + b)];
    winograd4_weight_store(g, U, (size_t)n * Cin_s + c, (size_t)Cout_p * Cin_s);
}
int launch_winograd4_weight(hipStream_t s, const float* w, float* U, int Cout, int Cin, int Cout_p, int Cin_s, int adjoint) {
    if (adjoint == 2)
        hipLaunchKernelGGL(winograd4_weight_adjoint_kernel<false>, dim3((Cin_s + 31) / 32, (Cout_p + 7) / 8), dim3(256), 0, s, w,
                           U, Cout, Cin, Cout_p, Cin_s);
    else if (adjoint)
        hipLaunchKernelGGL(winograd4_weight_adjoint_kernel<true>, dim3((Cin_s + 31) / 32, (Cout_p + 7) / 8), dim3(256), 0, s, w, U,
                           Cout, Cin, Cout_p, Cin_s);
    else
        hipLaunchKernelGGL(winograd4_weight_kernel, dim3(wg_grid((long)Cout_p * Cin_s, 256)), dim3(256), 0, s, w, U, Cout, Cin,
                           Cout_p, Cin_s);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// V[a*6+b][tile][c] = (B^T d B)[a][b]; one thread = one tile x 2 channels (float2; 36 live values each)
// Tt / t0: V is [36][Tt][C] and this image's tiles start at row t0 (a batch of images shares one matrix)
//
// MODE 1 / 2: x is a conv output that has not been normalised yet.  The transform applies the consumer side of the
// norm layer on the fly -- d = [relu]((x - mean) * rstd [* gamma + beta]) [+ res], the arithmetic of
// inorm_apply_kernel in the same order, so the result is bit-identical to apply-then-transform.  MODE 1: with ReLU
// (a ResnetBlock's first norm).  MODE 2: plus the residual, and the tile's own 4x4 pixels of d are written to `xout`
// as well (every pixel belongs to exactly one tile): a ResnetBlock's output, which the next block needs again as
// its residual.  All pointers are distinct buffers (restrict: the side stores must not fence the patch loads).
// img_tiles / last_tiles: rows of V between consecutive images of the launch, and the tiles the LAST image walks
// (its own T plus the zero rows that pad the whole matrix up to Tt).  Two layouts: slots of Tp rows per image (the weight
// gradient's batch workspace: img_tiles = last_tiles = Tp) and packed (a forward batch: image i owns rows [i*T, (i+1)*T), only
// the total is padded -- two 160-tile images are 320 GEMM rows per position, not 2 x 192).
template <int MODE, bool XCD = false>
__global__ __launch_bounds__(256) void winograd4_input_kernel(const float2* __restrict__ x, float2* __restrict__ V, int H,
                                                              int W, int C2, int TW, int T, int img_tiles, int pad, int reflect,
                                                              int Tt, int t0, const float2* __restrict__ mean_rstd,
                                                              const float2* __restrict__ gamma,
                                                              const float2* __restrict__ beta,
                                                              const float2* __restrict__ res, float2* __restrict__ xout,
                                                              long img_stride, int last_tiles) {
    // blockIdx.y = image of a batch (independent sequences advanced in lock-step): its map, residual and side output
    // sit img_stride float2 after the previous image's, its (mean, rstd) table 2*C2 float2 after, and its tiles
    // occupy rows [t0 + image*Tp, ...) of the batch-wide V
    {
        const long im = blockIdx.y;
        x += im * img_stride;
        t0 += (int)im * img_tiles;
        if (MODE) mean_rstd += im * 2 * C2;
        if (MODE == 2) {
            res += im * img_stride;
            xout += im * img_stride;
        }
    }
    const int ntiles = blockIdx.y == gridDim.y - 1 ? last_tiles : img_tiles;
    auto item = [&](const long tile, const int c2) {
        if (tile >= T) {   // padding tiles: zeros
#pragma unroll
            for (int xi = 0; xi < 36; ++xi) V[((long)xi * Tt + t0 + tile) * C2 + c2] = make_float2(0.f, 0.f);
            return;
        }
        const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
        int ry[6], rx[6];
        bool oky[6], okx[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int yy = 4 * ty - pad + k, xx = 4 * tx - pad + k;
            oky[k] = reflect || ((unsigned)yy < (unsigned)H);
            okx[k] = reflect || ((unsigned)xx < (unsigned)W);
            yy = yy < 0 ? -yy : yy;
            xx = xx < 0 ? -xx : xx;
            ry[k] = max(min(yy, 2 * H - 2 - yy), 0);   // past the reflected border: ragged tile, outputs masked
            rx[k] = max(min(xx, 2 * W - 2 - xx), 0);
        }
        float2 mr0, mr1, gm = make_float2(1.f, 1.f), bt = make_float2(0.f, 0.f);
        if (MODE) {
            mr0 = mean_rstd[2 * c2];
            mr1 = mean_rstd[2 * c2 + 1];
            if (gamma) {
                gm = gamma[c2];
                bt = beta[c2];
            }
        }
        // rows first: r[a][j] = sum_b B^T[j][b] d[a][b]
        float rxv[6][6], ryv[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float2 d[6], r[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const long at = ((long)ry[a] * W + rx[b]) * C2 + c2;
                d[b] = (oky[a] && okx[b]) ? x[at] : make_float2(0.f, 0.f);
                if (MODE == 2) r[b] = (oky[a] && okx[b]) ? res[at] : make_float2(0.f, 0.f);
            }
            float dx[6], dy[6];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                float2 v = d[b];
                if (MODE && oky[a] && okx[b]) {
                    v.x = (v.x - mr0.x) * mr0.y;
                    v.y = (v.y - mr1.x) * mr1.y;
                    if (gamma) {
                        v.x = v.x * gm.x + bt.x;
                        v.y = v.y * gm.y + bt.y;
                    }
                    if (MODE == 1) {
                        v.x = fmaxf(v.x, 0.f);
                        v.y = fmaxf(v.y, 0.f);
                    }
                    if (MODE == 2) {
                        v.x += r[b].x;
                        v.y += r[b].y;
                        // the tile's own pixels: rows / columns pad .. pad+3 of the patch, inside the map
                        if (a >= pad && a < pad + 4 && b >= pad && b < pad + 4 && 4 * ty - pad + a < H && 4 * tx - pad + b < W)
                            xout[((long)ry[a] * W + rx[b]) * C2 + c2] = v;
                    }
                }
                dx[b] = v.x;
                dy[b] = v.y;
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                rxv[a][j] = cdot<6>(f4::kBT[j], dx);
                ryv[a][j] = cdot<6>(f4::kBT[j], dy);
            }
        }
        // columns: v[a2][j] = sum_a B^T[a2][a] r[a][j]
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float cx[6], cy[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                cx[a] = rxv[a][j];
                cy[a] = ryv[a][j];
            }
#pragma unroll
            for (int a2 = 0; a2 < 6; ++a2)
                V[((long)(a2 * 6 + j) * Tt + t0 + tile) * C2 + c2] = make_float2(cdot<6>(f4::kBT[a2], cx), cdot<6>(f4::kBT[a2], cy));
        }
    };
    if constexpr (XCD) {
        // XCD x owns the channel slices {x, x + 8, ..} of 64 pairs for ALL tiles (C2 % 512 == 0): the up to four tiles whose
        // 6x6 patches share an input pixel read it through one L2 instead of through the fabric from four XCDs (block b runs on
        // XCD b % 8: observed, relied on for speed only).  A wave = one tile x one slice; the same arithmetic per item
        const long it = (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
        if (it < (long)(C2 >> 9) * ntiles) {
            const int sl = (int)(it / ntiles);
            item(it - (long)sl * ntiles, ((sl << 3) + (int)(blockIdx.x & 7)) * 64 + (int)(threadIdx.x & 63));
        }
    } else {
        const long total = (long)ntiles * C2;
        const long stride = (long)gridDim.x * blockDim.x;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long tile = i / C2;
            item(tile, (int)(i - tile * C2));
        }
    }
}
// grid.x of the XCD form for `tiles` tiles of C2 channel pairs (0: the shape does not take it)
static inline unsigned xcd_slice_grid(long tiles, int C2) {
    if (C2 % 512 || !options().xcd_slices) return 0;
    const long g = (((long)(C2 >> 9) * tiles + 3) / 4) * 8;
    return g <= 0x7fffffffL ? (unsigned)g : 0;
}
// Slot layout (the weight gradient's batch workspace): the image at `x` goes to slot `image` of a V sized for `batch`
// slots of Tp rows.  Packed layout (nimg > 0: a forward batch): the nimg images starting at `x` (img_stride floats apart)
// own T rows each, the total padded to wino_pad_tiles(nimg * T).
int launch_winograd4_input(hipStream_t s, const float* x, float* V, int H, int W, int C, int pad, int reflect, int batch,
                           int image, int nimg, long img_stride) {
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    const int TW = (Wo + 3) / 4, T = ((Ho + 3) / 4) * TW, Tp = wino_pad_tiles(T);
    const bool packed = nimg > 0;
    const int n = packed ? nimg : 1;
    const int Tt = packed ? wino_pad_tiles(n * T) : batch * Tp;
    const int img_tiles = packed ? T : Tp, last_tiles = packed ? Tt - (n - 1) * T : Tp, t0 = packed ? 0 : image * Tp;
    const long most = last_tiles > img_tiles ? last_tiles : img_tiles;
    const unsigned xg = xcd_slice_grid(most, C / 2);
    auto kern = xg ? winograd4_input_kernel<0, true> : winograd4_input_kernel<0, false>;
    hipLaunchKernelGGL(kern, dim3(xg ? xg : wg_grid(most * (C / 2), 256), n),
                       dim3(256), 0, s, reinterpret_cast<const float2*>(x), reinterpret_cast<float2*>(V), H, W, C / 2, TW, T,
                       img_tiles, pad, reflect, Tt, t0, nullptr, nullptr, nullptr, nullptr, nullptr, img_stride / 2, last_tiles);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
// relu_only: 1 = [ReLU](norm(x)); 0 = norm(x) + res with the result also written to xout (both required)
int launch_winograd4_input_lazy(hipStream_t s, const float* x, float* V, int H, int W, int C, int pad, int reflect,
                                const float* mean_rstd, const float* gamma, const float* beta, int relu_only,
                                const float* res, float* xout, int nimg, long img_stride) {
    T2V_REQUIRE(mean_rstd && (gamma == nullptr) == (beta == nullptr), "winograd4_input_lazy: bad norm arguments");
    T2V_REQUIRE(relu_only ? (!res && !xout) : (res && xout), "winograd4_input_lazy: residual and side output go together");
    // with another padding a tile's own 4x4 block would not tile the input map
    T2V_REQUIRE(relu_only || pad == 1, "winograd4_input_lazy: the side output needs pad == 1");
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    const int TW = (Wo + 3) / 4, T = ((Ho + 3) / 4) * TW;
    const int Tt = wino_pad_tiles(nimg * T), last_tiles = Tt - (nimg - 1) * T;      // packed layout
    const unsigned xg = xcd_slice_grid(last_tiles, C / 2);      // (last_tiles >= T: the image with the padding rows walks the most)
    auto kern = xg ? (relu_only ? winograd4_input_kernel<1, true> : winograd4_input_kernel<2, true>)
                   : (relu_only ? winograd4_input_kernel<1, false> : winograd4_input_kernel<2, false>);
    hipLaunchKernelGGL(kern, dim3(xg ? xg : wg_grid((long)last_tiles * (C / 2), 256), nimg), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(x), reinterpret_cast<float2*>(V), H, W, C / 2, TW, T, T, pad, reflect, Tt, 0,
                       reinterpret_cast<const float2*>(mean_rstd), reinterpret_cast<const float2*>(gamma),
                       reinterpret_cast<const float2*>(beta), reinterpret_cast<const float2*>(res),
                       reinterpret_cast<float2*>(xout), img_stride / 2, last_tiles);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// y = A^T M A + bias; block = 64 channels x 4 tile lanes, 8 tiles (2 per thread) = 128 output pixels
// blockIdx.z = image of a batch: its tiles are rows [image*T, (image+1)*T) of the batch-wide M ([36][Tt][N], packed: only
// the total is padded), its map / statistics partials (Tp/8 per image) / (mean, rstd) table follow the previous image's.
__global__ __launch_bounds__(256) void winograd4_output_kernel(const float* __restrict__ Mm, const float* __restrict__ bias,
                                                               float* __restrict__ y, float2* __restrict__ stats, int H,
                                                               int W, int N, int TW, int T, int Tp, int lrelu,
                                                               float slope, int Tt) {
    __shared__ float sh[4][64];
    {
        const long im = blockIdx.z;
        Mm += im * T * N;                        // packed layout: image i owns rows [i*T, (i+1)*T) of every position's [Tt][N] matrix
        y += im * H * W * N;
        if (stats) stats += im * (Tp / 8) * N;
    }
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
        float r[4][6];   // r[i2][b] = sum_a A^T[i2][a] m[a][b]
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            float m[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) m[a] = (ok && tv) ? Mm[((long)(a * 6 + b) * Tt + tile) * N + n] : 0.f;
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) r[i2][b] = cdot<6>(f4::kAT[i2], m);
        }
        const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const float v = cdot<6>(f4::kAT[j2], r[i2]) + bv;
                out[i * 16 + i2 * 4 + j2] = v;
                const int oy = 4 * ty + i2, ox = 4 * tx + j2;
                if (tv && oy < H && ox < W) {
                    mask |= 1u << (i * 16 + i2 * 4 + j2);
                    // (Leaky)ReLU of layers without a norm (the VGG19 loss network); never together with statistics
                    if (ok) y[((long)oy * W + ox) * N + n] = (lrelu && v < 0.f) ? v * slope : v;
                }
            }
    }
    block_stats_128(out, mask, sh, tl, cl, ok, stats, N, n);
}
// nimg images: M is [36][nimg*T padded][N]; y and stats hold the images back to back
int launch_winograd4_output(hipStream_t s, const float* Mm, const float* bias, float* y, float* stats, int H, int W, int N,
                            int lrelu, float slope, int nimg) {
    const int TW = (W + 3) / 4, T = ((H + 3) / 4) * TW, Tp = wino_pad_tiles(T);
    const dim3 grid(Tp / 8, (N + 63) / 64, nimg);
    hipLaunchKernelGGL(winograd4_output_kernel, grid, dim3(256), 0, s, Mm, bias, y, reinterpret_cast<float2*>(stats), H, W, N,
                       TW, T, Tp, lrelu, slope, wino_pad_tiles(nimg * T));
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ------------------------------------------------------------------------------------------------
// Data gradient by the TRANSPOSED algorithm.  Forward: V = B^T d B, M = U V, Y = A^T M A per 6x6 patch d of the padded input.
// Its transpose: dM = A dY A^T (winograd4_dy_kernel -- the tensor the weight gradient needs as well), dV = U^T dM (a batched
// GEMM over the layer's OWN T tiles: the "full correlation" form works on (H+2) x (W+2) outputs, 289 -> 320 tiles for a 64 x 64
// map), dd = B dV B^T, and the 6x6 patches dd -- stride 4, two rows / columns of overlap -- are added up into the gradient of
// the PADDED input, dxp [(H+2)][(W+2)][C] (the reflect-pad adjoint folds it afterwards).  One thread = one 4x4 block of padded
// pixels x 2 channels: it gathers its 16 values from the (up to) four patches that cover them, each patch element computed
// exactly once chip-wide, contributions added in a fixed order (own tile, left, top, top-left).  H % 4 == 0 == W % 4.
struct F4B {   // B = (B^T)^T as a constexpr table: row i = the coefficients of patch row / column i
    double m[6][6];
    constexpr F4B() : m{} {
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) m[c][r] = f4::kBT[r][c];
    }
};
constexpr F4B kF4B{};

template <int DY, int DX>   // source tile = (by - DY, bx - DX); it contributes its patch rows 4*DY.., columns 4*DX..
__device__ __forceinline__ void dgrad_gather_tile(const float2* __restrict__ dV, long tile, int Tp, int C2, int c2,
                                                  float (&ox)[4][4], float (&oy)[4][4]) {
    constexpr int NR = DY ? 2 : 4, NS = DX ? 2 : 4;
    float tx[6][NS], ty[6][NS];      // stage 1: columns of the patch, per transform row a2
#pragma unroll
    for (int a2 = 0; a2 < 6; ++a2) {
        float vx[6], vy[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float2 v = dV[((long)(a2 * 6 + j) * Tp + tile) * C2 + c2];
            vx[j] = v.x;
            vy[j] = v.y;
        }
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
            tx[a2][sidx] = cdot<6>(kF4B.m[4 * DX + sidx], vx);
            ty[a2][sidx] = cdot<6>(kF4B.m[4 * DX + sidx], vy);
        }
    }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
        float cx[6], cy[6];
#pragma unroll
        for (int a2 = 0; a2 < 6; ++a2) {
            cx[a2] = tx[a2][sidx];
            cy[a2] = ty[a2][sidx];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            ox[r][sidx] += cdot<6>(kF4B.m[4 * DY + r], cx);
            oy[r][sidx] += cdot<6>(kF4B.m[4 * DY + r], cy);
        }
    }
}

// One (padded 4x4 block, channel pair) item: gather from the up to four patches that cover it, write the 16 padded pixels.
__device__ __forceinline__ void dgrad_output_item(const float2* __restrict__ dV, float2* __restrict__ dxp, int H, int W, int C2,
                                                  int TH, int TW, int Tp, int by, int bx, int c2) {
    float ox[4][4], oy[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) ox[r][q] = oy[r][q] = 0.f;
    if (by < TH && bx < TW) dgrad_gather_tile<0, 0>(dV, (long)by * TW + bx, Tp, C2, c2, ox, oy);
    if (by < TH && bx >= 1) dgrad_gather_tile<0, 1>(dV, (long)by * TW + bx - 1, Tp, C2, c2, ox, oy);
    if (by >= 1 && bx < TW) dgrad_gather_tile<1, 0>(dV, (long)(by - 1) * TW + bx, Tp, C2, c2, ox, oy);
    if (by >= 1 && bx >= 1) dgrad_gather_tile<1, 1>(dV, (long)(by - 1) * TW + bx - 1, Tp, C2, c2, ox, oy);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int py = 4 * by + r, px = 4 * bx + q;
            if (py < H + 2 && px < W + 2) dxp[((long)py * (W + 2) + px) * C2 + c2] = make_float2(ox[r][q], oy[r][q]);
        }
}
// XCD = true (C2 a multiple of 512, launch_winograd4_dgrad_output): every element of dV is read by the four padded blocks its
// patch overlaps.  With a thread index that runs over (block, channel pair) those four readers are workgroups on four different
// XCDs (block b runs on XCD b % 8 -- observed, relied on for speed only), each pulling the line through the fabric into its own
// L2: 4 x 37.7 MB per 1024-channel 64x64 layer.  Here XCD x owns the channel slices {x, x + 8, ...} of 64 pairs (512 contiguous
// bytes per (position, tile)) for ALL blocks: the four readers of a line sit on one XCD, three of them hit its L2 (reuse distance
// one row of tiles, 0.3 MB per slice).  A wave = one padded block x one slice; the arithmetic per item is the same: same bits.
template <bool XCD>
__global__ __launch_bounds__(256) void winograd4_dgrad_output_kernel(const float2* __restrict__ dV, float2* __restrict__ dxp,
                                                                     int H, int W, int C2, int TH, int TW, int Tp) {
    if constexpr (XCD) {
        const int nblk = (TH + 1) * (TW + 1), spx = C2 >> 9;          // slices per XCD
        const int xcd = blockIdx.x & 7;
        const long item = (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
        if (item >= (long)spx * nblk) return;
        const int sl = (int)(item / nblk), blk = (int)(item - (long)sl * nblk);
        const int by = blk / (TW + 1), bx = blk - by * (TW + 1);
        dgrad_output_item(dV, dxp, H, W, C2, TH, TW, Tp, by, bx, ((sl << 3) + xcd) * 64 + (threadIdx.x & 63));
    } else {
        const long total = (long)(TH + 1) * (TW + 1) * C2;
        const long stride = (long)gridDim.x * blockDim.x;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const long blk = i / C2;
            const int c2 = (int)(i - blk * C2);
            const int by = (int)(blk / (TW + 1)), bx = (int)(blk - (long)by * (TW + 1));
            dgrad_output_item(dV, dxp, H, W, C2, TH, TW, Tp, by, bx, c2);
        }
    }
}
// dV [36][Tp][C] of an H x W map (H, W multiples of 4, pad 1) -> dxp [(H+2)][(W+2)][C]
int launch_winograd4_dgrad_output(hipStream_t s, const float* dV, float* dxp, int H, int W, int C) {
    T2V_REQUIRE(H % 4 == 0 && W % 4 == 0 && C % 2 == 0, "winograd4_dgrad_output: H, W must be multiples of 4");
    const int TH = H / 4, TW = W / 4, Tp = wino_pad_tiles(TH * TW), C2 = C / 2;
    const long per_xcd = ((long)(C2 >> 9) * (TH + 1) * (TW + 1) + 3) / 4;
    if (C2 % 512 == 0 && options().xcd_slices && per_xcd * 8 <= 0x7fffffffL)
        hipLaunchKernelGGL(winograd4_dgrad_output_kernel<true>, dim3((unsigned)(per_xcd * 8)), dim3(256), 0, s,
                           reinterpret_cast<const float2*>(dV), reinterpret_cast<float2*>(dxp), H, W, C2, TH, TW, Tp);
    else
        hipLaunchKernelGGL(winograd4_dgrad_output_kernel<false>, dim3(wg_grid((long)(TH + 1) * (TW + 1) * C2, 256)), dim3(256), 0,
                           s, reinterpret_cast<const float2*>(dV), reinterpret_cast<float2*>(dxp), H, W, C2, TH, TW, Tp);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient in the Winograd domain (F(4x4,3x3)).  Forward: y_tile = A^T [U (.) V] A, U = G g G^T,
// V = B^T d B.  So
//   dU[xi][n][c] = sum_tiles (A dy_tile A^T)[xi][n] * V[xi][tile][c]          (36 reductions over tiles)
//   dg[n][c]     = G^T dU[.][n][c] G
// The reductions over tiles are 36 pixel-reduction GEMMs of 1/4 of the direct weight gradient's FLOPs; they run
// on conv_wgrad_kernel unchanged (the transform position plays the role of the tap: row xi of a [36][T] "image").
// winograd4_dy_kernel: Mdy[a*6+b][t0+tile][n] = (A dy A^T)[a][b], dy tile 4x4 (zero outside the map)
// NORM: `dy` is the gradient behind the layer's norm (+ activation) and `nb.x` the conv's raw output: the gradient in front of
// the norm -- rstd * gamma * (g - S0/N - xhat * S1/N), g = dy * act'(gamma * xhat + beta): the arithmetic of
// inorm_bwd_apply_kernel (elementwise.hip) in the same order, the same bits -- is formed per loaded element and never
// written out (nothing else reads it where the data gradient takes A dy A^T from this workspace)
struct DyNormBackward {
    const float2* x;           // the conv's raw output [Ho][Wo][cs2]
    const float2* mean_rstd;   // [C] (mean, rstd)
    const float* gamma;        // may be null
    const float* beta;
    const float2* sums;        // [C] (S0, S1) of inorm_bwd_final_kernel
    float invn;
    int relu;
};
template <bool NORM>
__global__ __launch_bounds__(256) void winograd4_dy_kernel(const float2* __restrict__ dy, float2* __restrict__ Md, int Ho,
                                                           int Wo, int C2, int cs2, int TW, int T, int Tp, int Tt, int t0,
                                                           const DyNormBackward nb) {
    const long total = (long)Tp * C2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long tile = i / C2;
        const int c2 = (int)(i - tile * C2);
        // (value-initialised: without NORM nothing below reads them -- `front` is only called under `if constexpr (NORM)` --
        // but a lambda capturing indeterminate arrays is one edit away from reading them)
        float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f}, ga[2] = {1.f, 1.f}, be[2] = {0.f, 0.f}, k0[2] = {0.f, 0.f}, k1[2] = {0.f, 0.f};
        if constexpr (NORM) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float2 mr = nb.mean_rstd[2 * c2 + k], sm = nb.sums[2 * c2 + k];
                mean[k] = mr.x;
                rstd[k] = mr.y;
                ga[k] = nb.gamma ? nb.gamma[2 * c2 + k] : 1.f;
                be[k] = nb.beta ? nb.beta[2 * c2 + k] : 0.f;
                k0[k] = sm.x * nb.invn;
                k1[k] = sm.y * nb.invn;
            }
        }
        auto front = [&](float xv, float gv, int k) {     // (inorm_bwd_apply_kernel's `one`)
            const float xh = (xv - mean[k]) * rstd[k];
            const float pre = ga[k] * xh + be[k];
            const float g = gv * (nb.relu == 1 ? (pre > 0.f ? 1.f : 0.f) : (nb.relu == 2 ? (pre > 0.f ? 1.f : 0.2f) : 1.f));
            return rstd[k] * ga[k] * (g - k0[k] - xh * k1[k]);
        };
        if (tile >= T) {
#pragma unroll
            for (int xi = 0; xi < 36; ++xi) Md[((long)xi * Tt + t0 + tile) * C2 + c2] = make_float2(0.f, 0.f);
            continue;
        }
        const int ty = (int)(tile / TW), tx = (int)(tile - (long)ty * TW);
        // rows first: r[i][b] = sum_j dy[i][j] A^T[j][b]
        float rx[4][6], ry[4][6];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            float vx[4], vy[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const int oy = 4 * ty + i2, ox = 4 * tx + j2;
                float2 d = (oy < Ho && ox < Wo) ? dy[((long)oy * Wo + ox) * cs2 + c2] : make_float2(0.f, 0.f);
                if constexpr (NORM) {
                    if (oy < Ho && ox < Wo) {
                        const float2 xv = nb.x[((long)oy * Wo + ox) * cs2 + c2];
                        d = make_float2(front(xv.x, d.x, 0), front(xv.y, d.y, 1));
                    }
                }
                vx[j2] = d.x;
                vy[j2] = d.y;
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                float ax = 0.f, ay = 0.f;
                bool first = true;
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2)
                    if (f4::kAT[j2][b] != 0.0) {
                        const float tx_ = (float)f4::kAT[j2][b] * vx[j2], ty_ = (float)f4::kAT[j2][b] * vy[j2];
                        ax = first ? tx_ : ax + tx_;
                        ay = first ? ty_ : ay + ty_;
                        first = false;
                    }
                rx[i2][b] = ax;
                ry[i2][b] = ay;
            }
        }
        // columns: m[a][b] = sum_i A^T[i][a] r[i][b]
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                float ax = 0.f, ay = 0.f;
                bool first = true;
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
                    if (f4::kAT[i2][a] != 0.0) {
                        const float tx_ = (float)f4::kAT[i2][a] * rx[i2][b], ty_ = (float)f4::kAT[i2][a] * ry[i2][b];
                        ax = first ? tx_ : ax + tx_;
                        ay = first ? ty_ : ay + ty_;
                        first = false;
                    }
                Md[((long)(a * 6 + b) * Tt + t0 + tile) * C2 + c2] = make_float2(ax, ay);
            }
    }
}
int launch_winograd4_dy(hipStream_t s, const float* dy, float* Md, int Ho, int Wo, int N, int dy_cs, int batch, int image) {
    const int TW = (Wo + 3) / 4, T = ((Ho + 3) / 4) * TW, Tp = wino_pad_tiles(T);
    hipLaunchKernelGGL(winograd4_dy_kernel<false>, dim3(wg_grid((long)Tp * (N / 2), 256)), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(dy), reinterpret_cast<float2*>(Md), Ho, Wo, N / 2, dy_cs / 2, TW, T, Tp,
                       batch * Tp, image * Tp, DyNormBackward{});
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
// ... of a gradient that still has to pass the layer's norm backwards (winograd4_dy_kernel<true>); x, dy: [Ho][Wo][N]
int launch_winograd4_dy_norm(hipStream_t s, const float* dy, const float* x, const float* mean_rstd, const float* gamma,
                             const float* beta, int relu, const float* sums, float* Md, int Ho, int Wo, int N, int batch,
                             int image) {
    const int TW = (Wo + 3) / 4, T = ((Ho + 3) / 4) * TW, Tp = wino_pad_tiles(T);
    DyNormBackward nb;
    nb.x = reinterpret_cast<const float2*>(x);
    nb.mean_rstd = reinterpret_cast<const float2*>(mean_rstd);
    nb.gamma = gamma;
    nb.beta = beta;
    nb.sums = reinterpret_cast<const float2*>(sums);
    nb.invn = 1.f / (float)((long)Ho * Wo);
    nb.relu = relu;
    hipLaunchKernelGGL(winograd4_dy_kernel<true>, dim3(wg_grid((long)Tp * (N / 2), 256)), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(dy), reinterpret_cast<float2*>(Md), Ho, Wo, N / 2, N / 2, TW, T, Tp,
                       batch * Tp, image * Tp, nb);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// dg[n][c][i][j] = sum_{a,b} G[a][i] G[b][j] dU[a*6+b][n][c]   (torch layout [Cout][Cin][3][3]); fp64 like the
// forward filter transform
__global__ void winograd4_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Cout, int Cin, int Cout_p,
                                    int Kp, int accumulate) {
    const long total = (long)Cout * Cin;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / Cin), c = (int)(i - (long)n * Cin);
        double t[3][6];   // t[i][b] = sum_a G[a][i] dU[a][b]
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            double u[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) u[a] = (double)dU[((size_t)(a * 6 + b) * Cout_p + n) * Kp + c];
#pragma unroll
            for (int i2 = 0; i2 < 3; ++i2) {
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < 6; ++a) acc += f4::kG[a][i2] * u[a];
                t[i2][b] = acc;
            }
        }
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 3; ++j2) {
                double acc = 0.0;
#pragma unroll
                for (int b = 0; b < 6; ++b) acc += t[i2][b] * f4::kG[b][j2];
                float* dst = dw + ((size_t)n * Cin + c) * 9 + i2 * 3 + j2;
                *dst = accumulate ? *dst + (float)acc : (float)acc;
            }
    }
}
int launch_winograd4_dw(hipStream_t s, const float* dU, float* dw, int Cout, int Cin, int Cout_p, int Kp, int accumulate) {
    hipLaunchKernelGGL(winograd4_dw_kernel, dim3(wg_grid((long)Cout * Cin, 256)), dim3(256), 0, s, dU, dw, Cout, Cin, Cout_p,
                       Kp, accumulate);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

}  // namespace t2v
