// elementwise.hip -- the HBM-bound kernels of the frame-synthesis path (gfx950).
//   instance-norm finalize/apply(+ReLU+residual), flow-warp compositor, 3x3/s2 average pool,
//   weight repacking and NCHW<->NHWC / uint8 plumbing.  All NHWC, 16-byte accesses where the
//   layout allows, grid-stride with <= 2048 blocks (cdna_hip_programming.md Guideline 11/13).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "t2v_internal.h"
#include "norm_pool.h"

namespace t2v {

static inline long gcd_l(long a, long b) {
    while (b) { const long t = a % b; a = b; b = t; }
    return a;
}
static inline int grid_for(long n, int block) {
    long g = (n + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------------------
// Instance norm.  Semantics: F.instance_norm -> batch_norm(training=True) on [1,B*C,H,W]
// ($SP/torch/nn/functional.py:1258-1301): biased variance, y=(x-mean)/sqrt(var+eps)*gamma+beta.
// Statistics arrive as per-tile (mean_b, M2_b) partials from the conv epilogue; they are pooled as
// count-weighted moments about a reference mean in fp64 (numerically a two-pass variance, no E[x^2]-E[x]^2
// cancellation).
// ---------------------------------------------------------------------------------------------
// block = kFinSlices slices x 16 channels; grid = ceil(C/16).  The high-resolution layers have thousands of
// partials per channel (2048 at 512x512, 8192 at 1024x1024) and only a few blocks' worth of channels, so the loop
// over partials IS the run time: it carries no dependent chain -- each slice accumulates count-weighted power
// sums of (mean_b - ref) in fp64 (ref = the first partial's mean: no cancellation, and a constant map gives
// exactly mean = ref, M2 = 0), the slices are tree-free summed in a fixed order at the end:
//   mean = ref + S1/S0,   M2 = sum M2_b + S2 - S1^2/S0        (the pooled-variance identity Chan's update telescopes)
// Pixel count of partial `part`: conv-kernel partials cover BM consecutive GEMM rows of a phase; the Winograd
// output transform's partials (wm = 2 | 4 > 0) cover 128/wm^2 consecutive wm x wm tiles of the ceil(H/wm) x
// ceil(W/wm) tile grid, ragged at the bottom / right edge and padded with empty tiles at the end.
// BatchNorm2d's running statistics, moved `times` times by the (mean, rstd) just written (bn_running_update_kernel's
// arithmetic, so the fused and the stand-alone update agree bit for bit)
__device__ __forceinline__ void running_update(const RunningUpdate& ru, int c, float2 mr, float eps) {
    if (ru.mean == nullptr) return;
    const float var = fmaxf(1.0f / (mr.y * mr.y) - eps, 0.f);
    const float unbiased = var * (ru.n / (ru.n - 1.0f));
    float rm = ru.mean[c], rv = ru.var[c];
    for (int t = 0; t < ru.times; ++t) {
        rm = (1.0f - ru.momentum) * rm + ru.momentum * mr.x;
        rv = (1.0f - ru.momentum) * rv + ru.momentum * unbiased;
    }
    ru.mean[c] = rm;
    ru.var[c] = rv;
}

__global__ __launch_bounds__(16 * kFinSlices) void inorm_finalize_kernel(const float2* __restrict__ stats, int nparts, int mtiles,
                                                             int BM, int M, int C, float eps,
                                                             float2* __restrict__ mean_rstd, int wm, int H, int W,
                                                             double* __restrict__ scratch, RunningUpdate ru) {
    // gridDim.y > 1 (scratch given): block y pools every gridDim.y-th group of partials and leaves its four sums
    // in scratch[y][c][4]; inorm_finalize_merge_kernel adds the groups up.  One block per 16 channels cannot pull
    // a 1024x1024 layer's 16384 x 64 partials (8 MB) through a single CU in less than ~1 ms.
    __shared__ double sh[4][kFinSlices][17];
    const int cc = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cc;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, sm2 = 0.0;
    float ref = 0.f;
    if (c < C) {
        ref = stats[c].x;
        const bool uniform = wm == 0 && M % BM == 0;   // every partial covers BM pixels: no per-partial count
#pragma unroll 4
        for (int part = blockIdx.y * kFinSlices + sl; part < nparts; part += kFinSlices * gridDim.y) {
            const int nb = uniform ? BM : partial_pixels(part, mtiles, BM, M, wm, H, W);
            if (nb == 0) continue;
            const float2 v = stats[(size_t)part * C + c];
            const double d = (double)(v.x - ref), n = (double)nb;
            s0 += n;
            s1 += n * d;
            s2 += n * d * d;
            sm2 += (double)v.y;
        }
    }
    sh[0][sl][cc] = s0;
    sh[1][sl][cc] = s1;
    sh[2][sl][cc] = s2;
    sh[3][sl][cc] = sm2;
    __syncthreads();
    if (sl == 0 && c < C) {
        for (int s = 1; s < kFinSlices; ++s) {
            s0 += sh[0][s][cc];
            s1 += sh[1][s][cc];
            s2 += sh[2][s][cc];
            sm2 += sh[3][s][cc];
        }
        if (scratch != nullptr) {
            double* o = scratch + ((size_t)blockIdx.y * C + c) * 4;
            o[0] = s0; o[1] = s1; o[2] = s2; o[3] = sm2;
            return;
        }
        const double mean_d = s1 / s0;
        double m2 = sm2 + s2 - s1 * mean_d;
        m2 = m2 > 0.0 ? m2 : 0.0;
        const float var = (float)(m2 / s0);
        const float2 mr = make_float2(ref + (float)mean_d, 1.0f / sqrtf(var + eps));
        mean_rstd[c] = mr;
        running_update(ru, c, mr, eps);
    }
}

__global__ void inorm_finalize_merge_kernel(const float2* __restrict__ stats, const double* __restrict__ scratch, int groups,
                                            int C, float eps, float2* __restrict__ mean_rstd, RunningUpdate ru) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, sm2 = 0.0;
    for (int g = 0; g < groups; ++g) {   // fixed order
        const double* o = scratch + ((size_t)g * C + c) * 4;
        s0 += o[0]; s1 += o[1]; s2 += o[2]; sm2 += o[3];
    }
    const double mean_d = s1 / s0;
    double m2 = sm2 + s2 - s1 * mean_d;
    m2 = m2 > 0.0 ? m2 : 0.0;
    const float2 mr = make_float2(stats[c].x + (float)mean_d, 1.0f / sqrtf((float)(m2 / s0) + eps));
    mean_rstd[c] = mr;
    running_update(ru, c, mr, eps);
}

// groups of partial blocks for a launch: 1 (single kernel) unless a scratch buffer is given and the layer is big
static int finalize_groups(int nparts, int C, const double* scratch) {
    if (scratch == nullptr || nparts < 2048) return 1;
    int g = nparts / 512;                       // >= 8 partials per thread
    const int want = 512 / ((C + 15) / 16);     // ~2 blocks per CU in total
    if (g > want) g = want;
    if (g > kFinalizeMaxGroups) g = kFinalizeMaxGroups;
    return g < 1 ? 1 : g;
}
static int run_finalize(hipStream_t s, const float* stats, int nparts, int mtiles, int BM, int M, int C, float eps,
                        float* mean_rstd, int wm, int H, int W, double* scratch, const RunningUpdate* ru) {
    const int groups = finalize_groups(nparts, C, scratch);
    const RunningUpdate none{nullptr, nullptr, 0.f, 0.f, 0};
    const RunningUpdate upd = ru ? *ru : none;
    hipLaunchKernelGGL(inorm_finalize_kernel, dim3((C + 15) / 16, groups), dim3(16 * kFinSlices), 0, s,
                       reinterpret_cast<const float2*>(stats), nparts, mtiles, BM, M, C, eps,
                       reinterpret_cast<float2*>(mean_rstd), wm, H, W, groups > 1 ? scratch : nullptr, groups > 1 ? none : upd);
    T2V_HIP_CHECK(hipGetLastError());
    if (groups > 1) {
        hipLaunchKernelGGL(inorm_finalize_merge_kernel, dim3((C + 255) / 256), dim3(256), 0, s,
                           reinterpret_cast<const float2*>(stats), scratch, groups, C, eps,
                           reinterpret_cast<float2*>(mean_rstd), upd);
        T2V_HIP_CHECK(hipGetLastError());
    }
    return T2V_OK;
}

int launch_inorm_finalize(hipStream_t s, const float* stats, int nparts, int mtiles, int BM, int M, int C,
                          float eps, float* mean_rstd, double* scratch, const RunningUpdate* ru) {
    return run_finalize(s, stats, nparts, mtiles, BM, M, C, eps, mean_rstd, 0, 0, 0, scratch, ru);
}
// partials of square edge x edge pixel tiles (the stem kernel), `batch` images back to back
int launch_inorm_finalize_tiles(hipStream_t s, const float* stats, int edge, int H, int W, int C, float eps,
                                float* mean_rstd, int batch, double* scratch, const RunningUpdate* ru) {
    const int nparts = ((H + edge - 1) / edge) * ((W + edge - 1) / edge);
    return run_finalize(s, stats, batch * nparts, nparts, edge * edge, H * W, C, eps, mean_rstd, -edge, H, W, scratch, ru);
}

// partials written by the Winograd output transform F(wm x wm, 3x3) of an H x W map
int launch_inorm_finalize_winograd(hipStream_t s, const float* stats, int wm, int H, int W, int C, float eps,
                                   float* mean_rstd, int batch, double* scratch, const RunningUpdate* ru) {
    const int T = ((H + wm - 1) / wm) * ((W + wm - 1) / wm), Tp = wino_pad_tiles(T);
    const int nparts = Tp / (128 / (wm * wm));   // per image; a batch's partial blocks are contiguous
    return run_finalize(s, stats, batch * nparts, nparts, 0, H * W, C, eps, mean_rstd, wm, H, W, scratch, ru);
}

// BatchNorm2d's running statistics in training mode ($SP/torch/nn/modules/batchnorm.py:57-64 -> BatchNormalization_
// updateOutput(train, momentum), THCUNN.h:33-45): running = (1 - momentum) * running + momentum * batch statistic,
// the variance entering with the UNBIASED estimate n/(n-1) * var.  The batch statistics are the (mean, rstd) table the
// norm layer was applied with (biased variance: var = 1/rstd^2 - eps).
__global__ void bn_running_update_kernel(const float2* __restrict__ mean_rstd, float* __restrict__ running_mean,
                                         float* __restrict__ running_var, float n, float momentum, float eps, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float2 mr = mean_rstd[c];
    const float var = fmaxf(1.0f / (mr.y * mr.y) - eps, 0.f);
    const float unbiased = var * (n / (n - 1.0f));
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mr.x;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
}
int launch_bn_running_update(hipStream_t s, const float* mean_rstd, float* running_mean, float* running_var, long n,
                             float momentum, float eps, int C) {
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(mean_rstd), running_mean, running_var, (float)n, momentum, eps, C);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// y = [relu]((x-mean)*rstd*gamma+beta) + res1 + res2 ; float4 over NHWC, C % 4 == 0
__global__ __launch_bounds__(256) void inorm_apply_kernel(const float4* __restrict__ x,
                                                          const float4* __restrict__ mean_rstd,
                                                          const float4* __restrict__ gamma,
                                                          const float4* __restrict__ beta,
                                                          const float4* __restrict__ res1,
                                                          const float4* __restrict__ res2, float4* __restrict__ y,
                                                          long n4, int C4, int relu) {
    {   // blockIdx.y = image of a batch: maps n4 float4 apart, (mean, rstd) tables 2*C4 float4 apart
        const long im = blockIdx.y;
        x += im * n4;
        y += im * n4;
        mean_rstd += im * 2 * C4;
        if (res1) res1 += im * n4;
        if (res2) res2 += im * n4;
    }
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int c4 = (int)(i % C4);
        const float4 a = mean_rstd[2 * c4], b = mean_rstd[2 * c4 + 1];  // (m0,r0,m1,r1) (m2,r2,m3,r3)
        float4 v = x[i];
        v.x = (v.x - a.x) * a.y;
        v.y = (v.y - a.z) * a.w;
        v.z = (v.z - b.x) * b.y;
        v.w = (v.w - b.z) * b.w;
        if (gamma) {
            const float4 gm = gamma[c4], bt = beta[c4];
            v.x = v.x * gm.x + bt.x;
            v.y = v.y * gm.y + bt.y;
            v.z = v.z * gm.z + bt.z;
            v.w = v.w * gm.w + bt.w;
        }
        if (relu == 1) {
            v.x = fmaxf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f);
        } else if (relu == 2) {  // LeakyReLU(0.2)
            v.x = v.x > 0.f ? v.x : 0.2f * v.x;
            v.y = v.y > 0.f ? v.y : 0.2f * v.y;
            v.z = v.z > 0.f ? v.z : 0.2f * v.z;
            v.w = v.w > 0.f ? v.w : 0.2f * v.w;
        }
        if (res1) {
            const float4 r = res1[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (res2) {
            const float4 r = res2[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        y[i] = v;
    }
}

// nimg > 1: a batch of images back to back (x, res1, res2, y npix*C floats apart, mean_rstd 2*C apart), one launch
int launch_inorm_apply(hipStream_t s, const float* x, const float* mean_rstd, const float* gamma,
                       const float* beta, const float* res1, const float* res2, float* y, long npix, int C,
                       int relu, int nimg) {
    T2V_REQUIRE(C % 4 == 0, "inorm_apply: C=%d must be a multiple of 4", C);
    const long n4 = npix * (C / 4);
    hipLaunchKernelGGL(inorm_apply_kernel, dim3(grid_for(n4, 256), nimg), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(mean_rstd),
                       reinterpret_cast<const float4*>(gamma), reinterpret_cast<const float4*>(beta),
                       reinterpret_cast<const float4*>(res1), reinterpret_cast<const float4*>(res2),
                       reinterpret_cast<float4*>(y), n4, C / 4, relu);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

__global__ void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ y,
                           long n4) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 u = a[i];
        const float4 v = b[i];
        u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
        y[i] = u;
    }
}
int launch_add(hipStream_t s, const float* a, const float* b, float* y, long n) {
    T2V_REQUIRE(n % 4 == 0, "add: n must be a multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                       reinterpret_cast<float4*>(y), n / 4);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---------------------------------------------------------------------------------------------
// train-step scalars: deterministic two-level reductions (per-block partial -> one block), Adam
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;  // valid in thread 0
}
template <int OP>  // 0: (x-c)^2 ; 1: |a-b|
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float c, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (OP == 0) {
            const float d = a[i] - c;
            s += d * d;
        } else {
            s += fabsf(a[i] - b[i]);
        }
    }
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ part, int np, float* out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) s += part[i];
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = t;
}
// sum over pixels and the channels [c0, c0+C) of mask[pix] * |a - b|  (b == nullptr: 0; mask == nullptr: 1):
// MaskedL1Loss of the flow / warp / weight losses, numerator only
__global__ __launch_bounds__(256) void reduce_masked_l1_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ mask, long npix, int c0, int C,
                                                               int cs, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    const long n = npix * cs, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long pix = i / cs;
        if ((unsigned)((int)(i - pix * cs) - c0) < (unsigned)C) {
            const float d = fabsf(a[i] - (b ? b[i] : 0.f));
            s += mask ? mask[pix] * d : d;
        }
    }
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void masked_l1_backward_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                          const float* __restrict__ mask, float scale, long npix, int c0, int C,
                                          int cs, float* __restrict__ da) {
    const long n = npix * cs, stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long pix = i / cs;
        float g = 0.f;
        if ((unsigned)((int)(i - pix * cs) - c0) < (unsigned)C) {
            const float d = a[i] - (b ? b[i] : 0.f);
            g = (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) * (mask ? mask[pix] : 1.f);
        }
        da[i] = g;
    }
}
int launch_masked_l1(hipStream_t s, const float* a, const float* b, const float* mask, long npix, int c0, int C, int cs,
                     float* scratch, float* out) {
    const int g = grid_for(npix * cs, 256);
    hipLaunchKernelGGL(reduce_masked_l1_kernel, dim3(g), dim3(256), 0, s, a, b, mask, npix, c0, C, cs, scratch);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, s, scratch, g, out);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
int launch_masked_l1_backward(hipStream_t s, const float* a, const float* b, const float* mask, float scale, long npix,
                              int c0, int C, int cs, float* da) {
    hipLaunchKernelGGL(masked_l1_backward_kernel, dim3(grid_for(npix * cs, 256)), dim3(256), 0, s, a, b, mask, scale,
                       npix, c0, C, cs, da);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
int launch_reduce(hipStream_t s, int op, const float* a, const float* b, float c, long n, float* scratch, float* out) {
    const int g = grid_for(n, 256);
    if (op == 0)
        hipLaunchKernelGGL(reduce_partial_kernel<0>, dim3(g), dim3(256), 0, s, a, b, c, n, scratch);
    else
        hipLaunchKernelGGL(reduce_partial_kernel<1>, dim3(g), dim3(256), 0, s, a, b, c, n, scratch);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, s, scratch, g, out);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---- all scalar loss terms of a train step in ONE launch (+ one per-term final pass) ------------------------------------------
// The step has ~65 such terms (LSGAN MSE on the logits of every discriminator scale and pass, feature-matching L1 on every
// intermediate feature map): run one by one they were ~65 x (partial + final reduction) forward, ~65 gradient kernels
// backward and ~140 one-element ATen multiplies / adds / divides around them -- 1-2 ms of 3-5 us launches in a 90 ms step.
// Here a term is a row of three small device tables; block b works on `chunk` consecutive elements of term chunk_term[b]
// and, in the same pass over the operands, writes the term's gradient seed (its weight is a host number, so the gradient
// needs no upstream scalar):
//   op 0  MSE against a constant on channel 0 of an [n][cs] logit tensor: sum (x - c)^2, seed = 2 s (x - c) | 0 in the pad channels
//   op 1  L1: sum |a - b|, seed = s sign(a - b)
//   op 2  no value: the seed rows are exact zeros (passes a loss does not reach)
// The final kernel sums a term's partials in chunk order (deterministic, no atomics) and applies the term's value scale.
__global__ __launch_bounds__(256) void loss_terms_kernel(const long long* __restrict__ tp, const int* __restrict__ ti,
                                                         const float* __restrict__ tf, const int* __restrict__ chunk_term,
                                                         const long long* __restrict__ chunk_off, int chunk,
                                                         float* __restrict__ part) {
    __shared__ float sh[4];
    const int t = chunk_term[blockIdx.x];
    const long long off = chunk_off[blockIdx.x];
    const float* __restrict__ a = reinterpret_cast<const float*>(tp[4 * t]);
    const float* __restrict__ b = reinterpret_cast<const float*>(tp[4 * t + 1]);
    float* __restrict__ seed = reinterpret_cast<float*>(tp[4 * t + 2]);
    const long long end = min(tp[4 * t + 3], off + (long long)chunk);
    const int op = ti[2 * t], cs = ti[2 * t + 1];
    const float c = tf[3 * t], ss = tf[3 * t + 1];
    float s = 0.f;
    if (op == 0) {
        for (long long i = off + threadIdx.x; i < end; i += blockDim.x) {
            const float d = a[i * cs] - c;
            s += d * d;
            if (seed) {
                seed[i * cs] = 2.f * ss * d;
                for (int k = 1; k < cs; ++k) seed[i * cs + k] = 0.f;
            }
        }
    } else if (op == 1) {
        // 16-byte lanes over the aligned body (chunk offsets are multiples of 4; torch allocations are 256-byte aligned and
        // the slices taken of them start at whole images), scalars for a ragged tail
        long long body = off;
        if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)seed) & 15) == 0) {
            const long long n4 = (end - off) >> 2;
            const float4* a4 = reinterpret_cast<const float4*>(a + off);
            const float4* b4 = reinterpret_cast<const float4*>(b + off);
            float4* s4 = seed ? reinterpret_cast<float4*>(seed + off) : nullptr;
            auto sg = [&](float d) { return d > 0.f ? ss : (d < 0.f ? -ss : 0.f); };
            for (long long i = threadIdx.x; i < n4; i += blockDim.x) {
                const float4 av = a4[i], bv = b4[i];
                const float d0 = av.x - bv.x, d1 = av.y - bv.y, d2 = av.z - bv.z, d3 = av.w - bv.w;
                s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
                if (s4) s4[i] = make_float4(sg(d0), sg(d1), sg(d2), sg(d3));
            }
            body = off + (n4 << 2);
        }
        for (long long i = body + threadIdx.x; i < end; i += blockDim.x) {
            const float d = a[i] - b[i];
            s += fabsf(d);
            if (seed) seed[i] = d > 0.f ? ss : (d < 0.f ? -ss : 0.f);
        }
    } else if (seed) {
        for (long long i = off + threadIdx.x; i < end; i += blockDim.x) seed[i] = 0.f;
    }
    const float tot = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
__global__ __launch_bounds__(256) void loss_terms_final_kernel(const float* __restrict__ part, const int* __restrict__ term_chunk0,
                                                               const float* __restrict__ tf, float* __restrict__ out) {
    __shared__ float sh[4];
    const int t = blockIdx.x;
    float s = 0.f;
    for (int i = term_chunk0[t] + threadIdx.x; i < term_chunk0[t + 1]; i += blockDim.x) s += part[i];
    const float tot = block_sum(s, sh);
    if (threadIdx.x == 0) out[t] = tot * tf[3 * t + 2];
}
int launch_loss_terms(hipStream_t s, const long long* tp, const int* ti, const float* tf, const int* chunk_term,
                      const long long* chunk_off, const int* term_chunk0, int nterms, int nchunks, int chunk, float* part,
                      float* out) {
    hipLaunchKernelGGL(loss_terms_kernel, dim3(nchunks), dim3(256), 0, s, tp, ti, tf, chunk_term, chunk_off, chunk, part);
    hipLaunchKernelGGL(loss_terms_final_kernel, dim3(nterms), dim3(256), 0, s, part, term_chunk0, tf, out);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// torch.optim.Adam.step ($SP/torch/optim/adam.py:86-98), same operation order
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float b1, float b2, float omb1, float omb2, float eps,
                            float step_size) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = m[i] * b1 + omb1 * gi;
        const float vi = v[i] * b2 + omb2 * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) + eps));
    }
}
int launch_adam(hipStream_t s, float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                double eps, int step) {
    // adam.py:86-96 evaluates 1 - beta, the bias corrections and step_size in Python doubles and hands each to a
    // float op once: the betas arrive here as doubles for the same reason (0.999f != 0.999)
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const float step_size = (float)(lr * sqrt(bc2) / bc1);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, p, g, m, v, n, (float)b1, (float)b2,
                       (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, step_size);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// The same update for MANY tensors in one launch (a generator has ~250 parameter tensors: one launch each costs more
// than the arithmetic).  Block b works on chunk b: `chunk` consecutive elements at chunk_off[b] of tensor
// chunk_tensor[b]; ptrs[t] = (param, grad, exp_avg, exp_avg_sq) device addresses; step_size per tensor (every
// parameter keeps its own step count).
__global__ __launch_bounds__(256) void adam_multi_kernel(const long long* __restrict__ ptrs, const long long* __restrict__ nelem,
                                                         const float* __restrict__ step_size,
                                                         const int* __restrict__ chunk_tensor,
                                                         const long long* __restrict__ chunk_off, int chunk, float b1,
                                                         float b2, float omb1, float omb2, float eps) {
    const int t = chunk_tensor[blockIdx.x];
    const long long off = chunk_off[blockIdx.x];
    float* __restrict__ p = reinterpret_cast<float*>(ptrs[4 * t]);
    const float* __restrict__ g = reinterpret_cast<const float*>(ptrs[4 * t + 1]);
    if (g == nullptr) return;          // no gradient this step: parameter and moments stay as they are
    float* __restrict__ m = reinterpret_cast<float*>(ptrs[4 * t + 2]);
    float* __restrict__ v = reinterpret_cast<float*>(ptrs[4 * t + 3]);
    const float ss = step_size[t];
    const long long end = min(nelem[t], off + (long long)chunk);
    auto upd = [&](float gi, float& mi, float& vi, float& pi) {
        mi = mi * b1 + omb1 * gi;
        vi = vi * b2 + omb2 * gi * gi;
        pi = pi - ss * (mi / (sqrtf(vi) + eps));
    };
    // 16-byte lanes over the aligned body (torch allocations are 256-byte aligned and the chunk size is a multiple of
    // 4: only a view with an odd storage offset falls back to scalars), scalars for the tail
    long long body = off;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && (off & 3) == 0) {
        const long long n4 = (end - off) >> 2;
        float4* p4 = reinterpret_cast<float4*>(p + off);
        const float4* g4 = reinterpret_cast<const float4*>(g + off);
        float4* m4 = reinterpret_cast<float4*>(m + off);
        float4* v4 = reinterpret_cast<float4*>(v + off);
        for (long long i = threadIdx.x; i < n4; i += blockDim.x) {
            const float4 gv = g4[i];
            float4 mv = m4[i], vv = v4[i], pv = p4[i];
            upd(gv.x, mv.x, vv.x, pv.x);
            upd(gv.y, mv.y, vv.y, pv.y);
            upd(gv.z, mv.z, vv.z, pv.z);
            upd(gv.w, mv.w, vv.w, pv.w);
            m4[i] = mv;
            v4[i] = vv;
            p4[i] = pv;
        }
        body = off + (n4 << 2);
    }
    for (long long i = body + threadIdx.x; i < end; i += blockDim.x) {
        float mi = m[i], vi = v[i], pi = p[i];
        upd(g[i], mi, vi, pi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
}
int launch_adam_multi(hipStream_t s, const long long* ptrs, const long long* nelem, const float* step_size,
                      const int* chunk_tensor, const long long* chunk_off, int nchunks, int chunk, double b1, double b2,
                      double eps) {
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, s, ptrs, nelem, step_size, chunk_tensor, chunk_off,
                       chunk, (float)b1, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---------------------------------------------------------------------------------------------
// weight repacking: torch layouts ($SP/torch/nn/modules/conv.py:28-33) -> [Cout_p][Kp], K =
// tap*Cin_s + c, zero padded.  One-time, at checkpoint load.
// ---------------------------------------------------------------------------------------------
// adjoint != 0: `w` is the FORWARD layer's weight [Cin][Cout][KH][KW] as seen from the data-gradient conv being packed
// (its Cout = the forward Cin): element (n, c, kh, kw) = w[c][n][KH-1-kh][KW-1-kw] -- flip + transpose folded into the gather
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                        int KH, int KW, int Cin_s, int Kp, long total, int adjoint) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
        const int tap = k / Cin_s, c = k - tap * Cin_s;
        float v = 0.f;
        if (n < Cout && tap < KH * KW && c < Cin) {
            const int kh = tap / KW, kw = tap - kh * KW;
            v = adjoint ? w[(((size_t)c * Cout + n) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)]
                        : w[(((size_t)n * Cin + c) * KH + kh) * KW + kw];
        }
        out[i] = v;
    }
}
int launch_pack_conv_weight(hipStream_t s, const float* w, float* packed, int Cout, int Cin, int KH, int KW,
                            int Cin_s, int Kp, int Cout_p, int adjoint) {
    const long total = (long)Cout_p * Kp;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, w, packed, Cout, Cin,
                       KH, KW, Cin_s, Kp, total, adjoint);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ConvTranspose2d(k, stride 2, padding p), weight [Cin][Cout][k][k]:  out[2i+a][2j+b] = sum over the
// taps of phase (a,b).  y = 2i - p + kh  =>  kh has the parity of (a + p); input row i = i' + dy with
// dy = (a + p - kh) / 2 for output row y = 2i' + a.   (k3 p1: a=0 -> kh=1 dy 0 ; a=1 -> kh=0 dy +1, kh=2 dy 0.)
// Phase order (heaviest first for k=3): (1,1), (1,0), (0,1), (0,0) -- shared by packer, unpacker and
// the launch planner through this one function.
__device__ __host__ inline void convT_phase_taps(int phase, int k, int pad, int* ntaps, int kh[4], int kw[4], int dy[4],
                                                 int dx[4], int* a, int* b) {
    const int pa[4] = {1, 1, 0, 0}, pb[4] = {1, 0, 1, 0};
    *a = pa[phase];
    *b = pb[phase];
    int ykh[2], ydy[2], ny = 0, xkw[2], xdx[2], nx = 0;
    for (int q = k - 1; q >= 0; --q) {   // descending kh: dy ascending (matches the k3 order 2,0)
        if (((*a + pad - q) & 1) == 0 && ny < 2) { ykh[ny] = q; ydy[ny] = (*a + pad - q) / 2; ++ny; }
        if (((*b + pad - q) & 1) == 0 && nx < 2) { xkw[nx] = q; xdx[nx] = (*b + pad - q) / 2; ++nx; }
    }
    int n = 0;
    for (int iy = 0; iy < ny; ++iy)
        for (int ix = 0; ix < nx; ++ix) {
            kh[n] = ykh[iy]; dy[n] = ydy[iy];
            kw[n] = xkw[ix]; dx[n] = xdx[ix];
            ++n;
        }
    *ntaps = n;
}

__global__ void pack_convT_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout,
                                         int Cout_p, int Cin_s, int K, int pad) {
    // one launch per phase via blockIdx.y
    const int phase = blockIdx.y;
    int ntaps, kh[4], kw[4], dy[4], dx[4], a, b;
    convT_phase_taps(phase, K, pad, &ntaps, kh, kw, dy, dx, &a, &b);
    long off = 0;
    for (int q = 0; q < phase; ++q) {
        int nt, t1[4], t2[4], t3[4], t4[4], aa, bb;
        convT_phase_taps(q, K, pad, &nt, t1, t2, t3, t4, &aa, &bb);
        const int Kq = (nt * Cin_s + kBK - 1) / kBK * kBK;
        off += (long)Cout_p * Kq;
    }
    const int Kp = (ntaps * Cin_s + kBK - 1) / kBK * kBK;
    const long total = (long)Cout_p * Kp;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
        const int tap = k / Cin_s, c = k - tap * Cin_s;
        float v = 0.f;
        if (n < Cout && tap < ntaps && c < Cin) v = w[(((size_t)c * Cout + n) * K + kh[tap]) * K + kw[tap]];
        out[off + i] = v;
    }
}
void convT_phase_taps_host(int phase, int k, int pad, int* ntaps, int kh[4], int kw[4], int dy[4], int dx[4], int* a,
                           int* b) {
    convT_phase_taps(phase, k, pad, ntaps, kh, kw, dy, dx, a, b);
}
int launch_pack_convT_weight(hipStream_t s, const float* w, float* packed, int Cin, int Cout, int Cin_s, int Cout_p,
                             int K, int pad) {
    const long total = (long)Cout_p * 4 * Cin_s;   // per phase: at most 4 taps
    hipLaunchKernelGGL(pack_convT_weight_kernel, dim3(grid_for(total, 256), 4), dim3(256), 0, s, w, packed, Cin,
                       Cout, Cout_p, Cin_s, K, pad);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---------------------------------------------------------------------------------------------
// backward of the pointwise / norm / padding / pooling ops (train step)
// ---------------------------------------------------------------------------------------------
// xp[b][H+2p][W+2p][C] = pad(x[b]): reflection (no edge repeat) or zeros; float4 over channels
__global__ __launch_bounds__(256) void pad_copy_kernel(const float4* __restrict__ x, float4* __restrict__ xp, int batch, int H,
                                                       int W, int C4, int pad, int reflect) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long total = (long)batch * Hp * Wp * C4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c4 = (int)(i % C4);
        long pix = i / C4;
        const int qx = (int)(pix % Wp);
        pix /= Wp;
        const int qy = (int)(pix % Hp), b = (int)(pix / Hp);
        int iy = qy - pad, ix = qx - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (reflect) {
            iy = iy < 0 ? -iy : iy;
            ix = ix < 0 ? -ix : ix;
            iy = min(iy, 2 * H - 2 - iy);
            ix = min(ix, 2 * W - 2 - ix);
            v = x[(((long)b * H + iy) * W + ix) * C4 + c4];
        } else if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
            v = x[(((long)b * H + iy) * W + ix) * C4 + c4];
        }
        xp[i] = v;
    }
}
int launch_pad_copy(hipStream_t s, const float* x, float* xp, int batch, int H, int W, int C, int pad, int reflect) {
    const long n4 = (long)batch * (H + 2 * pad) * (W + 2 * pad) * (C / 4);
    hipLaunchKernelGGL(pad_copy_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(xp), batch, H, W, C / 4, pad, reflect);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// adjoint of ReflectionPad2d(p): dx[i][j] = sum of dxp over every padded position that mirrors to (i,j)
// VT = float4 when C % 4 == 0 (every layer of the generator), float otherwise; same order of additions either way
template <class VT>
__global__ void reflect_pad_backward_kernel(const VT* __restrict__ dxp, VT* __restrict__ dx, int H, int W, int Cv, int p) {
    const int Wp = W + 2 * p;
    const long total = (long)H * W * Cv;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % Cv);
        const long pix = i / Cv;
        const int x = (int)(pix % W), y = (int)(pix / W);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y >= H - 1 - p && y <= H - 2) ys[ny++] = 2 * (H - 1) + p - y;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x >= W - 1 - p && x <= W - 2) xs[nx++] = 2 * (W - 1) + p - x;
        VT s = dxp[((long)ys[0] * Wp + xs[0]) * Cv + c];   // 0 + v == v: the first term starts the sum
        for (int a = 0; a < ny; ++a)
            for (int b = (a == 0 ? 1 : 0); b < nx; ++b) s += dxp[((long)ys[a] * Wp + xs[b]) * Cv + c];
        dx[i] = s;
    }
}
int launch_reflect_pad_backward(hipStream_t s, const float* dxp, float* dx, int H, int W, int C, int p) {
    if (C % 4 == 0)
        hipLaunchKernelGGL(reflect_pad_backward_kernel<float4>, dim3(grid_for((long)H * W * (C / 4), 256)), dim3(256), 0, s,
                           reinterpret_cast<const float4*>(dxp), reinterpret_cast<float4*>(dx), H, W, C / 4, p);
    else
        hipLaunchKernelGGL(reflect_pad_backward_kernel<float>, dim3(grid_for((long)H * W * C, 256)), dim3(256), 0, s, dxp, dx,
                           H, W, C, p);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// norm backward, stage 1: per channel  S0 = sum g,  S1 = sum g*xhat   with g = dy * act'(gamma*xhat+beta)
// block = 64 channels x 4 pixel lanes; grid = (C/64, slices); partial[slice][c] (float2), then a final pass.
__device__ __forceinline__ float act_grad(float pre, int relu) {
    return relu == 1 ? (pre > 0.f ? 1.f : 0.f) : (relu == 2 ? (pre > 0.f ? 1.f : 0.2f) : 1.f);
}
__global__ __launch_bounds__(256) void inorm_bwd_reduce_kernel(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                               const float2* __restrict__ mean_rstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int relu, long npix, int C,
                                                               float2* __restrict__ partial) {
    // block = 16 channel quads (64 channels, float4 loads: 256 contiguous bytes per pixel) x 16 pixel lanes
    __shared__ float sh[2][16][64];
    const int ql = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + ql * 4;
    const int C4 = C >> 2;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        float mean[4], rstd[4], ga[4], be[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float2 mr = mean_rstd[c0 + k];
            mean[k] = mr.x;
            rstd[k] = mr.y;
            ga[k] = gamma ? gamma[c0 + k] : 1.f;
            be[k] = beta ? beta[c0 + k] : 0.f;
        }
        auto add = [&](const float4 xv, const float4 gv) {
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (xs[k] - mean[k]) * rstd[k];
                const float g = gs[k] * act_grad(ga[k] * xh + be[k], relu);
                s0[k] += g;
                s1[k] += g * xh;
            }
        };
        // four pixels' loads in flight per thread, added in pixel order (the sums do not depend on the unrolling)
        const long step = (long)gridDim.y * 16, q = c0 >> 2;
        long p = (long)blockIdx.y * 16 + sl;
        for (; p + 3 * step < npix; p += 4 * step) {
            float4 xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = x[(p + u * step) * C4 + q];
                gv[u] = dy[(p + u * step) * C4 + q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add(xv[u], gv[u]);
        }
        for (; p < npix; p += step) add(x[p * C4 + q], dy[p * C4 + q]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[0][sl][ql * 4 + k] = s0[k];
        sh[1][sl][ql * 4 + k] = s1[k];
    }
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        const int cl = threadIdx.x;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {   // fixed order
            a0 += sh[0][i][cl];
            a1 += sh[1][i][cl];
        }
        partial[(size_t)blockIdx.y * C + blockIdx.x * 64 + cl] = make_float2(a0, a1);
    }
}
// block = 64 channels x 4 slice lanes (fixed summation order: lane-strided partial sums, then the 4 lanes)
// d_beta / d_gamma (optional): the two sums also go straight into the affine parameters' gradient slices -- written when
// `overwrite`, added otherwise (what a separate unzip2 pass did)
__global__ __launch_bounds__(256) void inorm_bwd_final_kernel(const float2* __restrict__ partial, int slices, int C,
                                                              float2* __restrict__ sums, float* __restrict__ d_beta,
                                                              float* __restrict__ d_gamma, int overwrite) {
    __shared__ float sh[2][4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < C)
        for (int i = sl; i < slices; i += 4) {
            const float2 v = partial[(size_t)i * C + c];
            s0 += v.x;
            s1 += v.y;
        }
    sh[0][sl][cl] = s0;
    sh[1][sl][cl] = s1;
    __syncthreads();
    if (sl == 0 && c < C) {   // (dbeta, dgamma) of an affine norm
        const float2 v = make_float2((sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]),
                                     (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]));
        sums[c] = v;
        if (d_beta) {
            d_beta[c] = overwrite ? v.x : d_beta[c] + v.x;
            d_gamma[c] = overwrite ? v.y : d_gamma[c] + v.y;
        }
    }
}
// stage 2: dx = rstd * gamma * (g - S0/N - xhat * S1/N)      (biased variance, N = pixels in the statistics)
// 16 bytes per lane; the launch makes the thread count a multiple of the C/4 channel quads, so a thread keeps ITS quad
// across the grid-stride loop and the per-channel terms are loaded and combined once; two float4 pairs in flight per
// iteration.  (Round 2's form -- 4 bytes per lane, five per-element parameter loads and a modulo -- ran at 3 TB/s.)
__global__ __launch_bounds__(256) void inorm_bwd_apply_kernel(const float4* __restrict__ x, const float4* __restrict__ dy,
                                                              const float2* __restrict__ mean_rstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int relu,
                                                              const float2* __restrict__ sums, long npix, int C4,
                                                              float4* __restrict__ dx) {
    const long total = npix * C4;
    const float invn = 1.f / (float)npix;
    const long stride = (long)gridDim.x * blockDim.x;          // a multiple of C4
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c0 = (int)(i % C4) * 4;
    float mean[4], rstd[4], ga[4], be[4], k0[4], k1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float2 mr = mean_rstd[c0 + k], sm = sums[c0 + k];
        mean[k] = mr.x;
        rstd[k] = mr.y;
        ga[k] = gamma ? gamma[c0 + k] : 1.f;
        be[k] = beta ? beta[c0 + k] : 0.f;
        k0[k] = sm.x * invn;
        k1[k] = sm.y * invn;
    }
    auto one = [&](const float4 xv, const float4 gv) {
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xs[k] - mean[k]) * rstd[k];
            const float g = gs[k] * act_grad(ga[k] * xh + be[k], relu);
            o[k] = rstd[k] * ga[k] * (g - k0[k] - xh * k1[k]);
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    };
    for (; i + stride < total; i += 2 * stride) {
        const float4 xa = x[i], ga_ = dy[i], xb = x[i + stride], gb = dy[i + stride];
        dx[i] = one(xa, ga_);
        dx[i + stride] = one(xb, gb);
    }
    if (i < total) dx[i] = one(x[i], dy[i]);
}
int launch_inorm_backward(hipStream_t s, const float* x, const float* dy, const float* mean_rstd, const float* gamma,
                          const float* beta, int relu, long npix, int C, float* scratch, float* dx, float* sums, float* d_beta,
                          float* d_gamma, int overwrite) {
    T2V_REQUIRE(C % 4 == 0, "inorm_backward: C=%d must be a multiple of 4", C);
    // enough (channel group, pixel slice) blocks to cover the chip a few times over; scratch holds 256*C float2
    int slices = (int)((npix + 63) / 64);
    const int want = 1024 / ((C + 63) / 64);
    if (slices > want) slices = want;
    if (slices > 128) slices = 128;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(inorm_bwd_reduce_kernel, dim3((C + 63) / 64, slices), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(dy),
                       reinterpret_cast<const float2*>(mean_rstd), gamma, beta, relu, npix, C,
                       reinterpret_cast<float2*>(scratch));
    hipLaunchKernelGGL(inorm_bwd_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s,
                       reinterpret_cast<const float2*>(scratch), slices, C, reinterpret_cast<float2*>(sums), d_beta, d_gamma,
                       overwrite);
    if (dx) {
        // threads = a multiple of the C/4 channel quads (every thread keeps its quad), ~8 float4 per thread
        const int C4 = C / 4;
        const long total4 = npix * C4;
        long threads = (total4 + 7) / 8;
        threads = (threads + C4 - 1) / C4 * C4;
        long blocks = (threads + 255) / 256;
        if ((blocks * 256) % C4 != 0) {                       // C4 does not divide a whole number of blocks' threads:
            const long lcm_blocks = C4 / gcd_l(C4, 256);      // round the block count up to where it does
            blocks = (blocks + lcm_blocks - 1) / lcm_blocks * lcm_blocks;
        }
        hipLaunchKernelGGL(inorm_bwd_apply_kernel, dim3((int)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                           reinterpret_cast<const float4*>(dy), reinterpret_cast<const float2*>(mean_rstd), gamma, beta, relu,
                           reinterpret_cast<const float2*>(sums), npix, C4, reinterpret_cast<float4*>(dx));
    }
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// activation backward from the OUTPUT y: tanh' = 1-y^2 ; sigmoid' = y(1-y) ; leaky: y>0 ? 1 : slope ; 4: flow/weight head ; scale: slope
__global__ void act_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y, int mode, float slope,
                                    long n, float* __restrict__ dpre) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = y[i], g = dy[i];
        float d;
        if (mode == 1) d = g * (1.f - v * v);
        else if (mode == 2) d = g * v * (1.f - v);
        else if (mode == 3) d = v > 0.f ? g : g * slope;
        else if (mode == 4) {   // T2V_ACT_FLOW_W on [.,4] storage: ch 0,1 = flow * slope, ch 2 = sigmoid, ch 3 unused
            const int c = (int)(i & 3);
            d = c < 2 ? g * slope : (c == 2 ? g * v * (1.f - v) : 0.f);
        } else d = g * slope;
        dpre[i] = d;
    }
}
int launch_act_backward(hipStream_t s, const float* dy, const float* y, int mode, float slope, long n, float* dpre) {
    hipLaunchKernelGGL(act_backward_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, dy, y, mode, slope, n, dpre);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// AvgPool2d(3,2,1,count_include_pad=False) backward: gather form (each input pixel sums its <= 4 windows)
__global__ void avgpool3s2_backward_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C,
                                           int Ho, int Wo) {
    const long total = (long)H * W * C;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int x = (int)(pix % W), y = (int)(pix / W);
        float s = 0.f;
        for (int oy = (y + 1 - 2 + 1) / 2; oy <= (y + 1) / 2; ++oy) {   // windows with 2*oy-1 <= y <= 2*oy+1
            if (oy < 0 || oy >= Ho || 2 * oy - 1 > y || 2 * oy + 1 < y) continue;
            const int ny = min(2 * oy + 1, H - 1) - max(2 * oy - 1, 0) + 1;
            for (int ox = (x + 1 - 2 + 1) / 2; ox <= (x + 1) / 2; ++ox) {
                if (ox < 0 || ox >= Wo || 2 * ox - 1 > x || 2 * ox + 1 < x) continue;
                const int nx = min(2 * ox + 1, W - 1) - max(2 * ox - 1, 0) + 1;
                s += dy[((long)oy * Wo + ox) * C + c] / (float)(ny * nx);
            }
        }
        dx[i] = s;
    }
}
// MaxPool2d(2, 2) (floor mode; VGG19): NHWC, batch folded into the row count by the caller
__global__ void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y, int W, int C, int Ho, int Wo) {
    const long total = (long)Ho * Wo * C;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        const float* r0 = x + ((long)(2 * oy) * W + 2 * ox) * C + c;
        const float* r1 = r0 + (long)W * C;
        y[i] = fmaxf(fmaxf(r0[0], r0[C]), fmaxf(r1[0], r1[C]));
    }
}
// backward: the gradient goes to the first maximum of the window in row-major order (torch's scan order); rows /
// columns past the last full window get zero
__global__ void maxpool2x2_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                           float* __restrict__ dx, int H, int W, int C, int Ho, int Wo) {
    const long total = (long)H * W * C;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int ix = (int)(pix % W), iy = (int)(pix / W);
        const int ox = ix >> 1, oy = iy >> 1;
        float g = 0.f;
        if (ox < Wo && oy < Ho) {
            const float* r0 = x + ((long)(2 * oy) * W + 2 * ox) * C + c;
            const float* r1 = r0 + (long)W * C;
            const float v[4] = {r0[0], r0[C], r1[0], r1[C]};
            int arg = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k] > v[arg]) arg = k;
            if (arg == ((iy & 1) * 2 + (ix & 1))) g = dy[((long)oy * Wo + ox) * C + c];
        }
        dx[i] = g;
    }
}
int launch_maxpool2x2(hipStream_t s, const float* x, float* y, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for((long)Ho * Wo * C, 256)), dim3(256), 0, s, x, y, W, C, Ho, Wo);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
int launch_maxpool2x2_backward(hipStream_t s, const float* x, const float* dy, float* dx, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    hipLaunchKernelGGL(maxpool2x2_backward_kernel, dim3(grid_for((long)H * W * C, 256)), dim3(256), 0, s, x, dy, dx, H, W,
                       C, Ho, Wo);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
int launch_avgpool3s2_backward(hipStream_t s, const float* dy, float* dx, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(avgpool3s2_backward_kernel, dim3(grid_for((long)H * W * C, 256)), dim3(256), 0, s, dy, dx, H, W,
                       C, Ho, Wo);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// loss backward: d/dx sum((x-c)^2)*scale = 2*scale*(x-c) ; d/da sum|a-b|*scale = scale*sign(a-b)
__global__ void loss_backward_kernel(const float* __restrict__ a, const float* __restrict__ b, float c, float scale,
                                     int op, long n, float* __restrict__ da) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (op == 0) {
            da[i] = 2.f * scale * (a[i] - c);
        } else {
            const float d = a[i] - b[i];
            da[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
        }
    }
}
int launch_loss_backward(hipStream_t s, int op, const float* a, const float* b, float c, float scale, long n, float* da) {
    hipLaunchKernelGGL(loss_backward_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, a, b, c, scale, op, n, da);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// inverse of the packers (checkpoint save, gradient export): packed -> torch layout
__global__ void unpack_conv_weight_kernel(const float* __restrict__ packed, float* __restrict__ w, int Cout, int Cin,
                                          int KH, int KW, int Cin_s, int Kp, long total, int accumulate) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long t = i;
        const int kw = (int)(t % KW); t /= KW;
        const int kh = (int)(t % KH); t /= KH;
        const int c = (int)(t % Cin); t /= Cin;
        const int n = (int)t;
        const float v = packed[(size_t)n * Kp + (kh * KW + kw) * Cin_s + c];
        w[i] = accumulate ? w[i] + v : v;
    }
}
int launch_unpack_conv_weight(hipStream_t s, const float* packed, float* w, int Cout, int Cin, int KH, int KW, int Cin_s,
                              int Kp, int accumulate) {
    const long total = (long)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(unpack_conv_weight_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, packed, w, Cout, Cin, KH,
                       KW, Cin_s, Kp, total, accumulate);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
__global__ void unpack_convT_weight_kernel(const float* __restrict__ packed, float* __restrict__ w, int Cin, int Cout,
                                           int Cout_p, int Cin_s, int K, int pad, long total, int accumulate) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long t = i;
        const int kw = (int)(t % K); t /= K;
        const int kh = (int)(t % K); t /= K;
        const int n = (int)(t % Cout); t /= Cout;
        const int c = (int)t;
        // find the phase / tap holding (kh, kw)
        long off = 0;
        float v = 0.f;
        for (int ph = 0; ph < 4; ++ph) {
            int nt, pkh[4], pkw[4], pdy[4], pdx[4], a, b;
            convT_phase_taps(ph, K, pad, &nt, pkh, pkw, pdy, pdx, &a, &b);
            const int Kq = (nt * Cin_s + kBK - 1) / kBK * kBK;
            for (int q = 0; q < nt; ++q)
                if (pkh[q] == kh && pkw[q] == kw) v = packed[off + (size_t)n * Kq + q * Cin_s + c];
            off += (long)Cout_p * Kq;
        }
        w[i] = accumulate ? w[i] + v : v;
    }
}
int launch_unpack_convT_weight(hipStream_t s, const float* packed, float* w, int Cin, int Cout, int Cin_s, int Cout_p,
                               int K, int pad, int accumulate) {
    const long total = (long)Cin * Cout * K * K;
    hipLaunchKernelGGL(unpack_convT_weight_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, packed, w, Cin, Cout,
                       Cout_p, Cin_s, K, pad, total, accumulate);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// dst = src (overwrite) or dst += src; x *= s -- the small-tensor ends of the gradient path (bias / affine gradients
// landing in their slice of a flat exchange bucket, averaging a summed bucket)
__global__ void accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, long n, int overwrite) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = overwrite ? src[i] : dst[i] + src[i];
}
int launch_accumulate(hipStream_t s, float* dst, const float* src, long n, int overwrite) {
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, dst, src, n, overwrite);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
// src [C][2] -> dst0[c] (+)= src[c][0], dst1[c] (+)= src[c][1]: the (sum g, sum g*xhat) pairs of the norm backward into
// the bias / weight gradient slots of the affine parameters
__global__ void unzip2_kernel(const float2* __restrict__ src, float* __restrict__ dst0, float* __restrict__ dst1, int C,
                              int overwrite) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float2 v = src[c];
    dst0[c] = overwrite ? v.x : dst0[c] + v.x;
    dst1[c] = overwrite ? v.y : dst1[c] + v.y;
}
int launch_unzip2(hipStream_t s, const float* src, float* dst0, float* dst1, int C, int overwrite) {
    hipLaunchKernelGGL(unzip2_kernel, dim3((C + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float2*>(src), dst0, dst1,
                       C, overwrite);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
__global__ void scale_kernel(float* __restrict__ x, long n, float sc) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= sc;
}
int launch_scale(hipStream_t s, float* x, long n, float sc) {
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, n, sc);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// per-channel sum over pixels of an NHWC tensor (bias gradient): deterministic two-level reduction,
// grid = (C/64, pixel slices) partials + one final pass
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ x, long npix, int C, int cs,
                                                                  float* __restrict__ partial) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (long p = (long)blockIdx.y * 4 + sl; p < npix; p += (long)gridDim.y * 4) s += x[p * cs + c];
    sh[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
// ... 16 bytes per lane where the channel storage allows: Qp (a power of two >= the quads a block sums, <= 64) quad lanes x
// 256 / Qp pixel lanes, four pixels' loads in flight per thread, pixel lanes added in a fixed order.  (The 4-byte form above
// keeps 4 of 64 lanes busy on a 3-channel head and one load in flight everywhere: 38 us for the 34 MB of a discriminator's
// first layer.)
__global__ __launch_bounds__(256) void channel_sum_partial4_kernel(const float4* __restrict__ x, long npix, int Q, int cs4, int Qp,
                                                                   float4* __restrict__ partial) {
    __shared__ float4 sh[256];
    const int ql = threadIdx.x & (Qp - 1), pl = threadIdx.x / Qp, PL = 256 / Qp;
    const int q = blockIdx.x * 64 + ql;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](const float4 v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
    if (q < Q) {
        const long step = (long)gridDim.y * PL;
        long p = (long)blockIdx.y * PL + pl;
        for (; p + 3 * step < npix; p += 4 * step) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x[(p + u * step) * cs4 + q];
#pragma unroll
            for (int u = 0; u < 4; ++u) add(v[u]);
        }
        for (; p < npix; p += step) add(x[p * cs4 + q]);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < Qp && q < Q) {
        float4 a = sh[ql];
        for (int i = 1; i < PL; ++i) {   // fixed order
            const float4 v = sh[i * Qp + ql];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        partial[(size_t)blockIdx.y * Q + q] = a;
    }
}
// block = 64 channels x 4 slice lanes; `pitch` floats per slice
__global__ __launch_bounds__(256) void channel_sum_final_kernel(const float* __restrict__ partial, int slices, int C, int pitch,
                                                                float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C)
        for (int i = sl; i < slices; i += 4) s += partial[(size_t)i * pitch + c];
    sh[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && c < C) out[c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
int launch_channel_sum(hipStream_t s, const float* x, long npix, int C, int cs, float* scratch, float* out) {
    if (cs % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0) {
        const int Q = (C + 3) / 4;
        int Qp = 1;
        while (Qp < Q && Qp < 64) Qp *= 2;
        const int PL = 256 / Qp, xb = (Q + 63) / 64;
        long slices = (npix + (long)PL * 16 - 1) / ((long)PL * 16);      // ~16 pixels per thread ...
        slices = std::min<long>(slices, std::max(1, 2048 / xb));         // ... up to ~8 blocks per CU
        slices = std::min<long>(slices, (long)256 * C / (4 * Q));        // (scratch: 256 * C floats)
        slices = std::max<long>(slices, 1);
        hipLaunchKernelGGL(channel_sum_partial4_kernel, dim3(xb, (int)slices), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                           npix, Q, cs / 4, Qp, reinterpret_cast<float4*>(scratch));
        hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, scratch, (int)slices, C, 4 * Q, out);
        T2V_HIP_CHECK(hipGetLastError());
        return T2V_OK;
    }
    int slices = (int)((npix + 511) / 512);
    if (slices > 256) slices = 256;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3((C + 63) / 64, slices), dim3(256), 0, s, x, npix, C, cs, scratch);
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s, scratch, slices, C, C, out);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---------------------------------------------------------------------------------------------
// layout plumbing
// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long HW, int Cs) {
    const long total = HW * Cs;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long pix = i / Cs;
        const int c = (int)(i - pix * Cs);
        dst[i] = c < C ? src[(long)c * HW + pix] : 0.f;
    }
}
int launch_nchw_to_nhwc(hipStream_t s, const float* src, float* dst, int C, int H, int W, int Cs) {
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(HW * Cs, 256)), dim3(256), 0, s, src, dst, C, HW, Cs);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long HW, int Cs) {
    const long total = HW * C;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i / HW);
        const long pix = i - (long)c * HW;
        dst[i] = src[pix * Cs + c];
    }
}
int launch_nhwc_to_nchw(hipStream_t s, const float* src, float* dst, int C, int H, int W, int Cs) {
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(HW * C, 256)), dim3(256), 0, s, src, dst, C, HW, Cs);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

__global__ void copy_channels_kernel(const float* __restrict__ src, int src_cs, int src_c0, float* __restrict__ dst,
                                     int dst_cs, int dst_c0, int nc, long npix) {
    const long total = npix * nc;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long pix = i / nc;
        const int c = (int)(i - pix * nc);
        dst[pix * dst_cs + dst_c0 + c] = src[pix * src_cs + src_c0 + c];
    }
}
int launch_copy_channels(hipStream_t s, const float* src, int src_cs, int src_c0, float* dst, int dst_cs,
                         int dst_c0, int nc, long npix) {
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(npix * nc, 256)), dim3(256), 0, s, src, src_cs, src_c0,
                       dst, dst_cs, dst_c0, nc, npix);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ToTensor (u8/255) + Normalize(0.5,0.5): (v/255 - 0.5)/0.5, same operation order as
// $SP/torchvision/transforms/functional.py:38-60,206-208
__global__ void u8_pose_to_f32_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long npix, int Cs,
                                      int c0) {
    const long total = npix * 3;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long pix = i / 3;
        const int c = (int)(i - pix * 3);
        const float v = (float)src[i] / 255.0f;
        dst[pix * Cs + c0 + c] = (v - 0.5f) / 0.5f;
    }
}
int launch_u8_pose_to_f32(hipStream_t s, const uint8_t* src, float* dst, long npix, int Cs, int c0) {
    hipLaunchKernelGGL(u8_pose_to_f32_kernel, dim3(grid_for(npix * 3, 256)), dim3(256), 0, s, src, dst, npix, Cs, c0);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// util.tensor2im: (x+1)/2*255, clip to [0,255], astype(uint8) (truncation)
__global__ void to_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = (x[i] + 1.0f) / 2.0f * 255.0f;
        v = fminf(fmaxf(v, 0.f), 255.f);
        y[i] = (uint8_t)v;
    }
}
int launch_to_u8(hipStream_t s, const float* x, uint8_t* y, long n) {
    hipLaunchKernelGGL(to_u8_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, y, n);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ---------------------------------------------------------------------------------------------
// Flow-warp compositor: grid_sample(prev, base_grid + flow/((W-1)/2,(H-1)/2), bilinear, border,
// corner-aligned) fused with out = raw*w + warp*(1-w).  Same operation order as the reference
// formulation (normalised grid, then un-normalise) so fp32 rounding tracks the oracle.
// raw/out: [H,W,4]; fw: [H,W,4] = (flow_x, flow_y, weight, 0).
// ---------------------------------------------------------------------------------------------
// sampling position of output pixel (x, y): pixel coordinates BEFORE the border clamp
struct WarpPos { float px, py; };
__device__ __forceinline__ WarpPos warp_position(int x, int y, int H, int W, float flow_x, float flow_y) {
    const float sx = (float)(W - 1) / 2.0f, sy = (float)(H - 1) / 2.0f;
    // torch.linspace(-1,1,n): start + i*step for the lower half, end - (n-1-i)*step above
    const float stepx = 2.0f / (float)(W - 1), stepy = 2.0f / (float)(H - 1);
    const float gx0 = x < W / 2 ? -1.0f + stepx * (float)x : 1.0f - stepx * (float)(W - 1 - x);
    const float gy0 = y < H / 2 ? -1.0f + stepy * (float)y : 1.0f - stepy * (float)(H - 1 - y);
    const float gx = gx0 + flow_x / sx, gy = gy0 + flow_y / sy;
    WarpPos p;
    p.px = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    p.py = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    return p;
}

// raw == nullptr: plain resample (out, if given, receives the warped image as well)
__global__ __launch_bounds__(256) void warp_composite_kernel(const float4* __restrict__ raw,
                                                             const float4* __restrict__ fw,
                                                             const float* __restrict__ prev, int prev_cs,
                                                             int prev_c0, float4* __restrict__ out,
                                                             float4* __restrict__ warp_out, int H, int W) {
    const long npix = (long)H * W;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const float4 f = fw[i];
        const WarpPos wp = warp_position(x, y, H, W, f.x, f.y);
        const float px = fminf(fmaxf(wp.px, 0.f), (float)(W - 1));  // padding_mode='border'
        const float py = fminf(fmaxf(wp.py, 0.f), (float)(H - 1));
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const int x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
        const bool okx = x1 < W, oky = y1 < H;
        float wv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pc = prev + prev_c0 + c;
            float v = pc[((long)y0 * W + x0) * prev_cs] * w00;
            if (okx) v += pc[((long)y0 * W + x1) * prev_cs] * w10;
            if (oky) v += pc[((long)y1 * W + x0) * prev_cs] * w01;
            if (okx && oky) v += pc[((long)y1 * W + x1) * prev_cs] * w11;
            wv[c] = v;
        }
        if (raw) {
            const float4 r = raw[i];
            const float wt = f.z;
            out[i] = make_float4(r.x * wt + wv[0] * (1.0f - wt), r.y * wt + wv[1] * (1.0f - wt),
                                 r.z * wt + wv[2] * (1.0f - wt), 0.f);
        } else if (out) {
            out[i] = make_float4(wv[0], wv[1], wv[2], 0.f);
        }
        if (warp_out) warp_out[i] = make_float4(wv[0], wv[1], wv[2], 0.f);
    }
}
int launch_warp_composite(hipStream_t s, const float* raw, const float* fw, const float* prev, int prev_cs,
                          int prev_c0, float* out, float* warp_out, int H, int W) {
    hipLaunchKernelGGL(warp_composite_kernel, dim3(grid_for((long)H * W, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(raw), reinterpret_cast<const float4*>(fw), prev, prev_cs,
                       prev_c0, reinterpret_cast<float4*>(out), reinterpret_cast<float4*>(warp_out), H, W);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// Adjoint of the compositor (SpatialGridSamplerBilinear_updateGradInput, THCUNN.h:1055, fused with the blend's
// adjoint).  With g_c = d_out_c*(1-w) + d_warp_c the gradient that reaches the warped image:
//   d_raw_c  = d_out_c * w                       d_w = sum_c d_out_c * (raw_c - warp_c)
//   d_flow_x = sum_c g_c * [wy0*(v10-v00) + wy1*(v11-v01)]_c      (d px / d flow_x = 1: pixel-unit flow)
//   d_flow_y = sum_c g_c * [wx0*(v01-v00) + wx1*(v11-v10)]_c
//   d_prev   : g_c * (w00, w10, w01, w11) scattered onto the four taps (atomicAdd; only on request)
// Border rule of torch 0.4.1's kernel: bilinear weights from the unclipped position, the four corner INDICES clipped
// into the image -- so outside the image (and exactly on the last row / column, where both corners clip to the
// same pixel) the gradient with respect to that coordinate is zero.  The warp value itself is re-computed (12 taps
// from a 3-channel image that sits in L2) rather than saved by the forward pass.
__global__ __launch_bounds__(256) void warp_composite_backward_kernel(
    const float4* __restrict__ d_out, const float4* __restrict__ d_warp, const float4* __restrict__ raw,
    const float4* __restrict__ fw, const float* __restrict__ prev, int prev_cs, int prev_c0,
    float4* __restrict__ d_raw, float4* __restrict__ d_fw, float* __restrict__ d_prev, int H, int W) {
    const long npix = (long)H * W;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += stride) {
        const int y = (int)(i / W), x = (int)(i - (long)y * W);
        const float4 f = fw[i];
        const WarpPos wp = warp_position(x, y, H, W, f.x, f.y);
        const bool inx = wp.px >= 0.f && wp.px <= (float)(W - 1), iny = wp.py >= 0.f && wp.py <= (float)(H - 1);
        const float px = fminf(fmaxf(wp.px, 0.f), (float)(W - 1));
        const float py = fminf(fmaxf(wp.py, 0.f), (float)(H - 1));
        const float fx0 = floorf(px), fy0 = floorf(py);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);   // clipped corner indices
        const float wx1 = px - fx0, wy1 = py - fy0, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        const float4 go = d_out ? d_out[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gw = d_warp ? d_warp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float wt = d_out ? f.z : 0.f;
        const float gout[3] = {go.x, go.y, go.z};
        const float g[3] = {go.x * (1.0f - wt) + gw.x, go.y * (1.0f - wt) + gw.y, go.z * (1.0f - wt) + gw.z};
        float rawv[3] = {0.f, 0.f, 0.f};
        if (raw) {
            const float4 r = raw[i];
            rawv[0] = r.x; rawv[1] = r.y; rawv[2] = r.z;
        }
        const long o00 = ((long)y0 * W + x0) * prev_cs + prev_c0, o10 = ((long)y0 * W + x1) * prev_cs + prev_c0;
        const long o01 = ((long)y1 * W + x0) * prev_cs + prev_c0, o11 = ((long)y1 * W + x1) * prev_cs + prev_c0;
        float gfx = 0.f, gfy = 0.f, gwt = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v00 = prev[o00 + c], v10 = prev[o10 + c], v01 = prev[o01 + c], v11 = prev[o11 + c];
            const float warp = v00 * (wx0 * wy0) + v10 * (wx1 * wy0) + v01 * (wx0 * wy1) + v11 * (wx1 * wy1);
            gfx += g[c] * (wy0 * (v10 - v00) + wy1 * (v11 - v01));
            gfy += g[c] * (wx0 * (v01 - v00) + wx1 * (v11 - v10));
            gwt += gout[c] * (rawv[c] - warp);
            if (d_prev) {
                atomicAdd(d_prev + o00 + c, g[c] * (wx0 * wy0));
                atomicAdd(d_prev + o10 + c, g[c] * (wx1 * wy0));
                atomicAdd(d_prev + o01 + c, g[c] * (wx0 * wy1));
                atomicAdd(d_prev + o11 + c, g[c] * (wx1 * wy1));
            }
        }
        if (d_raw) d_raw[i] = make_float4(go.x * wt, go.y * wt, go.z * wt, 0.f);
        d_fw[i] = make_float4(inx ? gfx : 0.f, iny ? gfy : 0.f, d_out ? gwt : 0.f, 0.f);
    }
}
int launch_warp_composite_backward(hipStream_t s, const float* d_out, const float* d_warp, const float* raw,
                                   const float* fw, const float* prev, int prev_cs, int prev_c0, float* d_raw,
                                   float* d_fw, float* d_prev, int H, int W) {
    hipLaunchKernelGGL(warp_composite_backward_kernel, dim3(grid_for((long)H * W, 256)), dim3(256), 0, s,
                       reinterpret_cast<const float4*>(d_out), reinterpret_cast<const float4*>(d_warp),
                       reinterpret_cast<const float4*>(raw), reinterpret_cast<const float4*>(fw), prev, prev_cs,
                       prev_c0, reinterpret_cast<float4*>(d_raw), reinterpret_cast<float4*>(d_fw), d_prev, H, W);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// AvgPool2d(3, stride 2, padding 1, count_include_pad=False): divisor = taps inside the image
__global__ void avgpool3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int Ho,
                                  int Wo) {
    const long total = (long)Ho * Wo * C;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        float s = 0.f;
        int n = 0;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                s += x[((long)iy * W + ix) * C + c];
                ++n;
            }
        }
        y[i] = s / (float)n;
    }
}
int launch_avgpool3s2(hipStream_t s, const float* x, float* y, int H, int W, int C) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(avgpool3s2_kernel, dim3(grid_for((long)Ho * Wo * C, 256)), dim3(256), 0, s, x, y, H, W, C,
                       Ho, Wo);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

}  // namespace t2v
