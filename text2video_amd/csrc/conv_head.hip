// conv_head.hip -- the generator's 7x7 heads (ReflectionPad2d(3) + Conv2d(C, <=4, 7) + tanh | flow*20,sigmoid)
// on gfx950.  SURVEY.md section 8a rows a5/a10.
//
// Why not the implicit-GEMM kernel: with 3 output channels an MFMA tile is >= 80 % padding, and the
// 49-tap gather re-reads every input pixel 49 times through L2 (6.6 GB at 512x512, the measured
// bound of the MFMA version: 0.5 ms = 19 TFLOP/s).  The kernel below is shaped by the data instead -- fp32 VALU
// (v_pk_fma_f32 at 151 TFLOP/s issue rate with an SGPR-pair multiplier, scripts/pkfma_probe.hip), the 22x22 halo of a
// 16x16 pixel tile staged once per channel group in LDS and reused by all 49 taps, the weights wave-uniform scalars out of
// the SAME packed [Cout_p][Kp] matrix the MFMA kernels use.  9.87 GFLOP per 512x512 launch: 132 us = 75 TFLOP/s.
// (Rounds 1-4 ran a thread-per-pixel form: one ds_read_b128 per 6 packed FMAs made it LDS-bound, 176 us; deleted.)
#include <stdlib.h>

#include "t2v_internal.h"

namespace t2v {

__device__ __forceinline__ void hd_dma16(const float* base, int nbytes, char* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#endif
}

constexpr int kHdTile = 16;                 // 16 x 16 output pixels per block
constexpr int kHdHalo = kHdTile + 6;        // 22
// ---- the strip form ---------------------------------------------------------------------------------------------------
// A thread per pixel is LDS-bound, not VALU-bound: one ds_read_b128 (8 LDS cycles per wave) feeds 6 packed FMAs (24 VALU
// cycles), and with four SIMDs sharing one LDS pipe that is 32 LDS cycles per 24 VALU cycles -- 205 us for 9.87 GFLOP =
// 0.31 of the fp32 VALU peak.  Here a thread owns a 1 x 4 strip of pixels: the 10 halo pixels of a kernel row are read ONCE
// and feed 7 taps x 4 pixels x 3 outputs = 168 packed FMAs (16.8 per LDS read instead of 6).  A strip per thread means a
// quarter of the threads per tile, so the INPUT CHANNELS are split across the four waves of the block instead: every wave
// convolves the whole 16 x 16 tile over its own quarter of the channels -- its 22 x 22 halo of 4 channels per pass staged by
// itself into its own 8 KiB of LDS (LDS-DMA; no block barrier in the channel loop, the waves run decoupled), its weights
// wave-uniform scalar loads as before -- and the four partial sums meet once, through LDS, in the epilogue.
//   A ds_read_b128 walks DOWN a column of the halo (row pitch 23 pixels, odd) with the lanes of one LDS service group on the
//   16 rows of one strip column: no bank conflicts (the first cut, lane -> (lane & 15, lane >> 4), had 52 % conflict cycles).
constexpr int kHsPitch = 23;                // halo row pitch in pixels (22 real + 1 pad)
constexpr int kHsPlane = 512 * 16;          // 22 x 23 = 506 halo slots of 16 bytes, staged by 8 DMA instructions
template <int CIN>      // input channel storage known at compile time (128 | 64: the generators' heads): every weight load of a pass
                        // is base + immediate; 0 = read it from the parameters
__global__ __launch_bounds__(256) void conv_head7x7_strip_kernel(const HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin_s = CIN ? CIN : p.Cin_s;
    int tile;
    {   // block b runs on XCD b % 8: every XCD gets a contiguous band of tile rows, so that the halo overlap of neighbouring
        // tiles (22x22 fetched per 16x16 outputs = 1.9x) is served by that XCD's own L2 (measured on the first form of this
        // kernel: 457 vs 315 us without)
        const int nb = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, idx = b >> 3;
        const int q = nb >> 3, r = nb & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_x = (p.W + kHdTile - 1) / kHdTile;
    const int x0 = (tile % tiles_x) * kHdTile, y0 = (tile / tiles_x) * kHdTile;
    int voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int hp = i * 64 + lane;
        const int hy = hp / kHsPitch, hx = hp - hy * kHsPitch;
        int gy = y0 - 3 + hy, gx = x0 - 3 + hx;
        gy = gy < 0 ? -gy : gy;
        gx = gx < 0 ? -gx : gx;
        gy = min(gy, 2 * p.H - 2 - gy);
        gx = min(gx, 2 * p.W - 2 - gx);
        gy = max(gy, 0);
        gx = max(gx, 0);
        voff[i] = (hy < kHdHalo && hx < kHdHalo) ? (gy * p.W + gx) * cin_s * 4 : 0x7fff0000;     // (pad slots: out of range -> 0)
    }
    const int x_bytes = p.H * p.W * cin_s * 4;
    char* plane = smem + wave * kHsPlane;
    // ds_read_b128 is serviced in four fixed 16-lane groups -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
    // (MI355X_MICROARCH.md, LDS) -- and only lanes of one group conflict: a group = one strip column sx, its 16 lanes = the
    // 16 tile rows, whose halo slots are 23 (odd) apart: 16 different 16-byte slots of the 256-byte bank row
    const int lq = (lane & 31) >> 2;
    const int ty = ((lq >> 1) << 2) | (lane & 3), sx = ((lane >> 5) << 1) | ((0x96 >> lq) & 1);
    const char* sb = plane + (ty * kHsPitch + sx * 4) * 16;

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[3][4];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[o][q] = (f32x2){0.f, 0.f};
    const float* __restrict__ w0 = p.w;
    const float* __restrict__ w1 = p.w + p.Kp;
    const float* __restrict__ w2 = p.w + 2 * p.Kp;
    // wave w takes the 4-channel groups g = w (mod 4): at any time the four waves of a block -- and the other blocks of the CU,
    // which started with it -- sit on the SAME 16 channels = one 64-byte line per (tap, output) of the weight matrix, so the
    // scalar cache serves 15 of 16 weight loads (a wave walking its own contiguous quarter of the channels missed on every
    // one: 20 waves x 2.3 KB of weights per pass against a 16 KiB cache, ~1 us per tap group -- 215 us per launch), and the
    // four waves' halo pieces are the four quarters of one 64-byte segment per pixel
    auto stage = [&](int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) hd_dma16(p.x, x_bytes, plane + i * 1024, voff[i], c * 4);
    };
    for (int c0 = wave * 4; c0 < cin_s; c0 += 16) {
        // (issuing the next pass's DMA under the last two phases of this one -- the plane is dead once the last halo row is in
        // registers -- would hide ~17 us of a 132 us launch, measured by leaving the DMA out; with the DMA inside the unrolled
        // row loop the compiler's allocation went to 256 VGPRs + AGPR copies, one wave per SIMD: 822 us.  Left at the top.)
        stage(c0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's plane has landed (nobody else reads it)
        // A kernel row = 10 halo reads + 21 weight loads + 168 packed FMAs, software-pipelined by hand in phases.  Scalar loads return out of order, so ANY wait on them is lgkmcnt(0) -- for the LDS reads in flight as
        // well: a phase's operands are therefore requested one phase AHEAD, before the FMAs of the previous phase, and each
        // phase ends in one wait that those FMAs (288 / 192 issue cycles, plus the other waves of the SIMD) have covered.
        // wave-uniform weights: scalar loads through the constant address space (the memory clobbers of the waits would turn
        // plain loads into per-lane vector loads; nothing writes the weights while this kernel runs)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) f32x4* cf4;
        // three phases per kernel row: taps 0-2 | 3-4 | 5-6 (36 + 24 + 24 weight scalars, two phases live at a time)
        f32x4 wa[3][3], wb[2][3], wc[2][3];
        float4 xv[2][10];
        const float* __restrict__ b0 = w0 + c0;
        const float* __restrict__ b1 = w1 + c0;
        const float* __restrict__ b2 = w2 + c0;
        auto load_w = [&](int kh, int kw, f32x4(&w)[3]) {
            const int kofs = (kh * 7 + kw) * cin_s;
            w[0] = *(cf4)(b0 + kofs);
            w[1] = *(cf4)(b1 + kofs);
            w[2] = *(cf4)(b2 + kofs);
        };
        auto load_x = [&](int kh, float4(&x)[10]) {
#pragma unroll
            for (int j = 0; j < 10; ++j) x[j] = *reinterpret_cast<const float4*>(sb + (kh * kHsPitch + j) * 16);
        };
        auto taps = [&](const float4(&x)[10], int kw, const f32x4(&w)[3]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 xlo = {x[q + kw].x, x[q + kw].y}, xhi = {x[q + kw].z, x[q + kw].w};
#pragma unroll
                for (int o = 0; o < 3; ++o) acc[o][q] = __builtin_elementwise_fma(xlo, (f32x2){w[o].x, w[o].y}, acc[o][q]);
#pragma unroll
                for (int o = 0; o < 3; ++o) acc[o][q] = __builtin_elementwise_fma(xhi, (f32x2){w[o].z, w[o].w}, acc[o][q]);
            }
        };
        if constexpr (CIN == 0) {      // any other width (narrow test nets): plain tap loop, the compiler's schedule
#pragma unroll 1
            for (int kh = 0; kh < 7; ++kh) {
                load_x(kh, xv[0]);
#pragma unroll
                for (int kw = 0; kw < 7; ++kw) {
                    load_w(kh, kw, wa[0]);
                    taps(xv[0], kw, wa[0]);
                }
            }
        } else {
        load_x(0, xv[0]);
#pragma unroll
        for (int t = 0; t < 3; ++t) load_w(0, t, wa[t]);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) load_w(kh, 3 + t, wb[t]);         // phase B of this row: in flight under phase A
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 3; ++t) taps(xv[kh & 1], t, wa[t]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) load_w(kh, 5 + t, wc[t]);         // phase C: under phase B
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) taps(xv[kh & 1], 3 + t, wb[t]);
            __builtin_amdgcn_sched_barrier(0);
            if (kh < 6) {                                                 // phase A of the next row (+ its halo reads): under C
                load_x(kh + 1, xv[(kh + 1) & 1]);
#pragma unroll
                for (int t = 0; t < 3; ++t) load_w(kh + 1, t, wa[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) taps(xv[kh & 1], 5 + t, wc[t]);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the next pass's DMA overwrites this plane: every ds_read of this pass has returned -- its data fed the FMAs above)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the four waves' partial sums meet: wave w leaves its 12 sums per lane in its own (now idle) plane, thread t of the block
    // then owns pixel (row t >> 4, column t & 15) of the tile and adds the four in wave order
    float* red = reinterpret_cast<float*>(plane);
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(o * 4 + q) * 64 + lane] = acc[o][q].x + acc[o][q].y;
    __syncthreads();
    const int prow = tid >> 4, pcol = tid & 15;
    const int pq = pcol & 3, psx = pcol >> 2;          // strip column and pixel within the strip -> the lane that computed it
    const int src = ((psx >> 1) << 5) | ((((psx & 1) ? 0x7421 : 0x6530) >> (4 * (prow >> 2))) & 7) << 2 | (prow & 3);
    float v[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += reinterpret_cast<const float*>(smem + w * kHsPlane)[(o * 4 + pq) * 64 + src];
        v[o] = sum;
    }
    const int oy = y0 + prow, ox = x0 + pcol;
    if (oy < p.H && ox < p.W) {
        float v0 = v[0] + (p.bias ? p.bias[0] : 0.f);
        float v1 = v[1] + (p.bias && p.Cout > 1 ? p.bias[1] : 0.f);
        float v2 = v[2] + (p.bias && p.Cout > 2 ? p.bias[2] : 0.f);
        if (p.act == T2V_ACT_TANH) {
            v0 = tanhf(v0); v1 = tanhf(v1); v2 = tanhf(v2);
        } else if (p.act == T2V_ACT_FLOW_W) {
            v0 *= p.act_scale; v1 *= p.act_scale; v2 = 1.f / (1.f + expf(-v2));
        } else if (p.act == T2V_ACT_LRELU) {
            v0 = v0 > 0.f ? v0 : v0 * p.act_scale; v1 = v1 > 0.f ? v1 : v1 * p.act_scale; v2 = v2 > 0.f ? v2 : v2 * p.act_scale;
        }
        if (p.Cout < 2) v1 = 0.f;
        if (p.Cout < 3) v2 = 0.f;
        float* dst = p.y + (size_t)(oy * p.W + ox) * p.Cout_s;
        if (p.Cout_s == 4) {
            *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, 0.f);
        } else {
            dst[0] = v0;
            if (p.Cout_s > 1) dst[1] = v1;
            if (p.Cout_s > 2) dst[2] = v2;
            for (int c = 3; c < p.Cout_s; ++c) dst[c] = 0.f;
        }
    }
}

int launch_conv_head7x7(hipStream_t s, const HeadParams& p) {
    constexpr int lds = 4 * kHsPlane;
    const dim3 grid(((p.W + kHdTile - 1) / kHdTile) * ((p.H + kHdTile - 1) / kHdTile));
    if (p.Cin_s == 128) hipLaunchKernelGGL(conv_head7x7_strip_kernel<128>, grid, dim3(256), lds, s, p);
    else if (p.Cin_s == 64) hipLaunchKernelGGL(conv_head7x7_strip_kernel<64>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(conv_head7x7_strip_kernel<0>, grid, dim3(256), lds, s, p);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

// ------------------------------------------------------------------------------------------------
// Conv2d(C, 1, k, zero padding): the last layer of every PatchGAN discriminator (C = 8 * ndf = 512, k = 4: one
// 8192-long dot product per output pixel, 67 x 67 pixels at 512 x 512).  On the implicit-GEMM kernel that is 18
// blocks of 256 K stages each -- 130 us of latency for 37 MFLOP.  Here a wave owns an output pixel: the weight
// vector lives in registers (k*k*C/64 floats per lane, loaded once per wave), each tap is one coalesced 1-2 KiB
// row of the NHWC input, and the 64 partial sums meet in a cross-lane reduction.  Consecutive pixels of a wave
// share 3/4 of their taps through L1.  Bound: L2 reads of the taps (k*k x the input), ~10 us at 67 x 67 x 512.
constexpr int kC1Pix = 4;   // output pixels per wave
template <int KS, int CPL>  // CPL = float4 chunks per lane and tap = Cin_s / 256
__global__ __launch_bounds__(256) void conv_cout1_kernel(const Cout1Params p) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int C4 = p.Cin_s >> 2;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(p.x);
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(p.w);
    float4 wv[KS * KS][CPL];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int h = 0; h < CPL; ++h) wv[t][h] = w4[t * C4 + h * 64 + lane];
    const float bv = p.bias ? p.bias[0] : 0.f;
    const int P = p.Hout * p.Wout;
    for (int i = 0; i < kC1Pix; ++i) {
        const int pix = wave * kC1Pix + i;   // wave-uniform
        if (pix >= P) return;
        const int oy = pix / p.Wout, ox = pix - oy * p.Wout;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const int iy = oy * p.stride - p.pad + kh, ix = ox * p.stride - p.pad + kw;
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                // taps in the zero padding read pixel 0 and are masked: no branch between the loads
                const long base = ok ? ((long)iy * p.W + ix) * C4 : 0;
#pragma unroll
                for (int h = 0; h < CPL; ++h) {
                    float4 xv = x4[base + h * 64 + lane];
                    if (!ok) xv = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 wq = wv[kh * KS + kw][h];
                    acc.x = fmaf(xv.x, wq.x, acc.x);
                    acc.y = fmaf(xv.y, wq.y, acc.y);
                    acc.z = fmaf(xv.z, wq.z, acc.z);
                    acc.w = fmaf(xv.w, wq.w, acc.w);
                }
            }
        float s = (acc.x + acc.y) + (acc.z + acc.w);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) {
            float v = s + bv;
            if (p.act == T2V_ACT_LRELU) v = v > 0.f ? v : v * p.act_scale;
            float* dst = p.y + (size_t)pix * p.Cout_s;
            dst[0] = v;
            for (int c = 1; c < p.Cout_s; ++c) dst[c] = 0.f;
        }
    }
}
bool conv_cout1_supported(int ksize, int Cin_s) { return ksize == 4 && (Cin_s == 256 || Cin_s == 512); }
int launch_conv_cout1(hipStream_t s, const Cout1Params& p) {
    T2V_REQUIRE(conv_cout1_supported(p.ksize, p.Cin_s), "conv_cout1: k=%d Cin_s=%d not supported", p.ksize, p.Cin_s);
    const int P = p.Hout * p.Wout;
    const int blocks = (P + 4 * kC1Pix - 1) / (4 * kC1Pix);
    if (p.Cin_s == 512)
        hipLaunchKernelGGL((conv_cout1_kernel<4, 2>), dim3(blocks), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_cout1_kernel<4, 1>), dim3(blocks), dim3(256), 0, s, p);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

}  // namespace t2v
