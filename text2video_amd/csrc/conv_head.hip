// conv_head.hip -- the generator's 7x7 heads (ReflectionPad2d(3) + Conv2d(C, <=4, 7) + tanh | flow*20,sigmoid)
// on gfx950.  SURVEY.md section 8a rows a5/a10.
//
// Why not the implicit-GEMM kernel: with 3 output channels an MFMA tile is >= 80 % padding, and the
// 49-tap gather re-reads every input pixel 49 times through L2 (6.6 GB at 512x512, the measured
// bound of the MFMA version: 0.5 ms = 19 TFLOP/s).  This kernel is shaped by the data instead:
//   * one block = a 16x16 pixel tile, one thread per pixel, 3 fp32 accumulators per thread;
//   * the 22x22 input halo of the tile is staged ONCE per 16-channel chunk into LDS (LDS-DMA, buffer
//     addressing: per-lane halo offsets with reflection are computed once per block, the chunk is an
//     SGPR soffset) and reused by all 49 taps: L2 traffic 1.9x the input instead of 49x;
//   * LDS image is [4-channel group][halo pixel] so a wave's ds_read_b128 of one tap is contiguous;
//   * weights are wave-uniform: they stream through SGPRs (s_load from the SAME packed [Cout_p][Kp]
//     matrix the MFMA kernels use) and feed v_fmac directly -- no LDS or VGPR traffic for B;
//   * single-buffered 32 KiB halo => 4 blocks (16 waves) per CU: while one block stages its next chunk
//     the others compute -- thread-level parallelism hides DMA, scalar-load and LDS latency, and the
//     tap loop stays rolled (48 live weight scalars; a fully unrolled row spilled SGPRs).
// Bound: fp32 VALU then LDS (6 packed FMAs per ds_read_b128), 9.87 GFLOP per 512x512 launch: 0.16 ms = 62 TFLOP/s.
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
constexpr int kHdPlane = 512 * 16;          // bytes of one 4-channel plane (484 halo pixels, padded to 8 DMA instr)
// CH = channels staged per pass: 16 (32 KiB, 4 blocks/CU) or 32 (64 KiB, 2 blocks/CU; one full 128-byte line
// per halo pixel per pass)
template <int CH>
__global__ __launch_bounds__(256) void conv_head7x7_kernel(const HeadParams p) {
    constexpr int kPlanes = CH / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // also the 4-channel group this wave stages
    const int tx = tid & 15, ty = tid >> 4;
    // block b runs on XCD b % 8: give every XCD a contiguous band of tile rows, so that the halo overlap of
    // neighbouring tiles (22x22 fetched per 16x16 outputs = 1.9x) and the two 64-B chunk halves of a 128-B
    // line are served by that XCD's own L2 -- inside a frame the input was just written by another kernel
    // and does not sit in the Infinity Cache (measured: 457 vs 315 us without this).
    int tile;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, idx = b >> 3;
        const int q = nb >> 3, r = nb & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_x = (p.W + kHdTile - 1) / kHdTile;
    const int x0 = (tile % tiles_x) * kHdTile, y0 = (tile / tiles_x) * kHdTile;

    // halo pixel -> byte offset in x (reflection resolved once per block)
    int voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int hp = i * 64 + lane;
        const int hy = hp / kHdHalo, hx = hp - hy * kHdHalo;
        int gy = y0 - 3 + hy, gx = x0 - 3 + hx;
        gy = gy < 0 ? -gy : gy;
        gx = gx < 0 ? -gx : gx;
        gy = min(gy, 2 * p.H - 2 - gy);
        gx = min(gx, 2 * p.W - 2 - gx);
        gy = max(gy, 0);      // tiles hanging over the bottom/right edge: any valid pixel (their outputs are masked)
        gx = max(gx, 0);
        voff[i] = hp < kHdHalo * kHdHalo ? ((gy * p.W + gx) * p.Cin_s + wave * 4) * 4 : 0x7fff0000;
    }
    const int x_bytes = p.H * p.W * p.Cin_s * 4;
    auto stage = [&](int chunk, int buf) {
#pragma unroll
        for (int h = 0; h < kPlanes / 4; ++h) {   // this wave's 4-channel planes: wave, wave + 4
            char* dst = smem + (wave + 4 * h) * kHdPlane;
#pragma unroll
            for (int i = 0; i < 8; ++i) hd_dma16(p.x, x_bytes, dst + i * 1024, voff[i], chunk * (CH * 4) + h * 64);
        }
    };

    // packed fp32 FMAs (v_pk_fma_f32: two lanes of a register pair per instruction): every output channel keeps an
    // (even, odd) input-channel pair of partial sums, so one ds_read_b128 feeds 6 packed FMAs instead of 12 scalar ones
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
    const int nchunks = p.Cin_s / CH;
    const float* __restrict__ w0 = p.w;                 // row 0 of the packed [Cout_p][Kp] weight
    const float* __restrict__ w1 = p.w + p.Kp;
    const float* __restrict__ w2 = p.w + 2 * p.Kp;
    const int lbase = (ty * kHdHalo + tx) * 16;

    const char* sb = smem + lbase;
    for (int ch = 0; ch < nchunks; ++ch) {
        stage(ch, 0);
        __syncthreads();   // chunk landed (the barrier's fence drains vmcnt)
        for (int kh = 0; kh < 7; ++kh) {
#pragma unroll 1
            for (int kw = 0; kw < 7; ++kw) {
                const int kofs = (kh * 7 + kw) * p.Cin_s + ch * CH;   // wave-uniform: scalar loads below
                const char* st = sb + (kh * kHdHalo + kw) * 16;
#pragma unroll
                for (int q = 0; q < kPlanes; ++q) {
                    const float4 xv = *reinterpret_cast<const float4*>(st + q * kHdPlane);
                    const float4 a = *reinterpret_cast<const float4*>(w0 + kofs + q * 4);
                    const float4 b = *reinterpret_cast<const float4*>(w1 + kofs + q * 4);
                    const float4 c = *reinterpret_cast<const float4*>(w2 + kofs + q * 4);
                    const f32x2 xlo = {xv.x, xv.y}, xhi = {xv.z, xv.w};
                    acc0 = __builtin_elementwise_fma(xlo, (f32x2){a.x, a.y}, acc0);
                    acc1 = __builtin_elementwise_fma(xlo, (f32x2){b.x, b.y}, acc1);
                    acc2 = __builtin_elementwise_fma(xlo, (f32x2){c.x, c.y}, acc2);
                    acc0 = __builtin_elementwise_fma(xhi, (f32x2){a.z, a.w}, acc0);
                    acc1 = __builtin_elementwise_fma(xhi, (f32x2){b.z, b.w}, acc1);
                    acc2 = __builtin_elementwise_fma(xhi, (f32x2){c.z, c.w}, acc2);
                }
            }
        }
        __syncthreads();   // everyone is done with the halo before the next chunk overwrites it
    }

    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < p.H && ox < p.W) {
        float v0 = (acc0.x + acc0.y) + (p.bias ? p.bias[0] : 0.f);
        float v1 = (acc1.x + acc1.y) + (p.bias && p.Cout > 1 ? p.bias[1] : 0.f);
        float v2 = (acc2.x + acc2.y) + (p.bias && p.Cout > 2 ? p.bias[2] : 0.f);
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

template <int CH>
static int launch_head(hipStream_t s, const HeadParams& p) {
    constexpr int lds = (CH / 4) * kHdPlane;
    auto kern = conv_head7x7_kernel<CH>;
    static bool attr_done = false;
    if (!attr_done) {
        T2V_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(((p.W + kHdTile - 1) / kHdTile) * ((p.H + kHdTile - 1) / kHdTile)), dim3(256), lds, s, p);
    T2V_HIP_CHECK(hipGetLastError());
    return T2V_OK;
}

int launch_conv_head7x7(hipStream_t s, const HeadParams& p) {
    // 16 channels per pass (32 KiB, 4 blocks/CU); 32 per pass (64 KiB, 2 blocks/CU) measured slower: 402 vs 253 us
    return launch_head<16>(s, p);
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
