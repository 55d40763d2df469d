// Probe (not product code): what would the Winograd F(4x4) GEMM stage cost on the bf16 matrix cores with fp32
// operands split into bf16 terms?  36 x ([256 x 1024] x [1024 x 1024]^T), fp32 accumulate.
//   planes 3 / products 6: a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1   (fp32-grade result, bf16x3_error_probe.py)
//   planes 2 / products 3: a1b1 + a1b2 + a2b1                         (~16-bit mantissa result)
// Same structure as conv_igemm.hip: 128x128 block tile, 4 MFMA waves (64x64 each) + 4 loader waves, LDS-DMA into an
// XOR-swizzled image of 128-byte rows (here: two 32-element bf16 matrix rows per LDS row), ring of RING stages of
// K = 32, counted vmcnt, one barrier per stage.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/bf16x3_gemm_probe.hip -o scripts/bin/bf16x3_gemm_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int P = 36, M = 256, N = 1024, K = 1024;
constexpr int BM = 128, BN = 128, BK = 32;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const void* base, unsigned nbytes, char* lds_dst, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#endif
}

template <int NPL, int RING, int ABL = 0>   // ABL (ablations): 1 = no DMA inside the K loop, 2 = no MFMAs
__global__ __launch_bounds__(512) void gemm_split_kernel(const u16* __restrict__ A, const u16* __restrict__ B, float* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = NPL * 2 * 64 * 128;   // [plane][A|B][64 LDS rows][128 B]
    constexpr int LD = NPL * 4;                 // DMA instructions per loader wave and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wave >= 4;
    const int wid = wave & 3;
    // block -> tile, a contiguous run of tiles per XCD (block b runs on XCD b % 8)
    int tile;
    {
        const int nb = gridDim.x, b = blockIdx.x;
        const int xcd = b & 7, idx = b >> 3;
        const int q = nb >> 3, r = nb & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int p = tile / 16, mt = (tile >> 3) & 1, nt = tile & 7;
    const int m0 = mt * BM, n0 = nt * BN;
    constexpr int nk = K / BK;

    if (is_loader) {
        int voff[LD];
#pragma unroll
        for (int n = 0; n < LD; ++n) {
            const int plane = n / 4, op = (n % 4) / 2, rb = ((n % 4) % 2) * 4 + wid;
            const int rho = rb * 8 + (lane >> 3), slot = lane & 7;
            const int sp = slot ^ ((rho >> 1) & 7);
            const int r = 2 * rho + (sp >> 2), kc = sp & 3;
            voff[n] = op == 0 ? (((plane * P + p) * M + m0 + r) * K + kc * 8) * 2 : (((plane * P + p) * N + n0 + r) * K + kc * 8) * 2;
        }
        auto issue_stage = [&](int kt, int slot) {
#pragma unroll
            for (int n = 0; n < LD; ++n) {
                const int plane = n / 4, op = (n % 4) / 2, rb = ((n % 4) % 2) * 4 + wid;
                char* dst = smem + slot * STAGE + ((plane * 2 + op) * 64 + rb * 8) * 128;
                if (op == 0)
                    dma16(A, (unsigned)NPL * P * M * K * 2u, dst, voff[n], kt * (BK * 2));
                else
                    dma16(B, (unsigned)NPL * P * N * K * 2u, dst, voff[n], kt * (BK * 2));
            }
        };
        constexpr int AHEAD = RING - 1;
#pragma unroll
        for (int st = 0; st < AHEAD; ++st) issue_stage(st < nk ? st : nk - 1, st);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * LD) : "memory");
        __builtin_amdgcn_s_barrier();
        int slot = AHEAD % RING;
        for (int kt = 0; kt < nk; ++kt) {
            if (ABL != 1) issue_stage(kt + AHEAD < nk ? kt + AHEAD : nk - 1, slot);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * LD) : "memory");
            __builtin_amdgcn_s_barrier();
            slot = slot == RING - 1 ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    const int wm = wid >> 1, wn = wid & 1;
    const int fr = lane & 31, g = lane >> 5;
    const int fsw = (fr >> 2) & 7;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 af[2][2][NPL], bfr[2][2][NPL];   // [register set][tile][plane]
    auto load_frags = [&](int buf, int q, int set) {
        const char* st = smem + buf * STAGE;
        const int slot = ((((fr & 1) * 4) + 2 * q + g) ^ fsw) * 16;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[set][i][pl] = *reinterpret_cast<const bf16x8*>(st + ((pl * 2 + 0) * 64 + wm * 32 + i * 16 + (fr >> 1)) * 128 + slot);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bfr[set][j][pl] = *reinterpret_cast<const bf16x8*>(st + ((pl * 2 + 1) * 64 + wn * 32 + j * 16 + (fr >> 1)) * 128 + slot);
        }
    };
    auto mfmas = [&](int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (NPL == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i][2], bfr[set][j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i][0], bfr[set][j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i][1], bfr[set][j][1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i][1], bfr[set][j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i][0], bfr[set][j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[set][i][0], bfr[set][j][0], acc[i][j], 0, 0, 0);
            }
    };

    __syncthreads();   // B0
    load_frags(0, 0, 0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int nbuf = buf == RING - 1 ? 0 : buf + 1;
        __builtin_amdgcn_sched_barrier(0);
        load_frags(buf, 1, 1);
        if (ABL != 2) mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();   // barrier(kt): slot `buf` fully read, stage kt+1 visible
        __builtin_amdgcn_sched_barrier(0);
        load_frags(nbuf, 0, 0);
        if (ABL != 2) mfmas(1);
        buf = nbuf;
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
                const int col = n0 + wn * 64 + j * 32 + fr;
                C[((size_t)p * M + row) * N + col] = acc[i][j][r];
            }
}

__device__ __forceinline__ u16 f2bf(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((unsigned)h << 16); }

__global__ void init_kernel(float* x, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        x[i] = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.0f;
    }
}
__global__ void split_kernel(const float* x, u16* planes, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        const u16 a = f2bf(v);
        const float r1 = v - bf2f(a);
        const u16 b = f2bf(r1);
        const float r2 = r1 - bf2f(b);
        planes[i] = a;
        planes[n + i] = b;
        planes[2 * n + i] = f2bf(r2);
    }
}

template <int NPL, int RING, int ABL = 0>
static float run(const u16* A, const u16* B, float* C, int iters) {
    auto kern = gemm_split_kernel<NPL, RING, ABL>;
    const int lds = RING * NPL * 2 * 64 * 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = P * (M / BM) * (N / BN);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, A, B, C);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, A, B, C);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

static void check(const float* hA, const float* hB, const float* dC, const char* tag) {
    // 256 sampled outputs against fp64 dot products of the fp32 operands
    double worst = 0, rms_ref = 0, rms_err = 0;
    std::vector<float> got(1);
    for (int s = 0; s < 256; ++s) {
        const int p = (s * 7) % P, m = (s * 37 + 5) % M, n = (s * 101 + 13) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[((size_t)p * M + m) * K + k] * (double)hB[((size_t)p * N + n) * K + k];
        float v;
        CK(hipMemcpy(&v, dC + ((size_t)p * M + m) * N + n, 4, hipMemcpyDeviceToHost));
        const double e = fabs(v - ref);
        worst = e > worst ? e : worst;
        rms_ref += ref * ref;
        rms_err += e * e;
    }
    printf("  %s: max |err| %.3e, rms err %.3e, rms of outputs %.3e  (relative rms %.2e)\n", tag, worst, sqrt(rms_err / 256), sqrt(rms_ref / 256),
           sqrt(rms_err / rms_ref));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    const size_t nA = (size_t)P * M * K, nB = (size_t)P * N * K, nC = (size_t)P * M * N;
    float *dA, *dB, *dC;
    u16 *pA, *pB;
    CK(hipMalloc(&dA, nA * 4)); CK(hipMalloc(&dB, nB * 4)); CK(hipMalloc(&dC, nC * 4));
    CK(hipMalloc(&pA, nA * 2 * 3)); CK(hipMalloc(&pB, nB * 2 * 3));
    hipLaunchKernelGGL(init_kernel, dim3(2048), dim3(256), 0, 0, dA, nA, 1u);
    hipLaunchKernelGGL(init_kernel, dim3(2048), dim3(256), 0, 0, dB, nB, 2u);
    hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, 0, dA, pA, nA);
    hipLaunchKernelGGL(split_kernel, dim3(2048), dim3(256), 0, 0, dB, pB, nB);
    CK(hipDeviceSynchronize());
    std::vector<float> hA(nA), hB(nB);
    CK(hipMemcpy(hA.data(), dA, nA * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), dB, nB * 4, hipMemcpyDeviceToHost));
    const double gf = 2.0 * P * M * N * (double)K / 1e9;
    float ms;
    CK(hipMemset(dC, 0, nC * 4));
    ms = run<3, 3>(pA, pB, dC, iters);
    printf("6 products, ring 3: %.1f us  (%.0f TF executed bf16, %.0f TF fp32-equivalent)\n", ms * 1e3, 6 * gf / ms, gf / ms);
    check(hA.data(), hB.data(), dC, "6 products");
    CK(hipMemset(dC, 0, nC * 4));
    ms = run<3, 2>(pA, pB, dC, iters);
    printf("6 products, ring 2: %.1f us  (%.0f TF executed bf16)\n", ms * 1e3, 6 * gf / ms);
    CK(hipMemset(dC, 0, nC * 4));
    // 3 products read only the first two planes of the same buffers (the plane stride does not depend on NPL)
    ms = run<2, 3>(pA, pB, dC, iters);
    printf("3 products, ring 3: %.1f us  (%.0f TF executed bf16)\n", ms * 1e3, 3 * gf / ms);
    check(hA.data(), hB.data(), dC, "3 products");
    ms = run<3, 3, 1>(pA, pB, dC, iters);
    printf("ablation, 6 products, no DMA in the K loop: %.1f us\n", ms * 1e3);
    ms = run<3, 3, 2>(pA, pB, dC, iters);
    printf("ablation, 6 products, no MFMAs (DMA + LDS reads + barriers): %.1f us\n", ms * 1e3);
    return 0;
}
