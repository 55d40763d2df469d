// micro-benchmark: do the fp32 matrix pipe and the fp32 vector pipe of a SIMD run concurrently?
// Block = 8 waves (2 per SIMD).  mode 0: waves 0-3 issue fp32 MFMAs, waves 4-7 idle; mode 1: waves 4-7 issue
// v_pk_fma_f32 chains, waves 0-3 idle; mode 2: both.  Register-only, no memory in the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (mode == 1) return;
        float a[4], b[4];
        for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[(e + 1) & 3], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(e + 1) & 3], b[e], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(e + 2) & 3], b[(e + 3) & 3], acc[3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        if (mode == 0) return;
        // 16 independent packed accumulators, 2 packed operands: 16 v_pk_fma_f32 per unrolled step, x16 steps = the
        // VALU issue slots of 16 MFMAs' worth of time (16 x 64 cycles = 256 pk_fma at 4 cycles each)
        f32x2 acc[16], x[4], y[4];
        for (int i = 0; i < 4; ++i) {
            x[i] = f32x2{in[threadIdx.x * 8 + i], in[threadIdx.x * 8 + 4 + i]};
            y[i] = f32x2{in[threadIdx.x * 8 + 7 - i], in[threadIdx.x * 8 + 3 - i]};
        }
        for (int i = 0; i < 16; ++i) acc[i] = f32x2{0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(x[(i + r) & 3], y[(i * 3 + r) & 3], acc[i]);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    int iters = 4000;
    float *in, *out;
    hipMalloc(&in, 512 * 8 * 4); hipMalloc(&out, 1024 * 512 * 4);
    float h[512 * 8];
    for (int i = 0; i < 512 * 8; ++i) h[i] = (rand() / (float)RAND_MAX * 2 - 1) * 1e-3f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"MFMA only (4 waves/CU)", "pk_fma only (4 waves/CU)", "both (4 + 4 waves/CU)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256 * 4), dim3(512), 0, 0, in, out, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fm = mode != 1 ? 1024.0 * 4 * iters * 16 * 2.0 * 32 * 32 * 2 : 0;
            double fv = mode != 0 ? 1024.0 * 4 * iters * 256.0 * 64 * 2 * 2 : 0;
            printf("%-26s %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", names[mode], ms, fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
        }
    return 0;
}
