// micro-benchmark: fp32 MFMA issue ceiling with real (random) data, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
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
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    int threads = argc > 1 ? atoi(argv[1]) : 256;
    int zero = argc > 2 ? atoi(argv[2]) : 0;
    int iters = 20000;
    float *in, *out;
    hipMalloc(&in, 512 * 8 * 4); hipMalloc(&out, 256 * 512 * 4 * 4);
    float h[512 * 8];
    for (int i = 0; i < 512 * 8; ++i) h[i] = zero ? 0.f : (rand() / (float)RAND_MAX * 2 - 1) * 1e-3f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 256.0 * (threads / 64) * iters * 16 * 2.0 * 32 * 32 * 2;
        printf("threads=%d zero=%d: %.3f ms  %.1f TFLOP/s\n", threads, zero, ms, flop / ms / 1e9);
    }
    return 0;
}
