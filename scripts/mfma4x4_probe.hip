// micro-benchmark: issue rate of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4x1: 512 FLOP per instruction)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(e + i) & 3], b[(e * 3 + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    int iters = 40000;
    float *in, *out;
    hipMalloc(&in, 256 * 8 * 4); hipMalloc(&out, 1024 * 256 * 4);
    float h[256 * 8];
    for (int i = 0; i < 256 * 8; ++i) h[i] = (rand() / (float)RAND_MAX * 2 - 1) * 1e-3f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks = 256; blocks <= 1024; blocks *= 2)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * 32 * 512.0;
            printf("blocks=%d (x4 waves): %.3f ms  %.1f TFLOP/s  (%.2f cycles per MFMA per SIMD at 2.4 GHz)\n", blocks, ms, flop / ms / 1e9,
                   ms * 1e-3 * 2.4e9 / ((double)blocks / 256 * iters * 32));
        }
    return 0;
}
