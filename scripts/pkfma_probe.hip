// micro-benchmark: issue rate of v_pk_fma_f32 with a VGPR-pair vs an SGPR-pair multiplier (the 7x7 head's inner loop)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters) {
    f32x2 acc[12], x[10];
    for (int i = 0; i < 12; ++i) acc[i] = (f32x2){0.f, 0.f};
    for (int i = 0; i < 10; ++i) x[i] = (f32x2){in[threadIdx.x + i], in[threadIdx.x + 64 + i]};
    f32x2 w[6];
    for (int i = 0; i < 6; ++i) {
        if (MODE == 0) w[i] = (f32x2){in[threadIdx.x * 2 + i], in[threadIdx.x * 3 + i]};                   // VGPR pair
        else w[i] = (f32x2){__builtin_amdgcn_readfirstlane(__float_as_int(in[i])) * 1e-9f, in[0] * 0.f + 0.5f};
    }
    unsigned long long sw[6];
    for (int i = 0; i < 6; ++i)
        sw[i] = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(in[i + 1])) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(in[i]));
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) asm volatile("" : "+s"(sw[i]));      // stay SGPR pairs the compiler cannot fold
        }
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    if (MODE == 1) {
                        unsigned long long tw = sw[(o + t) % 6];
                        f32x2 ws;
                        __builtin_memcpy(&ws, &tw, 8);
                        acc[o * 4 + q] = __builtin_elementwise_fma(x[q + t], ws, acc[o * 4 + q]);
                    } else {
                        acc[o * 4 + q] = __builtin_elementwise_fma(x[q + t], w[(o + t) % 6], acc[o * 4 + q]);
                    }
                }
    }
    float s = 0;
    for (int i = 0; i < 12; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    int iters = 20000;
    float *in, *out;
    (void)hipMalloc(&in, 4096 * 4); (void)hipMalloc(&out, 4096 * 256 * 4);
    (void)hipMemset(in, 0, 4096 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks = 256; blocks <= 1024; blocks *= 2)
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                double flop = (double)blocks * 256 * iters * 84 * 4.0;
                printf("%s multiplier, blocks=%d (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  (%.2f cycles per v_pk_fma_f32 per SIMD at 2.4 GHz)\n",
                       mode ? "SGPR-pair" : "VGPR-pair", blocks, blocks / 256, ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)blocks / 256 * iters * 84));
            }
    return 0;
}
