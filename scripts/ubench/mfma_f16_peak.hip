// Micro-benchmark: achievable v_mfma_f32_16x16x32_f16 rate on this box (dev tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    f4 acc[NACC];
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[threadIdx.x + i]; b[i] = (_Float16)in[threadIdx.x + 8 + i]; }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&in, 512 * 4);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg : {256, 512, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 4000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<8>, dim3(wg), dim3(256), 0, 0, out, in, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mfma = (double)wg * 4 * iters * 16 * 8;
            printf("wg %d: %.3f ms  %.1f TFLOP/s  (%.2f cycles/mfma/SIMD at 2.4GHz if 1024 SIMDs busy)\n", wg, ms,
                   mfma * 16384 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / 1024));
        }
    }
    return 0;
}
