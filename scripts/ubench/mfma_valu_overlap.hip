// Micro-benchmark: do VALU / transcendental instructions of a wave execute in the shadow of its own MFMAs, and of
// another wave's MFMAs on the same SIMD?  (dev tool, not product)
//   mode 0: MFMA only (12 independent accumulators x ITER)      mode 1: VALU only (fma + exp + rcp mix of an epilogue)
//   mode 2: both in ONE wave, interleaved by the compiler        mode 3: waves 0-3 MFMA only, waves 4-7 VALU only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, const float* in, int iters, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    f4 acc[12];
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(threadIdx.x & 63) + i]; b[i] = (_Float16)in[(threadIdx.x & 63) + 8 + i]; }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = (f4){0, 0, 0, 0};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[(threadIdx.x & 63) + 16 + i];
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    if (MODE == 4 && wave >= 4) return;
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    if (MODE != 3 && wave >= 4) return;
    const int iters_main = MODE == 4 ? 0 : iters;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters_main; ++it) {
        if (do_m) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);   // 36 MFMAs
        }
        if (do_v) {
            // ~80 plain VALU + 16 transcendental: half an epilogue block
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = fmaf(v[i], 1.0001f, v[i + 8]);
                float e1 = __builtin_amdgcn_exp2f(-x), e2 = __builtin_amdgcn_exp2f(2.f * x);
                float r1 = __builtin_amdgcn_rcpf(1.f + e1), r2 = __builtin_amdgcn_rcpf(1.f + e2);
                float g = r1 * fmaf(-2.f, r2, 1.f);
                v[i] = fmaf(g, 0.5f, v[i] * 0.25f);
                v[i + 8] = fmaf(v[i + 8], 0.999f, g) + x * 1e-3f;
                v[i] = fmaf(v[i], 0.9f, 0.01f); v[i + 8] = fmaf(v[i + 8], 0.9f, 0.02f);
            }
        }
    }
    if (MODE == 4) {
        // one wave, MFMA and VALU instructions interleaved 1 : 3 by scheduling groups
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = fmaf(v[i], 1.0001f, v[i + 8]);
                float e1 = __builtin_amdgcn_exp2f(-x), e2 = __builtin_amdgcn_exp2f(2.f * x);
                float r1 = __builtin_amdgcn_rcpf(1.f + e1), r2 = __builtin_amdgcn_rcpf(1.f + e2);
                float g = r1 * fmaf(-2.f, r2, 1.f);
                v[i] = fmaf(g, 0.5f, v[i] * 0.25f);
                v[i + 8] = fmaf(v[i + 8], 0.999f, g) + x * 1e-3f;
                v[i] = fmaf(v[i], 0.9f, 0.01f); v[i + 8] = fmaf(v[i + 8], 0.9f, 0.02f);
            }
#pragma unroll
            for (int g = 0; g < 36; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // 3 VALU
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 3 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int MODE>
void run(float* out, float* in, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, in, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d: cycles per iteration: wave0 %.1f wave4 %.1f   (36 MFMAs = 576 pipe cycles)\n", MODE, (double)h[0] / iters, (double)h[4] / iters);
}
int main() {
    float *out, *in; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 512 * 4); hipMalloc(&cyc, 64);
    hipMemset(cyc, 0, 64);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) { run<0>(out, in, cyc); run<1>(out, in, cyc); run<2>(out, in, cyc); run<3>(out, in, cyc); run<4>(out, in, cyc); }
    return 0;
}
