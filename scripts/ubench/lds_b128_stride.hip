// Micro-benchmark (dev tool): cycles of ds_read_b128 when lane (q = lane / 16, n = lane % 16) reads the 16-byte word
// q * RS + n (+ a moving offset) -- the B-operand read of iaf_cond_h_kernel (RS = 136) against the lane-linear read of the
// group kernel (RS = 16).  Which row strides are free of bank conflicts depends on how the LDS splits a 64-lane b128 access
// into passes, which the ISA manual does not say.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(unsigned* out, unsigned long long* cyc, int RS, int iters) {
    extern __shared__ u4 lds[];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (u4){(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4, n = lane & 15;
    u4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u4 v = lds[(q * RS + n + 17 * u + (it & 7) * 512) & 8191];
            acc += v;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
    const int strides[] = {16, 136, 132, 140, 144, 152, 17, 20, 24, 32, 48, 64, 68, 72, 80, 272};
    for (int RS : strides) {
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 8192 * 16, 0, out, cyc, RS, 2000);
            (void)hipDeviceSynchronize();
            unsigned long long h[256];
            (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            unsigned long long s = 0;
            for (int i = 0; i < 256; ++i) s += h[i];
            if (s / 256 < best) best = s / 256;
        }
        // 8 waves x 16 reads per iteration on one CU: LDS cycles per wave-read = cycles(100 MHz memtime?) -- report relative
        printf("row stride %4d words: %8llu ticks per workgroup (%.3f per wave-read, 8 waves)\n", RS, best, best / (2000.0 * 16 * 8));
    }
    return 0;
}
