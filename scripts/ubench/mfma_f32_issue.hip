// Micro-benchmark: what keeps a v_mfma_f32_16x16x4_f32 stream from the pipe's rate?  One pure-MFMA loop per configuration:
// waves per SIMD (1 / 2), independent accumulators per wave (4 = the GEMMs' dependency distance, 8), and 16 bytes per lane
// of fresh B operand per 16 MFMAs (the fp32 GEMMs' inner loop) from LDS (one ds_read_b128 / four ds_read_b32) or from
// global memory, or none.  Prints TFLOP/s against 64 FLOP / cycle / SIMD (157.3 at 2.4 GHz); s_memtime is the shader clock.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_f32_issue mfma_f32_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDS, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, const float* in, int iters, unsigned long long* cyc) {
    __shared__ f4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += THREADS) sh[i] = (f4){in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]};
    __syncthreads();
    f4 acc[NACC];
    float a[4];
    for (int o = 0; o < 4; ++o) a[o] = in[(threadIdx.x + 17 * o) & 1023];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    const f4* p = sh + (threadIdx.x & 63);
    f4 bn = p[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f4 b = bn;
            if (LDS == 1) bn = p[64 * ((u + it) & 7)];
            if (LDS == 2) bn = __builtin_nontemporal_load(reinterpret_cast<const f4*>(in) + (threadIdx.x & 63) + 64 * ((u + it) & 3));
            if (LDS == 3) {
                const float* pf = reinterpret_cast<const float*>(sh) + (threadIdx.x & 63) + 256 * ((u + it) & 7);
                bn = (f4){pf[0], pf[64], pf[128], pf[192]};
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ai = NACC == 4 ? i : (i + 4 * (u & 1));
                    acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[jj], acc[ai], 0, 0, 0);
                }
            if (LDS == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (LDS == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if (LDS == 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int LDS, int THREADS>
void run(const char* name, float* out, const float* in, unsigned long long* cyc) {
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, LDS, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, in, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double cm = 0;
    for (auto v : c) cm += (double)v / grid;
    const double mfma_per_simd = (double)iters * 128 * (THREADS / 256);
    const double flop = mfma_per_simd * 2048.0 * 1024;      // 1 024 SIMDs
    // (thread 0's wave is the oldest of its SIMD and the matrix pipe serves waves by age: with two waves per SIMD its span is
    // its OWN work at the full rate, not the launch -- the cycle columns are printed for one wave per SIMD only)
    if (THREADS == 256)
        printf("%-44s %7.1f TFLOP/s  %6.2f cycles per MFMA (32 = the pipe)  clock %.0f MHz\n", name, flop / (ms * 1e-3) / 1e12,
               cm / mfma_per_simd, cm / (ms * 1e-3) / 1e6);
    else
        printf("%-44s %7.1f TFLOP/s\n", name, flop / (ms * 1e-3) / 1e12);
}

int main() {
    float *out, *in;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 8192); hipMalloc(&cyc, 256 * 8);
    std::vector<float> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h.data(), 8192, hipMemcpyHostToDevice);
    run<8, 0, 256>("1 wave / SIMD, 8 accumulators, MFMA only", out, in, cyc);
    run<4, 0, 256>("1 wave / SIMD, 4 accumulators, MFMA only", out, in, cyc);
    run<4, 1, 256>("1 wave / SIMD, ds_read_b128 per 16 MFMAs", out, in, cyc);
    run<4, 3, 256>("1 wave / SIMD, 4 ds_read_b32 per 16 MFMAs", out, in, cyc);
    run<4, 2, 256>("1 wave / SIMD, global_load_b128 per 16", out, in, cyc);
    run<4, 0, 512>("2 waves / SIMD, MFMA only", out, in, cyc);
    run<4, 1, 512>("2 waves / SIMD, ds_read_b128 per 16 MFMAs", out, in, cyc);
    run<4, 3, 512>("2 waves / SIMD, 4 ds_read_b32 per 16 MFMAs", out, in, cyc);
    run<4, 2, 512>("2 waves / SIMD, global_load_b128 per 16", out, in, cyc);
    return 0;
}
