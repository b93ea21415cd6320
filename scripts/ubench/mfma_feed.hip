// Micro-benchmark: split-fp16 GEMM inner loop of the conditioning GEMM in isolation -- weight fragments from
// LDS (two ds_read_b128 per six MFMAs), activation operand in registers, 8 accumulators per wave -- at 1, 2
// and 4 waves per SIMD, with and without the LDS reads.  What fraction of the bare MFMA rate survives the feed?
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_feed.hip -o scripts/ubench/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ inline f4 mf(u4 a, u4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ inline f4 mfma3(u4 ah, u4 al, u4 bh, u4 bl, f4 c) { return mf(al, bh, mf(ah, bl, mf(ah, bh, c))); }

constexpr int A_U4 = 8 * 4 * 2 * 64;          // one row block's fragments: 64 KB (two images in LDS)
template <int MODE, int RING, int NCB>        // NCB column blocks per wave; MODE 0: fragments fixed in registers; 1: from LDS, compiler order; 2: pinned ring
__global__ __launch_bounds__(1024) void k(float* out, const unsigned* in, int units) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    u4* Ab = reinterpret_cast<u4*>(lds);
    for (int i = threadIdx.x; i < 2 * A_U4; i += blockDim.x) Ab[i] = reinterpret_cast<const u4*>(in)[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u4 bh[8][NCB], bl[8][NCB];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < NCB; ++e) {
            bh[ks][e] = reinterpret_cast<const u4*>(in)[(threadIdx.x + 7 * ks + e) & 1023];
            bl[ks][e] = reinterpret_cast<const u4*>(in)[(threadIdx.x + 3 * ks + e + 5) & 1023];
        }
    f4 tot = {0, 0, 0, 0};
    for (int u = 0; u < units; ++u) {
        const u4* A = Ab + (u & 1) * A_U4 + lane;
        asm volatile("" : "+v"(bh[0][0]));          // the operands are not loop-invariant to the compiler
        f4 acc[4][NCB];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < NCB; ++e) acc[mb][e] = (f4){0, 0, 0, 0};
        if (MODE == 0) {
            const u4 a0 = A[0], a1 = A[64];
#pragma unroll
            for (int p = 0; p < 32; ++p)
#pragma unroll
                for (int e = 0; e < NCB; ++e) acc[p & 3][e] = mfma3(a0, a1, bh[p >> 2][e], bl[p >> 2][e], acc[p & 3][e]);
        } else {
            u4 ar[RING][2];
#pragma unroll
            for (int p = 0; p < RING - 1; ++p) { ar[p][0] = A[(p * 2) * 64]; ar[p][1] = A[(p * 2 + 1) * 64]; }
            if (MODE == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (RING - 1), 0);
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                if (p + RING - 1 < 32) {
                    ar[(p + RING - 1) % RING][0] = A[((p + RING - 1) * 2) * 64];
                    ar[(p + RING - 1) % RING][1] = A[((p + RING - 1) * 2 + 1) * 64];
                }
#pragma unroll
                for (int e = 0; e < NCB; ++e)
                    acc[p & 3][e] = mfma3(ar[p % RING][0], ar[p % RING][1], bh[p >> 2][e], bl[p >> 2][e], acc[p & 3][e]);
                if (MODE == 2) {
                    if (p + RING - 1 < 32) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * NCB, 0);
                }
            }
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < NCB; ++e) tot += acc[mb][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
}

template <int MODE, int RING, int NCB>
void run(const char* name, float* out, const unsigned* in) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, RING, NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * A_U4 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {4, 8}) {
        if (waves * 64 * (NCB * 80 + 64) > 512 * 1024 / 4 * 4) {}
        const int units = 2400 / (waves / 4);
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<MODE, RING, NCB>), dim3(256), dim3(64 * waves), 2 * A_U4 * 16, 0, out, in, units);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double mfma = 256.0 * waves * units * 96 * NCB;
        printf("%-28s %2d waves/CU: %7.3f ms  %7.1f TFLOP/s\n", name, waves, best, mfma * 16384 / best / 1e9);
    }
}
int main() {
    float* out; unsigned* in;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&in, 1024 * 16);
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0x38003800u + (unsigned)((i * 2654435761u) >> 22);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, 2, 2>("regs, 2 col blocks", out, in);
    run<0, 2, 4>("regs, 4 col blocks", out, in);
    run<1, 4, 2>("LDS compiler order, 2 cb", out, in);
    run<2, 3, 2>("LDS ring 3 pinned, 2 cb", out, in);
    run<2, 4, 2>("LDS ring 4 pinned, 2 cb", out, in);
    run<2, 6, 2>("LDS ring 6 pinned, 2 cb", out, in);
    run<2, 4, 4>("LDS ring 4 pinned, 4 cb", out, in);
    run<1, 4, 4>("LDS compiler order, 4 cb", out, in);
    return 0;
}
