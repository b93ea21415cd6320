// Micro-benchmark: does a VALU read of a v_mfma_f32_16x16x32_f16 result see the result?  (the withheld
// hoisted-resident form reads ~0 in lanes 48-63 of the first accumulator register after its twelve residual MFMAs,
// rarely, only in the lowest active wave of a SIMD that holds two active waves and an idle one.)
// The kernel mimics that epilogue in isolation: twelve waves per workgroup, `nact` of them active, all leave a barrier
// together, read their A fragments from LDS, run four chains of three MFMAs from a zero accumulator, optionally wait
// NOPS x 16 wait states, and read the results with VALU compares against the exact expected value (operands are small
// integers: every product is exact).  Idle waves issue LDS-DMA loads like the real kernel's idle waves; FRAG adds the
// 48 KB fragment DMA of every wave and wave 0's partly out-of-range buffer load landing under the MFMAs.
// RESULT (MI355X, ROCm 7.2): 0 wrong values in 200 000 iterations for 12 / 10 / 8 / 5 / 4 / 1 active waves in every
// variant (the positive control fires): the bare pattern is sound, the defect of wn_iaf_r.hip needs something else.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_hazard.hip -o scripts/ubench/mfma_hazard && scripts/ubench/mfma_hazard
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

constexpr int NW = 12;
constexpr int PR_U4 = 8 * 64;                  // residual fragments: 4 row blocks x 2 planes x 64 lanes x 16 B = 8 KB

template <int NOPS, bool DMA, bool FRAG = false>
__global__ __launch_bounds__(64 * NW, 1) void k(unsigned* out, const unsigned* blob, int nact, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    u4* PR = reinterpret_cast<u4*>(lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool mine = wave < nact % 100;
    // hi plane of A: 1.0 in every slot; lo plane: 0
    for (int i = threadIdx.x; i < PR_U4; i += blockDim.x) {
        const bool hi = ((i >> 6) & 1) == 0;
        PR[i] = hi ? (u4){0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u} : (u4){0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const u4* PRl = PR + lane;
    unsigned bad = 0, first = 0;
    for (int it = 0; it < iters; ++it) {
        // B operand of this iteration: column n holds the integer (n + 1 + it % 3) in all 8 K slots of every lane row
        const _Float16 bv = (_Float16)(float)((lane & 15) + 1 + it % 3);
        const unsigned short bb = __builtin_bit_cast(unsigned short, bv);
        const unsigned bw = (unsigned)bb | ((unsigned)bb << 16);
        u4 gh = (u4){bw, bw, bw, bw}, gl = (u4){0u, 0u, 0u, 0u};
        asm volatile("" : "+v"(gh), "+v"(gl));
        if (DMA && !mine) {        // what the idle waves of the real kernel do meanwhile: LDS-DMA into another LDS area
            for (int i = wave; i < 8; i += NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blob + (size_t)(i * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(lds + 4096 + i * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        u4 tl = (u4){0u, 0u, 0u, 0u};
        if (FRAG) {
            // like the real kernel after its K-loop barrier: EVERY wave sends its share of the next layer's 48 KB of
            // fragments to LDS by LDS-DMA and wave 0 has a partly out-of-range buffer load in flight, all of it
            // landing while the residual MFMAs run
            for (int i = wave; i < 48; i += NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blob + (size_t)((it & 3) * 12288 + i * 256 + lane * 4)),
                                                 (__attribute__((address_space(3))) void*)(lds + 4096 + i * 256), 16, 0, 0);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)blob, 0, 1 << 20, 0x00020000);
            tl = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rb, threadIdx.x < 33 ? (int)threadIdx.x * 16 : (int)0x80000000, 0, 0));
        }
        if (mine) {
            f4 rcs[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                rcs[mb] = mfma3(PRl[(mb * 2 + 0) * 64], PRl[(mb * 2 + 1) * 64], gh, gl, (f4){0.f, 0.f, 0.f, 0.f});
            if (NOPS > 0) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NOPS; ++q) asm volatile("s_nop 15" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            const float want = (nact > 100 ? 33.f : 32.f) * (float)((lane & 15) + 1 + it % 3);   // nact > 100: positive control
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rcs[mb][r] != want) {
                        if (!bad) first = (unsigned)it | ((unsigned)(mb * 4 + r) << 24);
                        ++bad;
                    }
        }
        if (FRAG) {
            if (threadIdx.x < 33) reinterpret_cast<u4*>(lds + 4096 + 12288 + 64)[threadIdx.x] = tl;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (bad) {
        const unsigned slot = atomicAdd(out, 1u);
        if (slot < 60) {
            out[4 + 4 * slot] = blockIdx.x | ((unsigned)wave << 16) | ((unsigned)lane << 24);
            out[5 + 4 * slot] = bad;
            out[6 + 4 * slot] = first;
        }
    }
}

template <int NOPS, bool DMA, bool FRAG = false>
void run(unsigned* d_out, const unsigned* d_blob, int nact, int iters, const char* tag) {
    hipMemset(d_out, 0, 1024);
    auto kern = k<NOPS, DMA, FRAG>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * NW), 120 * 1024, 0, d_out, d_blob, nact, iters);
    hipEventRecord(e1, 0);
    const hipError_t err = hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (err != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: LAUNCH FAILED (%s)\n", tag, hipGetErrorString(err)); return; }
    unsigned h[256];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s active waves %2d of 12: %u lanes saw a wrong value", tag, nact, h[0]);
    for (unsigned i = 0; i < h[0] && i < 4; ++i)
        printf("  [wg %u wave %u lane %u: %u times, first at iteration %u acc %u]", h[4 + 4 * i] & 0xffff, (h[4 + 4 * i] >> 16) & 0xff,
               h[4 + 4 * i] >> 24, h[5 + 4 * i], h[6 + 4 * i] & 0xffffff, h[6 + 4 * i] >> 24);
    printf("   (%.2f us per iteration)\n", ms * 1e3 / iters);
}

int main() {
    unsigned *d_out, *d_blob;
    hipMalloc(&d_out, 1024);
    hipMalloc(&d_blob, 1 << 20);
    hipMemset(d_blob, 0, 1 << 20);
    const int iters = 200000;
    run<0, false>(d_out, d_blob, 112, 1000, "positive control (wrong expectation)");
    for (int nact : {12, 10, 8, 5, 4, 1}) {
        run<0, true>(d_out, d_blob, nact, iters, "compiler's spacing, idle DMA");
        run<0, false>(d_out, d_blob, nact, iters, "compiler's spacing");
        run<4, true>(d_out, d_blob, nact, iters, "+64 wait states, idle DMA");
        run<0, false, true>(d_out, d_blob, nact, iters, "compiler's spacing, fragment DMA");
        run<4, false, true>(d_out, d_blob, nact, iters, "+64 wait states, fragment DMA");
    }
    return 0;
}
