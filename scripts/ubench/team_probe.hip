// Micro-benchmark (dev tool, not product): can ONE K-loop wave per SIMD keep the matrix pipe fed from LDS while an
// epilogue wave runs beside it on the same SIMD?  (The two-team form of the group kernel stands or falls with this; round 3
// measured 10-14 k cycles for a lone K wave's 216 MFMAs = 3.5 k cycles of pipe and concluded that the pipes serialise.)
//   waves 0-3 ("K team", one per SIMD): the group kernel's K loop on NB blocks -- per K-step 8 fragment reads + 2 NB operand
//     reads (ds_read_b128) and 12 NB MFMAs, fragments / operands from a 48 KB image and the l planes in LDS
//   waves 4-7 ("E team"): pair_epilogue_n<NB> on accumulators in registers, own-block words read from / written to LDS
// modes: 0 both teams   1 K team only   2 E team only.   Prints cycles per block-layer for either team.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I nsynth_wavenet_amd/csrc [-DWN_SPLIT_PLAIN=1] [-DNB=3] -o team_probe team_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include "wn_iaf_c.h"
#include "pair_epilogue_n.h"

#ifndef NB
#define NB 3
#endif

namespace {

constexpr int BLK = 2048, NBLK = 24, PLANE = (NBLK + 1) * BLK;
constexpr int A_OFF = 2 * PLANE, T_OFF = A_OFF + LC_A_WORDS * 4, LDS_BYTES = T_OFF + LC_TAIL_WORDS * 4;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, const float* in, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    char* lds = reinterpret_cast<char*>(ldsw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 512) {
        unsigned v = 0x14001800u + (unsigned)((i * 2654435761u) >> 28);
        if (i >= T_OFF / 4 + IAF_PR_FLOATS) v = __float_as_uint(0.01f * (float)(i & 15));
        ldsw[i] = v;
    }
    __syncthreads();
    const bool kteam = wave < 4;
    const int u = wave & 3;
    unsigned long long t0 = 0, t1 = 0;
    float sink = 0.f;
    if (kteam) {
        if (MODE == 2) return;
        const wn_u4* Pl = reinterpret_cast<const wn_u4*>(lds + A_OFF) + lane;
        int ba[NB][3];
#pragma unroll
        for (int e = 0; e < NB; ++e)
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int c = 16 * (u + 4 * e + 2) + n - (2 - tap) * 4;
                ba[e][tap] = ((c >> 4) + 1) * BLK + q * 256 + (c & 15) * 16;
            }
        f4 acc[NB][4];
#pragma unroll
        for (int e = 0; e < NB; ++e)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[e][mb] = (f4){0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime();
        // explicit one-step software pipeline: the operands of K-step ks + 1 are requested before the MFMAs of ks are issued,
        // and a scheduling barrier per step keeps hipcc from hoisting every read of the layer to the top (254 VGPRs + spills)
        auto ld = [&](int ks, wn_u4 (&a)[4][2], wn_u4 (&bh)[NB], wn_u4 (&bl)[NB]) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                a[mb][0] = Pl[((ks * 4 + mb) * 2 + 0) * 64];
                a[mb][1] = Pl[((ks * 4 + mb) * 2 + 1) * 64];
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                bh[e] = *reinterpret_cast<const wn_u4*>(lds + ba[e][ks >> 1] + (ks & 1) * 1024);
                bl[e] = *reinterpret_cast<const wn_u4*>(lds + PLANE + ba[e][ks >> 1] + (ks & 1) * 1024);
            }
        };
        for (int it = 0; it < iters; ++it) {
            wn_u4 a[2][4][2], bh[2][NB], bl[2][NB];
            ld(0, a[0], bh[0], bl[0]);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const int cur = ks & 1;
                if (ks < 5) ld(ks + 1, a[cur ^ 1], bh[cur ^ 1], bl[cur ^ 1]);
#pragma unroll
                for (int e = 0; e < NB; ++e)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) acc[e][mb] = mfma3(a[cur][mb][0], a[cur][mb][1], bh[cur][e], bl[cur][e], acc[e][mb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int e = 0; e < NB; ++e)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) sink += acc[e][mb][0] + acc[e][mb][3];
    } else {
        if (MODE == 1) return;
        const float* tailf = reinterpret_cast<const float*>(lds + T_OFF);
        PairLayer W;
        W.Pl = nullptr;
        W.PRl = reinterpret_cast<const wn_u4*>(lds + T_OFF) + lane;
        W.bg = tailf + IAF_PR_FLOATS + q * 16;
        W.br = W.bg + 64;
        W.inv_m = 0.5f;
        W.inv_r = 0.25f;
        f4 acc[NB][4];
#pragma unroll
        for (int e = 0; e < NB; ++e)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[e][mb][r] = in[(lane + 7 * e + 3 * mb + r) & 511];
        float amax = 0.f;
        const int own = q * 256 + n * 16;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            wn_u4 lh[NB][2], ll[NB][2], oh[NB][2], ol[NB][2];
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                char* blk = lds + (12 + u + 4 * e + 1) * BLK + own;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    lh[e][s] = *reinterpret_cast<const wn_u4*>(blk + s * 1024);
                    ll[e][s] = *reinterpret_cast<const wn_u4*>(blk + PLANE + s * 1024);
                }
            }
            pair_epilogue_n<NB>(W, acc, lh, ll, oh, ol, amax);
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                char* blk = lds + (12 + u + 4 * e + 1) * BLK + own;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // keep the stored words small fp16 pairs so that the loop stays finite
                    *reinterpret_cast<wn_u4*>(blk + s * 1024) = (oh[e][s] & 0x0fff0fffu) | 0x28002800u;
                    *reinterpret_cast<wn_u4*>(blk + PLANE + s * 1024) = (ol[e][s] & 0x03ff03ffu) | 0x10001000u;
                }
                const float d = __uint_as_float((ol[e][0][0] & 0x007fffffu) | 0x3c000000u) - 0.0078125f;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[e][mb][r] = acc[e][mb][r] * 0.5f + d;
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        sink = amax + acc[0][0][0];
    }
    out[blockIdx.x * 512 + threadIdx.x] = sink;
    if (blockIdx.x == 7 && lane == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
void run(float* out, float* in, unsigned long long* cyc) {
    const int iters = 1000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipMemset(cyc, 0, 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), LDS_BYTES, 0, out, in, iters, cyc);
        hipDeviceSynchronize();
    }
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double kmax = 0, emax = 0;
    for (int w = 0; w < 4; ++w) kmax = (double)h[w] > kmax ? (double)h[w] : kmax;
    for (int w = 4; w < 8; ++w) emax = (double)h[w] > emax ? (double)h[w] : emax;
    const char* names[] = {"K team + E team", "K team alone", "E team alone"};
    printf("%-16s NB=%d:  K wave %8.1f cycles per %d-block layer (%d MFMAs = %d pipe cycles)   E wave %8.1f cycles per %d blocks\n",
           names[MODE], NB, kmax / iters, NB, 72 * NB, 72 * NB * 16, emax / iters, NB);
}

}  // namespace

int main() {
    float *out, *in;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&in, 512 * 4);
    hipMalloc(&cyc, 64);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 2000) / 500.f - 2.f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    run<1>(out, in, cyc);
    run<2>(out, in, cyc);
    run<0>(out, in, cyc);
    return 0;
}
