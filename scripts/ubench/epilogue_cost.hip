// Micro-benchmark (dev tool, not product): what does the residual-layer epilogue of the group kernel cost per
// 16-sample block, and which part of it?  Runs the REAL device code (wn_iaf_c.h: pair_epilogue / pair_epilogue_n) on
// accumulators in registers with the layer tail in LDS, 1 / 2 / 3 waves per SIMD, and prints cycles per block for the
// slowest and the fastest wave of a workgroup.  Variants are selected at compile time: -DEPI_VARIANT=<n>
//   0 the shipped pair_epilogue, blocks one after the other        1 pair_epilogue_n<2> (two blocks in lock step)
//   2 gate only (no residual 1x1, no join / split of the output)   3 residual part only (gate replaced by a copy)
//   6 residual add through the MFMA (l enters as an extra K-step against an identity fragment held in registers:
//     8 more MFMAs per block instead of 16 v_fma_mix_f32 + 16 v_add_f32)
//   7 fp16 split with plain conversions (cvt / sub / cvt_pk) instead of v_fma_mix{lo,hi}_f16       8 = 6 + 7
//   (variants 0 - 3 use the product's codec -- since round 4 the plain split; 6 shows round 3's fused split for comparison)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I nsynth_wavenet_amd/csrc -DEPI_VARIANT=0 -o epilogue_cost epilogue_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include "wn_iaf_c.h"
#include "pair_epilogue_n.h"

#ifndef EPI_VARIANT
#define EPI_VARIANT 0
#endif

namespace {

__device__ inline void epi_variant(const PairLayer& w, const f4 (&acc)[4], const wn_u4 (&lh)[2], const wn_u4 (&ll)[2],
                                   wn_u4 (&oh)[2], wn_u4 (&ol)[2], float& amax) {
#if EPI_VARIANT == 2
    float g[2][4];
#pragma unroll
    for (int mg = 0; mg < 2; ++mg)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            g[mg][r] = gate_scaled(fmaf(acc[mg][r], -WN_LOG2E * w.inv_m, w.bg[mg * 4 + r]),
                                   fmaf(acc[mg + 2][r], 2.f * WN_LOG2E * w.inv_m, w.bg[(mg + 2) * 4 + r]));
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            oh[s][i] = lh[s][i] ^ __float_as_uint(g[s][i]);
            ol[s][i] = ll[s][i];
        }
#elif EPI_VARIANT == 3
    wn_u4 gh, gl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gh[i] = __float_as_uint(acc[0][i]);
        gl[i] = __float_as_uint(acc[1][i]);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const f4 rc = mfma3(w.PRl[(mb * 2 + 0) * 64], w.PRl[(mb * 2 + 1) * 64], gh, gl, (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            float l0, l1;
            wn_join_pair(lh[mb >> 1][(mb & 1) * 2 + rp], ll[mb >> 1][(mb & 1) * 2 + rp], l0, l1);
            const float v0 = l0 + fmaf(rc[2 * rp], w.inv_r, w.br[mb * 4 + 2 * rp]);
            const float v1 = l1 + fmaf(rc[2 * rp + 1], w.inv_r, w.br[mb * 4 + 2 * rp + 1]);
            unsigned hw, lw;
            wn_split_pair_t(v0, v1, hw, lw, amax);
            oh[mb >> 1][(mb & 1) * 2 + rp] = hw;
            ol[mb >> 1][(mb & 1) * 2 + rp] = lw;
        }
    }
#elif EPI_VARIANT >= 6
    constexpr bool JOIN_MFMA = EPI_VARIANT == 6 || EPI_VARIANT == 8, PLAIN = EPI_VARIANT >= 7;
    auto split = [&](float x0, float x1, unsigned& hi, unsigned& lo) {
        if (PLAIN) {
            wn_split_pair(x0, x1, hi, lo);                  // the product's split: convert / subtract / convert
        } else {                                            // round 3's form
            hi = __builtin_bit_cast(unsigned, (wn_h2){(_Float16)x0, (_Float16)x1});
            unsigned l;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(x0));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(x1));
            lo = l;
            asm volatile("s_nop 1" : "+v"(lo));
        }
    };
    float g[2][4];
#pragma unroll
    for (int mg = 0; mg < 2; ++mg)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            g[mg][r] = gate_scaled(fmaf(acc[mg][r], -WN_LOG2E * w.inv_m, w.bg[mg * 4 + r]),
                                   fmaf(acc[mg + 2][r], 2.f * WN_LOG2E * w.inv_m, w.bg[(mg + 2) * 4 + r]));
    wn_u4 gh, gl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned hw, lw;
        split(g[i >> 1][(i & 1) * 2], g[i >> 1][(i & 1) * 2 + 1], hw, lw);
        gh[i] = hw;
        gl[i] = lw;
    }
    // identity fragments (rows of an even / odd 16-row block against the 32 channels of their K-step): lane (m, kg)
    // holds a single 1.0 (times the residual prescale) where kg == m >> 2, at word (m & 3) >> 1 (+ 2 for odd blocks)
    const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
    const unsigned one = (kg == (m >> 2)) ? ((m & 1) ? 0x44000000u : 0x00004400u) : 0u;      // fp16 4.0 = 1 / inv_r here
    wn_u4 Ie = {0u, 0u, 0u, 0u}, Io = {0u, 0u, 0u, 0u};
    Ie[(m & 3) >> 1] = one;
    Io[2 + ((m & 3) >> 1)] = one;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f4 rc = mfma3(w.PRl[(mb * 2 + 0) * 64], w.PRl[(mb * 2 + 1) * 64], gh, gl, (f4){0.f, 0.f, 0.f, 0.f});
        if (JOIN_MFMA) {
            rc = mfma_h((mb & 1) ? Io : Ie, lh[mb >> 1], rc);
            rc = mfma_h((mb & 1) ? Io : Ie, ll[mb >> 1], rc);
        }
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            float v0 = fmaf(rc[2 * rp], w.inv_r, w.br[mb * 4 + 2 * rp]);
            float v1 = fmaf(rc[2 * rp + 1], w.inv_r, w.br[mb * 4 + 2 * rp + 1]);
            if (!JOIN_MFMA) {
                float l0, l1;
                wn_join_pair(lh[mb >> 1][(mb & 1) * 2 + rp], ll[mb >> 1][(mb & 1) * 2 + rp], l0, l1);
                v0 += l0;
                v1 += l1;
            }
            unsigned hw, lw;
            wn_range_track(amax, v0, v1);
            split(v0, v1, hw, lw);
            oh[mb >> 1][(mb & 1) * 2 + rp] = hw;
            ol[mb >> 1][(mb & 1) * 2 + rp] = lw;
        }
    }
#else
    pair_epilogue(w, acc, lh, ll, oh, ol, amax);
#endif
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void k(float* out, const float* in, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned tail[LC_TAIL_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4;
    for (int i = threadIdx.x; i < LC_TAIL_WORDS; i += NW * 64) {
        // residual fragments: small fp16 pairs; biases / scales: small floats
        tail[i] = i < IAF_PR_FLOATS ? 0x14001800u + (unsigned)(i & 7) : __float_as_uint(0.01f * (float)(i & 15));
    }
    __syncthreads();
    const float* tailf = reinterpret_cast<const float*>(tail);
    PairLayer W;
    W.Pl = nullptr;
    W.PRl = reinterpret_cast<const wn_u4*>(tail) + lane;
    W.bg = tailf + IAF_PR_FLOATS + q * 16;
    W.br = W.bg + 64;
    W.inv_m = 0.5f;
    W.inv_r = 0.25f;
    f4 acc[2][4];
    wn_u4 lh[2][2], ll[2][2], oh[2][2], ol[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[e][mb][r] = in[(lane + 7 * e + 3 * mb + r) & 511];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lh[e][s][i] = 0x2c003000u + lane + i;
                ll[e][s][i] = 0x10001400u + 2 * lane + s;
            }
    }
    float amax = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#if EPI_VARIANT == 1
        pair_epilogue_n<2>(W, acc, lh, ll, oh, ol, amax);
#else
        epi_variant(W, acc[0], lh[0], ll[0], oh[0], ol[0], amax);
        epi_variant(W, acc[1], lh[1], ll[1], oh[1], ol[1], amax);
#endif
        // feed the outputs back so that nothing is loop-invariant (cheap: 8 xors per block)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                lh[e][s] = (lh[e][s] & 0x0fff0fffu) | ((oh[e][s] & 0x00010001u) << 1) | 0x28002800u;
            }
        // ... every accumulator moves (16 v_add per block in every variant: otherwise hipcc hoists most of the gate)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float d = __uint_as_float((ol[e][0][0] & 0x007fffffu) | 0x3c000000u) - 0.0078125f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[e][mb][r] = acc[e][mb][r] * 0.5f + d;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = amax;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i) s += __uint_as_float(oh[e][i][0] ^ ol[e][i][3]) + acc[e][i][0];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (blockIdx.x == 5 && lane == 0) cyc[wave] = t1 - t0;
}

template <int NW>
void run(float* out, float* in, unsigned long long* cyc) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<NW>, dim3(256), dim3(NW * 64), 0, 0, out, in, iters, cyc);
        hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mn = 1e30, mx = 0;
    for (int w = 0; w < NW; ++w) {
        const double c = (double)h[w] / iters / 2.0;
        mn = c < mn ? c : mn;
        mx = c > mx ? c : mx;
    }
    // cycles per block as seen by one wave, and per SIMD (NW / 4 waves share a SIMD)
    printf("variant %d  %2d waves (%d per SIMD): cycles per block  fastest wave %7.1f  slowest %7.1f   -> per SIMD-block %7.1f\n",
           EPI_VARIANT, NW, NW / 4, mn, mx, mx / (NW / 4));
}

}  // namespace

int main() {
    float *out, *in;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 768 * 4);
    hipMalloc(&in, 512 * 4);
    hipMalloc(&cyc, 16 * 8);
    std::vector<float> h(512);
    for (int i = 0; i < 512; ++i) h[i] = (float)((i * 2654435761u) % 2000) / 500.f - 2.f;
    hipMemcpy(in, h.data(), 512 * 4, hipMemcpyHostToDevice);
    run<4>(out, in, cyc);
    run<8>(out, in, cyc);
    run<12>(out, in, cyc);
    return 0;
}
