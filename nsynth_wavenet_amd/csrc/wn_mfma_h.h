// Device helpers shared by the split-fp16 IAF kernels (wn_iaf_h.hip, wn_iaf_c.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "wn_internal.h"
#include "wn_codec.h"

namespace {


constexpr float EXP_M9 = 1.2340980408667956e-4f;
constexpr float EXP_7 = 1096.6331584284585f;

__device__ inline f4 mfma_h(wn_u4 a, wn_u4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, a), __builtin_bit_cast(wn_h8, b), c, 0, 0, 0);
}
// WN_F16X2 (a MEASUREMENT build, never the shipped library: build.py builds it beside libwnhip.so as libwnhip_f16x2*.so for
// bench.py's labelled `roofline_f16x2` extra): TWO terms per product, wh.xh + wl.xh -- the activations' lo halves are not
// multiplied, i.e. the activations enter with their 11-bit hi part only.  Narrower than the reference's fp32; it prices the
// error budget the contract leaves (1e-3) in microseconds and joules.  Bit 0: the conditioning GEMM, bit 1: the residual
// stack / heads (every mfma3 site).
#ifndef WN_F16X2
#define WN_F16X2 0
#endif
__device__ inline f4 mfma3(wn_u4 ah, wn_u4 al, wn_u4 bh, wn_u4 bl, f4 c) {
    c = mfma_h(ah, bh, c);
#if !(WN_F16X2 & 2)
    c = mfma_h(ah, bl, c);
#else
    (void)bl;
#endif
    return mfma_h(al, bh, c);
}
// sigmoid(a) * tanh(b) (parallel_wavenet.py:246-250) from PRE-SCALED arguments as = -a log2(e), bt = 2 b log2(e) -- the
// layer packs store the gate biases multiplied by those factors (wn_pack_iaf_h) and the kernels fold them into the
// de-scaling constant, so an argument is one FMA on the accumulator.  With ea = 2^as and t = 2^-|bt|:
//   sigmoid(a) = 1 / (1 + ea),   tanh(b) = sign(b) (1 - t) / (1 + t)      ->  ONE reciprocal for the pair,
// three transcendentals per gate value instead of four and 8 plain VALU operations instead of 12 (the epilogue is half
// of a layer kernel's time on gfx950, where VALU work does not hide behind the MFMAs).  t <= 1 never overflows; ea = inf
// (a << 0) gives 1 / inf = 0, the correct limit; |error| ~ 1e-7 like the two-reciprocal form.
constexpr float WN_LOG2E = 1.4426950408889634f;
__device__ inline float gate_scaled(float as, float bt) {
    const float ea = __builtin_amdgcn_exp2f(as);
    const float t = __builtin_amdgcn_exp2f(-fabsf(bt));
    const float r = __builtin_amdgcn_rcpf((1.f + t) * (1.f + ea));     // (NOT fma(t, 1 + ea, 1 + ea): 0 * inf for a << 0, |b| >> 0)
    return __builtin_copysignf(fmaf(-t, r, r), bt);                    // (1 - t) r: r <= 1, t <= 1
}
__device__ inline float sigmoidf_(float a) { return __builtin_amdgcn_rcpf(1.f + __expf(-a)); }
__device__ inline float tanhf_(float a) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * a) + 1.f); }

__device__ inline wn_u2 buf_ld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(wn_u2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ inline void buf_st2(unsigned w0, unsigned w1, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wn_u2, (wn_u2){w0, w1}), r, voff, soff, 0);
}

template <int NWORDS, int NT = 256>
__device__ inline void stage_words(const unsigned* __restrict__ wpack, unsigned* lds) {
    constexpr int NV = NWORDS / 4, NCHUNK = NV / NT, REM = NV - NCHUNK * NT;
    const wn_u4* src = reinterpret_cast<const wn_u4*>(wpack) + threadIdx.x;
    wn_u4* dst = reinterpret_cast<wn_u4*>(lds) + threadIdx.x;
    wn_u4 tmp[NCHUNK + 1];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) tmp[k] = src[k * NT];
    if (REM && (int)threadIdx.x < REM) tmp[NCHUNK] = src[NCHUNK * NT];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) dst[k * NT] = tmp[k];
    if (REM && (int)threadIdx.x < REM) dst[NCHUNK * NT] = tmp[NCHUNK];
    __syncthreads();
}

// ---------------------------------------------------------------------------
// "G4" activation layout of the split-fp16 path.  A lane's MFMA B operand for K-step s is
// four words: pair rows 16s + {0,1,8,9} + 2kg.  Those four rows are stored INTERLEAVED:
//   word(plane, group g = 4s + kg, column t, slot i)  at  ((plane*NG + g)*rowlen + t)*4 + i
// (NG = 8 groups for the 64 residual channels, 32 for the 256 enc channels), so the operand
// is ONE aligned 16-byte load per column -- no register shuffling between load and MFMA --
// and the accumulator layout stores back as full 16-byte words as well (slot = 2*(mb&1)+rp
// of group 4*(mb>>1) + kg).  Same bytes as the fp32 layout; DRAM sees 256-byte runs.
struct HSrc {
    __amdgpu_buffer_rsrc_t rl, re;
    int vo[3];
    int ve;
};

// Tile walk of a persistent workgroup.  Workgroup b runs on XCD b % 8 (observed dispatch
// order, used for speed only -- any placement is correct); each XCD gets one contiguous eighth
// of the tiles so that the causal taps t-d, t-2d of a tile mostly hit lines its own L2 already
// holds (measured 25.5 -> 24.2 us per layer launch at config 2).
struct TileWalk { int first, end, step; };
__device__ inline TileWalk tile_walk(int ntiles) {
#ifndef WN_NO_XCD_TILES
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, per = (ntiles + 7) >> 3;
        return {xcd * per + (int)(blockIdx.x >> 3), min(ntiles, (xcd + 1) * per), (int)(gridDim.x >> 3)};
    }
#endif
    return {(int)blockIdx.x, ntiles, (int)gridDim.x};
}
// the same walk for a workgroup made of `parts` independent 256-thread parts (part p of workgroup w walks like
// workgroup parts * w + p of a grid parts times as large)
__device__ inline TileWalk tile_walk_parts(int ntiles, int parts, int part) {
#ifndef WN_NO_XCD_TILES
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, per = (ntiles + 7) >> 3;
        return {xcd * per + parts * (int)(blockIdx.x >> 3) + part, min(ntiles, (xcd + 1) * per), parts * (int)(gridDim.x >> 3)};
    }
#endif
    return {parts * (int)blockIdx.x + part, ntiles, parts * (int)gridDim.x};
}

// operand words of one K-step: hi and lo plane, HN columns
template <int HN>
struct KOp {
    wn_u4 h[HN], l[HN];
};

#ifndef WN_ENC_AUX
#define WN_ENC_AUX 0
#endif
template <int AUX = 0>
__device__ inline wn_u4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(wn_u4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
// 16-byte store of a residual-stream word.  A wide store reads its data VGPRs some cycles after it issues; the ISA manual
// asks for a wait state before a VALU instruction overwrites them EXCEPT for buffer stores with an SGPR soffset, hipcc's
// hazard recognizer implements that exemption, and on gfx950 it does not hold: the group kernel lost quads of lanes of
// `buffer_store_dwordx4 v[48:51], v115, s[40:43], s75 offen` to the `v_max_f32 v48, ...` the register allocator and the
// scheduler had put right behind it (nondeterministic, nearly every call; profiles/r03_store_hazard.txt).  The asm
// statement behind the store READS the stored registers -- they stay allocated up to it, nothing scheduled in between can
// write them -- and holds two wait states.  It lives HERE, in the one helper every such store goes through (round 3 had
// it at the call sites: the next new call site would have been unguarded).  scripts/audit_store_hazard.py (run by
// tests/test_host.py) still reads the compiler's assembly of every kernel for the pattern.
#ifndef WN_G_ST_AUX
#define WN_G_ST_AUX 0
#endif
template <int AUX = 0>
__device__ inline void buf_st4(wn_u4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, AUX);
    asm volatile("s_nop 1" ::"v"(v));
}


// Operands of the FIRST residual layer of a flow (dilation 1), computed from the flow input instead
// of being loaded: the layer's input is l0 = start_conv(shift_right(x)) (parallel_wavenet.py:222-225),
// l0[c][t'] = b[c] + w0[c] x[t'-3] + w1[c] x[t'-2] + w2[c] x[t'-1], and 0 left of the utterance.
// xv[j] = x[t-5+j]; wq[c] = (w0, w1, w2, b) in LDS; bc[2*tap + s] = K-step operands of tap t-2+tap.
constexpr int IAF_START_LDS_WORDS = 64 * 4;
__device__ inline void first_layer_operands(const float (&xv)[5], long long t, int kg, const f4* __restrict__ wq,
                                            KOp<1> (&bc)[6], float& amax) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 2 * (16 * s + 8 * (i >> 1) + 2 * kg + (i & 1));
            const f4 wa = wq[c], wb = wq[c + 1];
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                float v0 = wa[3] + wa[0] * xv[tap] + wa[1] * xv[tap + 1] + wa[2] * xv[tap + 2];
                float v1 = wb[3] + wb[0] * xv[tap] + wb[1] * xv[tap + 1] + wb[2] * xv[tap + 2];
                if (t - 2 + tap < 0) v0 = v1 = 0.f;
                unsigned hw, lw;
                wn_split_pair_t(v0, v1, hw, lw, amax);
                bc[2 * tap + s].h[0][i] = hw;
                bc[2 * tap + s].l[0][i] = lw;
            }
        }
}
// stage (w0, w1, w2, b) per channel from the start-conv block w[3][64] | b[64]
__device__ inline void stage_start_weights(const float* __restrict__ wb, f4* wq) {
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        wq[c] = (f4){wb[c], wb[64 + c], wb[128 + c], wb[192 + c]};
    }
}

__device__ inline float softplus_tf(float p) {
    const float thr = -13.942384719848633f;      // tf.nn.softplus: log(eps) + 2
    if (p > -thr) return p;
    if (p < thr) return expf(p);
    return log1pf(expf(p));
}

}  // namespace
