// Device-side audio codecs shared by the IAF and AR units.
//   auxilaries/utils.py:72-105 (mu_law), :108-139 (inv_mu_law),
//   :142-159 (cast_quantize / inv_cast_quantize),
//   wavenet/parallel_wavenet.py:347-359 (_clip_quant_scale).
#pragma once
#include <hip/hip_runtime.h>

// inverse of the quantiser: index q in [-Q/2, Q/2) -> float sample
__device__ inline float wn_dequant(int qi, int Q, int mu) {
    const float qf = (float)qi;
    if (!mu) return qf / ((float)Q * 0.5f);                 // utils.py:157-159
    // utils.py:108-122: s=(q+.5)*2/256; sign(s)/255*(256^|s|-1); exactly 0 where q==0
    const float s = (qf + 0.5f) * 2.0f / 256.0f;
    const float m = exp2f(8.0f * fabsf(s)) - 1.0f;
    const float sg = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
    return qi == 0 ? 0.f : sg / 255.0f * m;
}

// clip to [-1, 1-2/Q], int32(floor(x*Q/2))  (parallel_wavenet.py:349, utils.py:153-154)
__device__ inline int wn_clip_quantize(float y, int Q) {
    const float hi = 1.0f - 2.0f / (float)Q;
    const float yc = fminf(fmaxf(y, -1.0f), hi);
    return (int)floorf(yc * (float)Q * 0.5f);               // power-of-two scaling: exact
}

// mu-law encode then /128: the teacher's input scaling (wavenet.py:412-418, utils.py:72-87)
__device__ inline float wn_mu_law_scaled(float x) {
    const float sg = x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
    const float o = sg * logf(1.0f + 255.0f * fabsf(x)) / 5.545177444479562f;
    return floorf(o * 128.0f) / 128.0f;
}

__device__ inline void wn_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__device__ inline float wn_u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }   // [0,1)

// ---------------------------------------------------------------------------
// Split-fp16 ("f16x3") operands.  A float x is carried as two halves hi = f16(x),
// lo = f16(x - hi) (22 significant bits); a product a*b is evaluated on the fp16 MFMA as
// ah*bh + ah*bl + al*bh with fp32 accumulation (the dropped al*bl term is 2^-22 relative).
// fp16 x fp16 products are exact in fp32, so the only errors are the operand split and the
// fp32 accumulation order.  On gfx950 the fp16 MFMA runs at 16x the fp32-MFMA rate, so
// three of them are still 5.3x faster than one fp32 MFMA.
// Activations live in HBM as "pair planes": for C channels, hi plane [C/2][time] and lo
// plane [C/2][time] of 32-bit words, word (cp, t) = {half(ch 2cp), half(ch 2cp+1)} -- the
// same bytes as fp32 [C][time], already in the k-pair order the MFMA B operand wants.
typedef _Float16 wn_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 wn_h2 __attribute__((ext_vector_type(2)));
typedef unsigned wn_u4 __attribute__((ext_vector_type(4)));
typedef unsigned wn_u2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> hi word {f16(x0), f16(x1)} and lo word of the remainders, in plain C++ (convert / subtract / convert):
// x - hi is exact in fp32, so the remainder is rounded ONCE to f16.  Round 3 used v_fma_mix{lo,hi}_f16 from inline asm
// (the same bits in fewer instructions, scripts/ubench/split_codec.hip); round 4 measured what those cost on gfx950:
// a v_fma_mix* occupies a SIMD ~8.6 cycles like a transcendental, a plain VALU operation ~2.7
// (profiles/r04_issue_overlap_ubench.txt; the residual epilogue alone 1 276 -> 1 198 cycles per block, 1 061 with its
// blocks in lock step: profiles/r04_epilogue_cost.txt).  And the compiler SEES these producers: the wait states
// gfx950 needs between a VALU write and an MFMA that reads the register (scripts/ubench/valu_to_mfma.hip) are inserted
// by its hazard recognizer for every consumer, present and future -- the use-site fences the asm form needed are gone.
typedef float wn_f2 __attribute__((ext_vector_type(2)));
__device__ inline void wn_split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    // All in two-element vectors, so that the hi word is converted ONCE, packed (v_cvt_pk_f16_f32), and the lo halves are
    // computed against the halves of that very word.  (With scalar conversions in the source the compiler -- without the
    // SLP vectorizer, build.py -- converts twice: packed for the word, scalar v_cvt_f16_f32 for the differences, and the
    // two instructions do not agree on fp16 denormals: hi + lo was then off by up to 6e-5 for small values, 100 times
    // the codec's error; tests/test_gpu_teacher.py caught it.)
    // The differences come from v_fma_mix_f32 reading the halves of that word in place (x - float(half) in ONE instruction,
    // exact: the difference of a float and its own 11-bit rounding fits 24 bits) instead of a conversion and a subtraction:
    // a wave-wide VALU instruction costs ~1.1 nJ on this part and the workload runs at the power cap.
    const wn_f2 x = {x0, x1};
    const wn_h2 h = __builtin_convertvector(x, wn_h2);
    hi = __builtin_bit_cast(unsigned, h);
    wn_f2 d;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d.x) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d.y) : "v"(hi), "v"(x1));
    const wn_h2 l = __builtin_convertvector(d, wn_h2);
    lo = __builtin_bit_cast(unsigned, l);
}

// Range guard of the split representation: hi = f16(x) is +-inf from |x| >= 65520 on (and the lo half NaN), where the
// reference's fp32 graph is still finite.  Every kernel that PRODUCES split words keeps the running maximum of the
// magnitudes it splits (one v_max3_f32 per pair) and raises bit 0 of the call's status word when that reaches the
// largest finite half; wn_iaf_generate then NaN-poisons its outputs and wn_iaf_range_status reports WN_ERANGE, so the
// caller re-runs on the fp32-MFMA form (the Python Engine does that by itself).  NaNs do not raise the maximum: one
// can only come out of an earlier inf, which was flagged where it was produced.
constexpr float WN_HALF_MAX = 65504.f;
__device__ inline void wn_range_track(float& amax, float x0, float x1) { amax = __builtin_fmaxf(__builtin_fmaxf(amax, fabsf(x0)), fabsf(x1)); }   // one v_max3_f32
__device__ inline void wn_range_flag(float amax, unsigned* status) {
    if (status && !(amax < WN_HALF_MAX)) atomicOr(status, 1u);
}
__device__ inline void wn_split_pair_t(float x0, float x1, unsigned& hi, unsigned& lo, float& amax) {
    wn_range_track(amax, x0, x1);
    wn_split_pair(x0, x1, hi, lo);
}

// The join stays on v_fma_mix_f32 (one instruction per value; its result feeds VALU arithmetic, never an MFMA operand
// directly, so no hardware hazard hides behind the asm statement).
__device__ inline void wn_join_pair(unsigned hi, unsigned lo, float& x0, float& x1) {
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(x0) : "v"(hi), "v"(lo));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x1) : "v"(hi), "v"(lo));
}
