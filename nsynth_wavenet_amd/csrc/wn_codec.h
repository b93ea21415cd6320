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
