// Host-side packing helpers for the split-fp16 paths (wn_iaf_h.hip, wn_deconv.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

// ---- host-side fp32 -> fp16 (round to nearest even), independent of host _Float16 support ----
inline uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));   // inf / nan
    if (x < 0x38800000u) {                         // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 126 - (int)(x >> 23);     // 14..24
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t rnd = 1u << (shift - 1);
        const uint32_t rem = m & ((1u << shift) - 1);
        m >>= shift;
        if (rem > rnd || (rem == rnd && (m & 1))) ++m;
        return (uint16_t)(sign | m);
    }
    uint32_t m = x - 0x38000000u;                   // rebias exponent
    const uint32_t rem = m & 0x1fffu;
    m >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (m & 1))) ++m;
    return (uint16_t)(sign | m);
}
inline float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            x = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// power-of-two prescale so that the lo halves of small weights stay normal fp16 numbers
inline float pick_scale(const float* w, size_t nw) {
    float mx = 0.f;
    for (size_t i = 0; i < nw; ++i) mx = std::max(mx, std::fabs(w[i]));
    if (!(mx > 0.f)) return 1.f;
    int k = (int)std::floor(std::log2(16384.0f / mx));
    k = std::max(-8, std::min(k, 14));
    return std::ldexp(1.0f, k);
}

// write the A-fragment words of one (K-step, row block): plane 0 = hi, plane 1 = lo
// wk(e, kg, i16) returns the (prescaled) weight of k-slot (kg, e) for output row i16
template <class F>
void pack_afrag(unsigned* dst, F wk) {
    for (int plane = 0; plane < 2; ++plane)
        for (int lane = 0; lane < 64; ++lane)
            for (int i = 0; i < 4; ++i) {
                const int i16 = lane & 15, kg = lane >> 4;
                uint16_t hh[2];
                for (int p = 0; p < 2; ++p) {
                    const float v = wk(2 * i + p, kg, i16);
                    const uint16_t hi = f2h(v);
                    hh[p] = plane == 0 ? hi : f2h(v - h2f(hi));
                }
                dst[(plane * 64 + lane) * 4 + i] = (uint32_t)hh[0] | ((uint32_t)hh[1] << 16);
            }
}

