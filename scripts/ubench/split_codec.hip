// Bitwise check: split/join of wn_codec.h written with v_fma_mix* against the plain conversions.
//   hipcc --offload-arch=gfx950 -O3 -I nsynth_wavenet_amd/csrc scripts/ubench/split_codec.hip -o scripts/ubench/split_codec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "wn_codec.h"

__device__ inline void split_ref(float x0, float x1, unsigned& hi, unsigned& lo) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    hi = __builtin_bit_cast(unsigned, (wn_h2){h0, h1});
    lo = __builtin_bit_cast(unsigned, (wn_h2){l0, l1});
}
__device__ inline void join_ref(unsigned hi, unsigned lo, float& x0, float& x1) {
    const wn_h2 h = __builtin_bit_cast(wn_h2, hi), l = __builtin_bit_cast(wn_h2, lo);
    x0 = (float)h[0] + (float)l[0];
    x1 = (float)h[1] + (float)l[1];
}

__global__ void check(const float* x, int n, unsigned long long* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    unsigned h0, l0, h1, l1;
    split_ref(a, b, h0, l0);
    wn_split_pair(a, b, h1, l1);
    float r0, r1, s0, s1;
    join_ref(h0, l0, r0, r1);
    wn_join_pair(h0, l0, s0, s1);
    const bool ok = h0 == h1 && l0 == l1 && __float_as_uint(r0) == __float_as_uint(s0) &&
                    __float_as_uint(r1) == __float_as_uint(s1);
    if (!ok) {
        if (atomicAdd(bad, 1ull) < 10)
            printf("mismatch x=(%a,%a) hi %08x/%08x lo %08x/%08x join (%a,%a)/(%a,%a)\n", a, b, h0, h1, l0, l1, r0, r1, s0, s1);
    }
}

int main() {
    const int n = 1 << 24;
    std::vector<float> x(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t bits = (uint32_t)(s >> 16);
        const int mode = i & 7;
        float v;
        if (mode < 4) {                       // every exponent the f16 range and beyond sees
            const uint32_t e = 80 + (bits >> 23) % 70;            // 2^-47 .. 2^22
            bits = (bits & 0x807fffffu) | (e << 23);
            memcpy(&v, &bits, 4);
        } else if (mode < 6) {
            v = ((int32_t)bits) * (1.0f / 2147483648.0f) * 8.0f;   // activations' range
        } else if (mode == 6) {
            v = ((int32_t)bits) * (1.0f / 2147483648.0f) * 1e-4f;  // lo halves subnormal in f16
        } else {
            v = (bits & 1) ? 0.f : -0.f;
        }
        x[i] = v;
    }
    float* dx; unsigned long long* dbad; unsigned long long bad = 0;
    hipMalloc(&dx, n * 4); hipMalloc(&dbad, 8);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbad, &bad, 8, hipMemcpyHostToDevice);
    check<<<n / 2 / 256, 256>>>(dx, n, dbad);
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
    printf("pairs checked %d, mismatches %llu\n", n / 2, bad);
    return bad != 0;
}
