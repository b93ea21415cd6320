// Micro-benchmark: what fp16 MFMA rate does the part SUSTAIN, and at which package power and shader clock?  A pure
// v_mfma_f32_16x16x32_f16 loop (no memory, no LDS, no VALU) on all 1 024 SIMDs for a few seconds per operand pattern, with
// the GPU's hwmon nodes (power1_input, freq1_input) sampled from the host beside it.  Patterns: all-zero operands, one
// constant, random halves rotating through 8 register pairs (an operand changes every instruction, as in the kernels).
// The kernels of this repository run at the 1 400 W package cap; this is the same question for the bare
// matrix pipe: the nominal 2.5 PFLOP/s is the rate AT 2.4 GHz, and the clock is what the power budget leaves.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_power mfma_power.hip     run: ./mfma_power <hwmon dir> [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// MODE 0: both operands change with every instruction; 1: A changes, B is held for 8 instructions; 2: B changes, A is
// held for 8; 3: both held for 8 instructions (only the accumulator changes)
template <int NACC, int NOP, int MODE = 0>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    f4 acc[NACC];
    h8 a[NOP], b[NOP];
    for (int o = 0; o < NOP; ++o)
        for (int i = 0; i < 8; ++i) {
            a[o][i] = (_Float16)in[(threadIdx.x + 17 * o + i) & 1023];
            b[o][i] = (_Float16)in[(threadIdx.x + 29 * o + 8 + i) & 1023];
        }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const int ia = (MODE == 0 || MODE == 1) ? (u + i) % NOP : u % NOP;
                const int ib = (MODE == 0 || MODE == 2) ? (u + 3 * i) % NOP : (u * 3) % NOP;
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia], b[ib], acc[i], 0, 0, 0);
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same loop on the bf16 and fp8 forms of the instruction (random operands): is the energy in the multiplier width?
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef short s8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_bf16(float* out, const float* in, int iters) {
    f4 acc[8];
    s8 a[8], b[8];
    for (int o = 0; o < 8; ++o)
        for (int i = 0; i < 8; ++i) {
            a[o][i] = (short)(__float_as_uint(in[(threadIdx.x + 17 * o + i) & 1023]) >> 16);
            b[o][i] = (short)(__float_as_uint(in[(threadIdx.x + 29 * o + 8 + i) & 1023]) >> 16);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a[(u + i) % 8]), __builtin_bit_cast(b8, b[(u + 3 * i) % 8]), acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// fp16 operands whose low mantissa bits are zero (`keep` of the 10 stored bits kept): does a sparser multiplicand cost less?
template <int KEEP>
__global__ __launch_bounds__(256) void k_trunc(float* out, const float* in, int iters) {
    f4 acc[8];
    h8 a[8], b[8];
    for (int o = 0; o < 8; ++o)
        for (int i = 0; i < 8; ++i) {
            _Float16 x = (_Float16)in[(threadIdx.x + 17 * o + i) & 1023], y = (_Float16)in[(threadIdx.x + 29 * o + 8 + i) & 1023];
            unsigned short xb = __builtin_bit_cast(unsigned short, x) & (unsigned short)(0xffffu << (10 - KEEP));
            unsigned short yb = __builtin_bit_cast(unsigned short, y) & (unsigned short)(0xffffu << (10 - KEEP));
            a[o][i] = __builtin_bit_cast(_Float16, xb);
            b[o][i] = __builtin_bit_cast(_Float16, yb);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + i) % 8], b[(u + 3 * i) % 8], acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the 32x32x16 shape of the same instruction family (4 accumulators of 16 registers): same FLOP rate, half the operand words per FLOP
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_32(float* out, const float* in, int iters) {
    f16v acc[4];
    h8 a[8], b[8];
    for (int o = 0; o < 8; ++o)
        for (int i = 0; i < 8; ++i) {
            a[o][i] = (_Float16)in[(threadIdx.x + 17 * o + i) & 1023];
            b[o][i] = (_Float16)in[(threadIdx.x + 29 * o + 8 + i) & 1023];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + i) % 8], b[(u + 3 * i) % 8], acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static long read_long(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    long v = -1;
    if (fscanf(f, "%ld", &v) != 1) v = -1;
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    const std::string hw = argc > 1 ? argv[1] : "";
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&in, 1024 * 4);
    std::vector<float> h(1024);
    printf("cap %.0f W\n", read_long(hw + "/power1_cap") * 1e-6);
    for (int pat = 0; pat < 3; ++pat) {
        for (int i = 0; i < 1024; ++i)
            h[i] = pat == 0 ? 0.f : pat == 1 ? 0.5f : (float)((i * 2654435761u) % 2000) / 1000.f - 1.0f;
        hipMemcpy(in, h.data(), 1024 * 4, hipMemcpyHostToDevice);
        const int iters = 4000, wg = 1024;            // 4 waves per SIMD
        const double flop = (double)wg * 4 * iters * 16 * 8 * 16384;
        for (int w = 0; w < 100; ++w) hipLaunchKernelGGL((k<8, 8>), dim3(wg), dim3(256), 0, 0, out, in, iters);   // reach the operating point
        hipDeviceSynchronize();
        double psum = 0, fsum = 0;
        int ns = 0, launches = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int w = 0; w < 8; ++w) hipLaunchKernelGGL((k<8, 8>), dim3(wg), dim3(256), 0, 0, out, in, iters);
            launches += 8;
            const long p = read_long(hw + "/power1_input"), f = read_long(hw + "/freq1_input");
            if (p > 0 && f > 0) { psum += p * 1e-6; fsum += f * 1e-6; ++ns; }
            hipDeviceSynchronize();
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double pf = flop * launches / dt / 1e15;
        printf("%-9s %.3f PFLOP/s sustained  %7.1f W  %6.0f MHz  -> %.3f of the 2.5 PFLOP/s nominal, %.3f of the rate at that clock, %.2f pJ/FLOP\n",
               pat == 0 ? "zeros" : pat == 1 ? "constant" : "random", pf, ns ? psum / ns : 0.0, ns ? fsum / ns : 0.0, pf / 2.5,
               ns ? pf / (2.5 * (fsum / ns) / 2400.0) : 0.0, ns ? (psum / ns) / (pf * 1e15) * 1e12 : 0.0);
    }
    // which operand changes between consecutive instructions (random data); modes 4..7: bf16 operands, fp16 operands with 7 / 4 / 1 stored mantissa bits
    for (int mode = (argc > 3 ? 4 : 0); mode < (argc > 3 ? 9 : 4); ++mode) {
        const int iters = 4000, wg = 1024;
        const double flop = (double)wg * 4 * iters * 16 * 8 * 16384;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL((k<8, 8, 0>), dim3(wg), dim3(256), 0, 0, out, in, iters);
            else if (mode == 1) hipLaunchKernelGGL((k<8, 8, 1>), dim3(wg), dim3(256), 0, 0, out, in, iters);
            else if (mode == 2) hipLaunchKernelGGL((k<8, 8, 2>), dim3(wg), dim3(256), 0, 0, out, in, iters);
            else if (mode == 3) hipLaunchKernelGGL((k<8, 8, 3>), dim3(wg), dim3(256), 0, 0, out, in, iters);
            else if (mode == 4) hipLaunchKernelGGL(k_bf16, dim3(wg), dim3(256), 0, 0, out, in, iters);
            else if (mode == 5) hipLaunchKernelGGL(k_trunc<7>, dim3(wg), dim3(256), 0, 0, out, in, iters);
            else if (mode == 6) hipLaunchKernelGGL(k_trunc<4>, dim3(wg), dim3(256), 0, 0, out, in, iters);
            else if (mode == 7) hipLaunchKernelGGL(k_trunc<1>, dim3(wg), dim3(256), 0, 0, out, in, iters);
            else hipLaunchKernelGGL(k_32, dim3(wg), dim3(256), 0, 0, out, in, iters);      // 16 x 4 x 32 768 FLOP per iteration = the same
        };
        for (int w = 0; w < 100; ++w) launch();
        hipDeviceSynchronize();
        double psum = 0, fsum = 0;
        int ns = 0, launches = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int w = 0; w < 8; ++w) launch();
            launches += 8;
            const long p = read_long(hw + "/power1_input"), f = read_long(hw + "/freq1_input");
            if (p > 0 && f > 0) { psum += p * 1e-6; fsum += f * 1e-6; ++ns; }
            hipDeviceSynchronize();
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double pf = flop * launches / dt / 1e15;
        printf("random, %-34s %.3f PFLOP/s  %7.1f W  %6.0f MHz  %.2f pJ/FLOP\n",
               mode == 0 ? "A and B change every instruction" : mode == 1 ? "A changes, B held for 8" : mode == 2 ? "B changes, A held for 8" : mode == 3 ? "A and B held for 8"
               : mode == 4 ? "bf16 operands" : mode == 5 ? "fp16, 7 mantissa bits kept" : mode == 6 ? "fp16, 4 mantissa bits kept" : mode == 7 ? "fp16, 1 mantissa bit kept" : "fp16, 32x32x16 shape",
               pf, ns ? psum / ns : 0.0, ns ? fsum / ns : 0.0, ns ? (psum / ns) / (pf * 1e15) * 1e12 : 0.0);
    }
    return 0;
}
