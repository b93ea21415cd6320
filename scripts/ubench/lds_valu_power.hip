// Micro-benchmark: what the NON-matrix work of the kernels costs in energy.  One instruction class per row, on all 1 024 SIMDs
// (4 waves each) for a few seconds, random operands, hwmon power / clock sampled beside it; pJ per unit = (package power - power of
// the idle, clocked chip) / units per second.  Rows: LDS reads (16 B, 8 B per lane) and writes, fp32 FMA, packed fp32 FMA,
// v_exp_f32 / v_rcp_f32 (the gate), the fp32 -> fp16 hi/lo split (cvt, cvt back, sub, cvt: the codec of the G4 words), and the
// fp16 MFMA for scale.  Together with mem_power.hip (HBM, Infinity Cache, L2) and mfma_power.hip this prices every item of the
// group kernel's energy.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lds_valu_power lds_valu_power.hip     run: ./lds_valu_power <hwmon dir> [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

enum { K_IDLE, K_LDS_R16, K_LDS_R8, K_LDS_W16, K_FMA, K_PKFMA, K_EXP, K_RCP, K_SPLIT, K_MFMA, K_N };

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    __shared__ f4 lds[4096];                                   // 64 KB
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) lds[i] = (f4){in[i & 1023], in[(i + 1) & 1023], in[(i + 2) & 1023], in[(i + 3) & 1023]};
    __syncthreads();
    float s = 0;
    if constexpr (KIND == K_IDLE) {
        for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    } else if constexpr (KIND == K_LDS_R16) {                 // bare reads: fixed address, immediate offsets, results dropped (no VALU beside them)
        const unsigned addr = t * 16;
        f4 a0, a1, a2, a3, a4, a5, a6, a7;
        for (int it = 0; it < iters; ++it)
            asm volatile(
                "ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                "ds_read_b128 %4, %8 offset:16384\n ds_read_b128 %5, %8 offset:20480\n ds_read_b128 %6, %8 offset:24576\n"
                "ds_read_b128 %7, %8 offset:28672\n s_waitcnt lgkmcnt(0)"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(addr));
        s = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + a6[2] + a7[3];
    } else if constexpr (KIND == K_LDS_R8) {
        const unsigned addr = t * 8;
        f2 a0, a1, a2, a3, a4, a5, a6, a7;
        for (int it = 0; it < iters; ++it)
            asm volatile(
                "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:2048\n ds_read_b64 %2, %8 offset:4096\n ds_read_b64 %3, %8 offset:6144\n"
                "ds_read_b64 %4, %8 offset:8192\n ds_read_b64 %5, %8 offset:10240\n ds_read_b64 %6, %8 offset:12288\n"
                "ds_read_b64 %7, %8 offset:14336\n s_waitcnt lgkmcnt(0)"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(addr));
        s = a0[0] + a1[1] + a2[0] + a3[1] + a4[0] + a5[1] + a6[0] + a7[1];
    } else if constexpr (KIND == K_LDS_W16) {
        const unsigned addr = t * 16;
        const f4 v0 = lds[t], v1 = lds[(t + 77) & 4095];
        __syncthreads();
        for (int it = 0; it < iters; ++it)
            asm volatile(
                "ds_write_b128 %0, %1\n ds_write_b128 %0, %2 offset:4096\n ds_write_b128 %0, %1 offset:8192\n ds_write_b128 %0, %2 offset:12288\n"
                "ds_write_b128 %0, %2 offset:16384\n ds_write_b128 %0, %1 offset:20480\n ds_write_b128 %0, %2 offset:24576\n"
                "ds_write_b128 %0, %1 offset:28672\n s_waitcnt lgkmcnt(0)"
                : : "v"(addr), "v"(v0), "v"(v1) : "memory");
        __syncthreads();
        s = lds[t][0];
    } else if constexpr (KIND == K_FMA) {
        float a[16];
        for (int u = 0; u < 16; ++u) a[u] = in[(t + u) & 1023];
        const float m = in[t & 1023] * 0.5f + 0.3f, c = in[(t + 5) & 1023] * 0.01f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(m), "v"(c));
        for (int u = 0; u < 16; ++u) s += a[u];
    } else if constexpr (KIND == K_PKFMA) {
        f2 a[8];
        for (int u = 0; u < 8; ++u) a[u] = (f2){in[(t + u) & 1023], in[(t + u + 9) & 1023]};
        const f2 m = {in[t & 1023] * 0.5f + 0.3f, in[(t + 1) & 1023] * 0.5f - 0.2f}, c = {in[(t + 5) & 1023] * 0.01f, 0.02f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(m), "v"(c));
        for (int u = 0; u < 8; ++u) s += a[u][0] + a[u][1];
    } else if constexpr (KIND == K_EXP) {
        float a[16];
        for (int u = 0; u < 16; ++u) a[u] = in[(t + u) & 1023];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_exp_f32 %0, %1" : "=v"(a[u]) : "v"(a[(u + 1) & 15]));
        for (int u = 0; u < 16; ++u) s += a[u];
    } else if constexpr (KIND == K_RCP) {
        float a[16];
        for (int u = 0; u < 16; ++u) a[u] = in[(t + u) & 1023] + 1.5f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[u]) : "v"(a[(u + 1) & 15]));
        for (int u = 0; u < 16; ++u) s += a[u];
    } else if constexpr (KIND == K_SPLIT) {                     // 8 values -> 8 (hi, lo) halves: 2 cvt_pkrtz-class + cvt back + sub per pair
        float a[8];
        for (int u = 0; u < 8; ++u) a[u] = in[(t + u) & 1023];
        unsigned acc = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                unsigned hi, lo;
                float r0, r1;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a[u]), "v"(a[u + 1]));
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r0) : "v"(hi));
                asm volatile("v_cvt_f32_f16 %0, %1 src0_sel:WORD_1" : "=v"(r1) : "v"(hi));
                asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r0) : "v"(a[u]));
                asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r1) : "v"(a[u + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
                acc ^= hi ^ lo;
                a[u] += 0.001f;
            }
        }
        s = (float)acc;
    } else if constexpr (KIND == K_MFMA) {
        f4 acc[8];
        h8 a[8], b[8];
        for (int o = 0; o < 8; ++o)
            for (int i = 0; i < 8; ++i) {
                a[o][i] = (_Float16)in[(t + 17 * o + i) & 1023];
                b[o][i] = (_Float16)in[(t + 29 * o + 8 + i) & 1023];
            }
        for (int i = 0; i < 8; ++i) acc[i] = (f4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(u + i) % 8], b[(u + 3 * i) % 8], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * 256 + t] = s;
}

static long read_long(const std::string& p) {
    FILE* f = fopen(p.c_str(), "r");
    if (!f) return -1;
    long v = -1;
    if (fscanf(f, "%ld", &v) != 1) v = -1;
    fclose(f);
    return v;
}

template <int KIND>
static void launch(float* out, const float* in, int iters) { hipLaunchKernelGGL((k<KIND>), dim3(1024), dim3(256), 0, 0, out, in, iters); }

int main(int argc, char** argv) {
    const std::string hw = argc > 1 ? argv[1] : "";
    const double seconds = argc > 2 ? atof(argv[2]) : 2.5;
    float *out, *in;
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&in, 1024 * 4);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.0f;
    hipMemcpy(in, h.data(), 1024 * 4, hipMemcpyHostToDevice);
    struct Row { const char* tag; void (*fn)(float*, const float*, int); int iters; double units; const char* unit; };
    const double W = 1024.0 * 4;                 // waves per launch
    const Row rows[] = {
        {"busy-idle (s_sleep loops)", launch<K_IDLE>, 2000, 0, ""},
        {"LDS read 16 B per lane", launch<K_LDS_R16>, 4000, W * 64 * 16 * 8, "B"},
        {"LDS read 8 B per lane", launch<K_LDS_R8>, 4000, W * 64 * 8 * 8, "B"},
        {"LDS write 16 B per lane", launch<K_LDS_W16>, 4000, W * 64 * 16 * 8, "B"},
        {"v_fma_f32 (lane-op)", launch<K_FMA>, 4000, W * 64 * 16, "lane-op"},
        {"v_pk_fma_f32 (2 FMAs per lane-op)", launch<K_PKFMA>, 4000, W * 64 * 8, "lane-op"},
        {"v_exp_f32 (lane-op)", launch<K_EXP>, 4000, W * 64 * 16, "lane-op"},
        {"v_rcp_f32 (lane-op)", launch<K_RCP>, 4000, W * 64 * 16, "lane-op"},
        {"fp32 -> fp16 hi/lo split (per value)", launch<K_SPLIT>, 4000, W * 64 * 8, "value"},
        {"fp16 MFMA 16x16x32 (per FLOP)", launch<K_MFMA>, 4000, W * 16 * 16384, "FLOP"},
    };
    double p_idle = 0;
    printf("cap %.0f W\n", read_long(hw + "/power1_cap") * 1e-6);
    for (const Row& r : rows) {
        for (int w = 0; w < 50; ++w) r.fn(out, in, r.iters);
        hipDeviceSynchronize();
        double psum = 0, fsum = 0;
        int ns = 0;
        long launches = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int w = 0; w < 8; ++w) r.fn(out, in, r.iters);
            launches += 8;
            const long p = read_long(hw + "/power1_input"), f = read_long(hw + "/freq1_input");
            if (p > 0 && f > 0) { psum += p * 1e-6; fsum += f * 1e-6; ++ns; }
            hipDeviceSynchronize();
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double P = ns ? psum / ns : 0, F = ns ? fsum / ns : 0;
        if (r.units == 0) { p_idle = P; printf("%-40s %7.1f W  %6.0f MHz\n", r.tag, P, F); continue; }
        const double ups = r.units * r.iters * launches / dt;
        printf("%-40s %9.3f T%s/s  %7.1f W  %6.0f MHz  -> %7.2f pJ per %s above the busy-idle chip\n", r.tag, ups / 1e12, r.unit, P, F,
               (P - p_idle) / ups * 1e12, r.unit);
    }
    return 0;
}
