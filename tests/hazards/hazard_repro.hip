// Reproducers of the three gfx950 hazards hipcc's hazard recognizer does not cover (profiles/r03_store_hazard.txt),
// as known-answer programs: tests/test_gpu_hazards.py builds this file with hipcc on the GPU box and requires that the
// UNGUARDED instruction sequences still go wrong and the guarded ones are exact -- so that a compiler or firmware update
// that changes either rule is noticed, in both directions.  (Test infrastructure, hand-placed asm; not product code.)
//   prints:  mfma0 <bad> <n>   mfma2 <bad> <n>     VALU write -> MFMA read of the register, 0 / 2 wait states between
//            store0 <bad> <n>  store2 <bad> <n>    four back-to-back buffer_store_dwordx4 (SGPR soffset), then a VALU write of
//                                                  the last store's data registers, 0 / 2 wait states between
//            pksel0 <bad> <n>  pksel1 <bad> <n>    v_pk_fma_f32 ... op_sel:[0,1,0] (high register of src1 for both lanes) with
//            pklow0 <bad> <n>                      an MFMA 0 / 1 instruction slots behind it; the op_sel_hi form (low register
//                                                  for both lanes) with the MFMA directly behind
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// ---- 1. an MFMA that reads a VGPR right behind the VALU instruction that wrote it ----
template <int WAIT>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + 0.001f * lane;
    for (int it = 0; it < iters; ++it) {
        const float x0 = 0.37f * (lane + 1) + 1e-3f * it, x1 = -0.11f * (lane + 3) + 2e-3f * it;
        float l;
        if (WAIT == 0)
            asm volatile("v_add_f32 %0, %2, %3\n\tv_mfma_f32_4x4x1_16b_f32 %1, %4, %0, %1" : "=&v"(l), "+v"(acc) : "v"(x0), "v"(x1), "v"(a));
        else
            asm volatile("v_add_f32 %0, %2, %3\n\ts_nop %5\n\tv_mfma_f32_4x4x1_16b_f32 %1, %4, %0, %1"
                         : "=&v"(l), "+v"(acc) : "v"(x0), "v"(x1), "v"(a), "n"(WAIT - 1));
        asm volatile("s_nop 7\n\ts_nop 7");
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // keep the accumulator finite: mantissa bits only
            unsigned u = __float_as_uint(acc[i]);
            u = (u & 0x007fffffu) | 0x3f800000u;
            acc[i] = __uint_as_float(u) - 1.0f;
        }
    }
    for (int i = 0; i < 4; ++i) out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + i] = acc[i];
}

// ---- 2. a VALU write of a wide store's data registers right behind a run of four stores ----
// rows of 64 lanes x 16 B; the four stores of a wave go to four rows `rowstride` bytes apart (64 different lines each)
template <int WAIT>
__global__ __launch_bounds__(256) void k_store(unsigned* out, int rows_per_wave, int rowstride_bytes, int iters) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7ffffff0, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        const int row0 = wave * rows_per_wave + 4 * it;
        const unsigned seed = (unsigned)(row0 * 64 + lane) * 2654435761u;
        const int vo = lane * 16;
        const int s0 = __builtin_amdgcn_readfirstlane(row0 * rowstride_bytes);
        const int s1 = s0 + rowstride_bytes, s2 = s1 + rowstride_bytes, s3 = s2 + rowstride_bytes;
        // v[40:55] = sixteen words derived from the seed; the fourth store's data v[40:43] is overwritten behind it
        asm volatile(
            "v_add_u32 v40, %0, 0\n\tv_add_u32 v41, %0, 1\n\tv_add_u32 v42, %0, 2\n\tv_add_u32 v43, %0, 3\n\t"
            "v_add_u32 v44, %0, 4\n\tv_add_u32 v45, %0, 5\n\tv_add_u32 v46, %0, 6\n\tv_add_u32 v47, %0, 7\n\t"
            "v_add_u32 v48, %0, 8\n\tv_add_u32 v49, %0, 9\n\tv_add_u32 v50, %0, 10\n\tv_add_u32 v51, %0, 11\n\t"
            "v_add_u32 v52, %0, 12\n\tv_add_u32 v53, %0, 13\n\tv_add_u32 v54, %0, 14\n\tv_add_u32 v55, %0, 15\n\t"
            "s_nop 7\n\t"
            "buffer_store_dwordx4 v[52:55], %1, %2, %3 offen\n\t"
            "buffer_store_dwordx4 v[48:51], %1, %2, %4 offen\n\t"
            "buffer_store_dwordx4 v[44:47], %1, %2, %5 offen\n\t"
            "buffer_store_dwordx4 v[40:43], %1, %2, %6 offen\n\t"
            ".if %7 > 0\n\ts_nop %7 - 1\n\t.endif\n\t"
            "v_mov_b32 v40, 0xdeadbeef\n\tv_mov_b32 v41, 0xdeadbeef\n\tv_mov_b32 v42, 0xdeadbeef\n\tv_mov_b32 v43, 0xdeadbeef\n\t"
            "s_nop 7"
            :: "v"(seed), "v"(vo), "s"(rs), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "n"(WAIT)
            : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    }
}

template <int WAIT>
static std::vector<float> run_mfma(int blocks, int iters) {
    const size_t n = (size_t)blocks * 256 * 4;
    float* d;
    (void)hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k_mfma<WAIT>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return h;
}
template <int WAIT>
static void run_store(const char* tag) {
    const int blocks = 1024, iters = 8, rows_per_wave = 4 * iters, rowstride = 1024 * 5;   // rows 5 KB apart
    const size_t waves = (size_t)blocks * 4, bytes = waves * rows_per_wave * rowstride + 1024;
    unsigned* d;
    (void)hipMalloc(&d, bytes);
    size_t bad = 0, n = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(d, 0, bytes);
        hipLaunchKernelGGL(k_store<WAIT>, dim3(blocks), dim3(256), 0, 0, d, rows_per_wave, rowstride, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> h(bytes / 4);
        (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost);
        for (size_t w = 0; w < waves; ++w)
            for (int it = 0; it < iters; ++it) {
                const size_t row0 = w * rows_per_wave + 4 * it;
                for (int lane = 0; lane < 64; ++lane) {
                    const unsigned seed = (unsigned)(row0 * 64 + lane) * 2654435761u;
                    for (int k = 0; k < 4; ++k)           // store k writes words seed + 12 - 4 k + {0..3} to row row0 + k
                        for (int i = 0; i < 4; ++i) {
                            const unsigned got = h[((row0 + k) * rowstride + lane * 16) / 4 + i];
                            bad += got != seed + 12 - 4 * k + i;
                            ++n;
                        }
                }
            }
    }
    printf("%s %zu %zu\n", tag, bad, n);
    (void)hipFree(d);
}

// ---- 3. a packed-fp32 instruction whose src1 select takes the high register for the low lane, an MFMA right behind ----
typedef _Float16 pk_h8 __attribute__((ext_vector_type(8)));
#define PK_BODY(PKINSTR, GAP, E0, E1)                                                                                   \
    asm volatile("ds_read_b64 v[100:101], %4\n\t"                                                                        \
                 "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v110, %7\n\tv_mov_b32 v111, %8\n\t"              \
                 "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t" PKINSTR "\n\t" GAP "v_mfma_f32_16x16x32_f16 v[114:117], %9, %10, 0\n\t" \
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" E0 "\n\t" E1 "\n\t"                                                 \
                 "v_mov_b32 %0, v106\n\tv_mov_b32 %1, v107"                                                              \
                 : "=&v"(d0), "=&v"(d1), "=&v"(e0), "=&v"(e1) : "v"(pair), "v"(c0), "v"(c1), "v"(a0), "v"(a1), "v"(a), "v"(b) \
                 : "v100", "v101", "v104", "v105", "v106", "v107", "v110", "v111", "v114", "v115", "v116", "v117", "memory")
template <int MODE>
__global__ __launch_bounds__(256) void k_pk(const float* in, unsigned* bad, int iters) {
    __shared__ __attribute__((aligned(16))) float pairs[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) pairs[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    pk_h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.01f * (lane + k)); b[k] = (_Float16)(0.02f * (lane - k)); }
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)pairs;
    unsigned nbad = 0;
    for (int i = 0; i < iters; ++i) {
        float d0, d1, e0, e1;
        const float a0 = in[(lane * 7 + i) & 1023], a1 = in[(lane * 13 + i + 5) & 1023], c0 = in[(lane + i) & 1023], c1 = in[(lane * 3 + i) & 1023];
        const unsigned pair = base + 8 * (i & 127);
        if (MODE == 0) PK_BODY("v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
        if (MODE == 1) PK_BODY("v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "s_nop 0\n\t", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
        if (MODE == 2) PK_BODY("v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel_hi:[1,0,1]", "", "v_fma_f32 %2, v110, v100, v104", "v_fma_f32 %3, v111, v100, v105");
        nbad += d0 != e0 || d1 != e1;
    }
    if (nbad) atomicAdd(bad, nbad);
}
template <int MODE>
static void run_pk(const char* tag) {
    std::vector<float> in(1024);
    for (int i = 0; i < 1024; ++i) in[i] = 0.001f * (float)((i * 7919) % 1999) - 1.f;
    float* din;
    unsigned* dbad;
    (void)hipMalloc(&din, 4096);
    (void)hipMalloc(&dbad, 4);
    (void)hipMemcpy(din, in.data(), 4096, hipMemcpyHostToDevice);
    (void)hipMemset(dbad, 0, 4);
    const int blocks = 256, iters = 1000;
    k_pk<MODE><<<blocks, 256>>>(din, dbad, iters);
    unsigned bad = 0;
    (void)hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    printf("%s %u %zu\n", tag, bad, (size_t)blocks * 256 * iters);
    (void)hipFree(din);
    (void)hipFree(dbad);
}

int main() {
    const int blocks = 1024, iters = 500;
    const auto ref = run_mfma<16>(blocks, iters);
    auto diff = [&](const std::vector<float>& a) {
        size_t bad = 0;
        for (size_t i = 0; i < a.size(); ++i) bad += memcmp(&a[i], &ref[i], 4) != 0;
        return bad;
    };
    printf("mfma0 %zu %zu\n", diff(run_mfma<0>(blocks, iters)), ref.size());
    printf("mfma2 %zu %zu\n", diff(run_mfma<2>(blocks, iters)), ref.size());
    printf("mfma16 %zu %zu\n", diff(run_mfma<16>(blocks, iters)), ref.size());
    run_store<0>("store0");
    run_store<2>("store2");
    run_pk<0>("pksel0");
    run_pk<1>("pksel1");
    run_pk<2>("pklow0");
    return 0;
}
