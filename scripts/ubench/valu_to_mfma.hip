// Micro-test (dev tool, not product): does an MFMA that reads a VGPR as srcB in the slot right behind the VALU
// instruction that wrote it see the new value on gfx950?  hipcc pads nothing around instructions inside an asm statement
// (the split-fp16 codec's v_fma_mix* are such), so the question is whether the hardware interlocks this pair.
//   mode 0: v_fma_mixlo/hi_f16 -> v_mfma (0 slots between)     mode 1: one v_mov between     mode 2: s_nop 7 between (reference)
// Every lane derives its operand from (lane, iteration); the accumulated MFMA results of modes 0 / 1 must equal mode 2's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + 0.001f * lane;
    unsigned scratch = 0;
    for (int it = 0; it < iters; ++it) {
        float x0 = 0.37f * (lane + 1) + 1e-3f * it, x1 = -0.11f * (lane + 3) + 2e-3f * it;
        unsigned hi, l;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
        if (MODE == 0)
            asm volatile("v_fma_mixlo_f16 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_mfma_f32_4x4x1_16b_f32 %1, %5, %0, %1"
                         : "=&v"(l), "+v"(acc) : "v"(hi), "v"(x0), "v"(x1), "v"(a));
        else if (MODE == 200)
            asm volatile("v_fma_mixlo_f16 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_mov_b32 %6, %5\n\t"
                         "v_mfma_f32_4x4x1_16b_f32 %1, %5, %0, %1"
                         : "=&v"(l), "+v"(acc), "+v"(scratch) : "v"(hi), "v"(x0), "v"(x1), "v"(a));
        else if (MODE == 300)      // plain full-register VALU producer instead of the op_sel one
            asm volatile("v_add_f32 %0, %3, %4\n\t"
                         "v_mfma_f32_4x4x1_16b_f32 %1, %5, %0, %1"
                         : "=&v"(l), "+v"(acc) : "v"(hi), "v"(x0), "v"(x1), "v"(a));
        else if (MODE == 301)
            asm volatile("v_add_f32 %0, %3, %4\n\ts_nop 7\n\ts_nop 7\n\t"
                         "v_mfma_f32_4x4x1_16b_f32 %1, %5, %0, %1"
                         : "=&v"(l), "+v"(acc) : "v"(hi), "v"(x0), "v"(x1), "v"(a));
        else
            asm volatile("v_fma_mixlo_f16 %0, %2, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %0, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                         "s_nop %6\n\t"
                         "v_mfma_f32_4x4x1_16b_f32 %1, %5, %0, %1"
                         : "=&v"(l), "+v"(acc) : "v"(hi), "v"(x0), "v"(x1), "v"(a), "n"(MODE >= 100 ? 15 : MODE - 1));
        // keep the accumulator finite: fold it every iteration (the packed halves read as f32 can be anything)
        asm volatile("s_nop 7\n\ts_nop 7");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned u = __float_as_uint(acc[i]);
            u = (u & 0x007fffffu) | 0x3f800000u;          // mantissa bits only, value in [1, 2)
            acc[i] = __uint_as_float(u) - 1.0f;
        }
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 0] = acc[0] + scratch * 0.f;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 1] = acc[1];
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 2] = acc[2];
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 3] = acc[3];
}

// the codec's two halves: v_fma_mixlo_f16 writes the low 16 bits of a register, v_fma_mixhi_f16 the high 16 bits of the
// same register right behind it -- does the second one keep what the first one wrote when they issue back to back?
template <int PAD>
__global__ __launch_bounds__(256) void k2(unsigned* out, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned sum = 0;
    for (int it = 0; it < iters; ++it) {
        float x0 = 0.37f * (lane + 1) + 1e-3f * it, x1 = -0.11f * (lane + 3) + 2e-3f * it;
        unsigned hi, l;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\ts_nop 3" : "=v"(hi) : "v"(x0), "v"(x1));
        if (PAD)
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\ts_nop 3\n\t"
                         "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 3"
                         : "=&v"(l) : "v"(hi), "v"(x0), "v"(x1));
        else
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                         "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 3"
                         : "=&v"(l) : "v"(hi), "v"(x0), "v"(x1));
        sum = sum * 1664525u + l;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <int PAD>
static std::vector<unsigned> run2(int blocks, int iters) {
    const size_t n = (size_t)blocks * 256;
    unsigned* d;
    (void)hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k2<PAD>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return h;
}

template <int MODE>
static std::vector<float> run(int blocks, int iters) {
    const size_t n = (size_t)blocks * 256 * 4;
    float* d;
    (void)hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return h;
}
static size_t diff(const std::vector<float>& a, const std::vector<float>& b) {
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += memcmp(&a[i], &b[i], 4) != 0;
    return bad;
}
int main() {
    const int blocks = 2048, iters = 2000;
    const auto ref = run<100>(blocks, iters), ref2 = run<100>(blocks, iters);
    printf("values %zu; reference (16 wait states) twice: %zu differ\n", ref.size(), diff(ref, ref2));
    printf("op_sel producer, 0 wait states : %zu differ\n", diff(run<0>(blocks, iters), ref));
    printf("op_sel producer, one v_mov     : %zu differ\n", diff(run<200>(blocks, iters), ref));
    printf("op_sel producer, 1 wait state  : %zu differ\n", diff(run<1>(blocks, iters), ref));
    printf("op_sel producer, 2 wait states : %zu differ\n", diff(run<2>(blocks, iters), ref));
    printf("op_sel producer, 3 wait states : %zu differ\n", diff(run<3>(blocks, iters), ref));
    printf("op_sel producer, 4 wait states : %zu differ\n", diff(run<4>(blocks, iters), ref));
    printf("op_sel producer, 6 wait states : %zu differ\n", diff(run<6>(blocks, iters), ref));
    printf("op_sel producer, 8 wait states : %zu differ\n", diff(run<8>(blocks, iters), ref));
    printf("v_add_f32 producer, 0 wait states vs 16: %zu differ\n", diff(run<300>(blocks, iters), run<301>(blocks, iters)));
    {
        const auto a = run2<0>(blocks, iters), b = run2<1>(blocks, iters);
        size_t bad = 0;
        for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
        printf("mixlo -> mixhi of one register back to back vs 4 wait states apart: %zu of %zu lane checksums differ\n", bad, a.size());
    }
    return 0;
}
