// Micro-benchmark: achievable streaming-read rate for the enc-like pattern (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
// pattern A: fully linear; pattern B: [64 rows][T][4 words] rows read at the same 1 KB column chunk (G4 tile order)
template <int UNR>
__global__ __launch_bounds__(256) void rd_linear(const u4* p, size_t n16, unsigned* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    u4 acc = {0, 0, 0, 0};
    for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
        u4 v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNR; ++k) acc ^= v[k];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345) out[0] = 1;
}
// each WG walks 64-column tiles; per tile each wave loads 64 group rows x 16 columns... (4 waves cover 64 columns)
__global__ __launch_bounds__(256) void rd_g4(const u4* p, int rows, int T, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    u4 acc = {0, 0, 0, 0};
    const int ntiles = T / 64;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t col = (size_t)tile * 64 + wave * 16 + n;
        u4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[((size_t)(4 * k + q)) * T + col];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc ^= v[k];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345) out[0] = 1;
}
int main() {
    const int rows = 64, T = 76800;            // 64 group rows x T x 16 B = 78.6 MB (enc)
    const size_t bytes = (size_t)rows * T * 16;
    u4* p; unsigned* out;
    (void)hipMalloc(&p, bytes); (void)hipMalloc(&out, 4);
    (void)hipMemset(p, 1, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](auto f, const char* name) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) f();
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%-28s %.2f us per pass  %.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
        }
    };
    time([&] { hipLaunchKernelGGL(rd_linear<8>, dim3(256), dim3(256), 0, 0, p, bytes / 16, out); }, "linear 256 WG unroll 8");
    time([&] { hipLaunchKernelGGL(rd_linear<8>, dim3(1024), dim3(256), 0, 0, p, bytes / 16, out); }, "linear 1024 WG unroll 8");
    time([&] { hipLaunchKernelGGL(rd_linear<16>, dim3(2048), dim3(256), 0, 0, p, bytes / 16, out); }, "linear 2048 WG unroll 16");
    time([&] { hipLaunchKernelGGL(rd_g4, dim3(256), dim3(256), 0, 0, p, rows, T, out); }, "g4 tiles 256 WG");
    time([&] { hipLaunchKernelGGL(rd_g4, dim3(512), dim3(256), 0, 0, p, rows, T, out); }, "g4 tiles 512 WG");
    time([&] { hipLaunchKernelGGL(rd_g4, dim3(1200), dim3(256), 0, 0, p, rows, T, out); }, "g4 tiles 1200 WG");
    return 0;
}
