// Micro-benchmark: achievable HBM write / copy bandwidth (dev tool, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr(f4* out, size_t n16, f4 v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void cp(f4* out, const f4* in, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// 2 reads : 1 write
__global__ __launch_bounds__(256) void rrw(f4* out, const f4* in, const f4* in2, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i] + in2[i];
}
int main() {
    const size_t bytes = (size_t)1344 << 20, n16 = bytes / 16;
    f4 *a, *b, *c;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1024, 4096, 16384}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, a, n16, (f4){1, 2, 3, 4});
                else if (mode == 1) hipLaunchKernelGGL(cp, dim3(grid), dim3(256), 0, 0, a, b, n16);
                else hipLaunchKernelGGL(rrw, dim3(grid), dim3(256), 0, 0, a, b, c, n16);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double moved = bytes * (mode == 0 ? 1.0 : mode == 1 ? 2.0 : 3.0);
            printf("grid %5d %s: %.3f ms  %.2f TB/s total\n", grid, mode == 0 ? "write" : mode == 1 ? "copy " : "2r+1w", best, moved / best / 1e9);
        }
    }
    return 0;
}
