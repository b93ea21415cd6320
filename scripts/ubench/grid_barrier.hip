// Micro-benchmark: cost of a device-wide barrier inside one cooperative kernel on MI355X
// (dev tool, not product).  Spins are bounded so that a bug cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// VARIANT 0: everybody polls the counter.  1: the last arriver publishes a flag on another line,
// the rest poll the flag.  2: as 1 without the fences (cost of the cache maintenance alone).
#ifndef VARIANT
#define VARIANT 1
#endif
__device__ inline bool grid_barrier(unsigned* counter, unsigned& epoch, unsigned nwg, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
#if VARIANT != 2
        __threadfence();                                  // release: write back this XCD's L2
#endif
        ++epoch;
        const unsigned target = epoch * nwg;
        unsigned* flag = counter + 64;
        long spins = 0;
#if VARIANT == 0
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000) { *err = 1; ok = false; break; }
        }
#else
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == target) {
            __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 20000000) { *err = 1; ok = false; break; }
            }
        }
#endif
#if VARIANT != 2
        __threadfence();                                  // acquire: drop stale lines
#endif
    }
    __syncthreads();
    return ok;
}

// mode 0: barrier only.  mode 1: every workgroup writes `words` words, barrier, reads the words of
// workgroup (b + gridDim/2 + 3) % gridDim (another XCD) and checks them.
__global__ __launch_bounds__(256) void k(unsigned* counter, int* err, unsigned* buf, int words, int iters, int mode,
                                         unsigned* bad) {
    unsigned epoch = 0;
    const unsigned nwg = gridDim.x;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        if (mode == 1) {
            unsigned* mine = buf + (size_t)blockIdx.x * words;
            for (int i = threadIdx.x; i < words; i += 256) mine[i] = (unsigned)(it * 7919 + blockIdx.x * 131 + i);
        }
        if (!grid_barrier(counter, epoch, nwg, err)) return;
        if (mode == 1) {
            const unsigned o = (blockIdx.x + nwg / 2 + 3) % nwg;
            const unsigned* other = buf + (size_t)o * words;
            for (int i = threadIdx.x; i < words; i += 256)
                if (other[i] != (unsigned)(it * 7919 + o * 131 + i)) ++nbad;
            if (!grid_barrier(counter, epoch, nwg, err)) return;   // nobody overwrites before all have read
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned *counter, *buf, *bad;
    int* err;
    hipMalloc(&counter, 1024); hipMalloc(&err, 4); hipMalloc(&bad, 4);
    const int words = 16384;                              // 64 KB per workgroup
    hipMalloc(&buf, (size_t)1024 * words * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512}) {
        for (int mode = 0; mode < 2; ++mode) {
            hipMemset(counter, 0, 1024); hipMemset(err, 0, 4); hipMemset(bad, 0, 4);
            int iters = mode == 0 ? 2000 : 500, w = words;
            void* args[] = {&counter, &err, &buf, &w, &iters, &mode, &bad};
            hipEventRecord(e0);
            hipError_t rc = hipLaunchCooperativeKernel((const void*)k, dim3(grid), dim3(256), args, 0, 0);
            hipEventRecord(e1);
            hipError_t rs = hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            int herr = 0; unsigned hbad = 0;
            hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
            const int nbar = mode == 0 ? iters : 2 * iters;
            printf("grid %d mode %d: launch %s sync %s  %.3f ms  %.2f us per barrier%s  spin-timeout %d  mismatches %u\n", grid, mode,
                   hipGetErrorString(rc), hipGetErrorString(rs), ms, ms * 1e3 / nbar,
                   mode ? " (incl. 64 KB write + read per workgroup per pair)" : "", herr, hbad);
        }
    }
    return 0;
}
