// Micro-benchmark: producer -> consumer tile hand-off between persistent workgroups through L2
// (same XCD) or through memory (different XCDs), with per-access cache policy instead of fences
// (dev tool, not product).  Spins are bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ inline u4 ld4(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, AUX));
}
template <int AUX>
__device__ inline void st4(u4 v, __amdgpu_buffer_rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, AUX);
}
template <int AUX>
__device__ inline unsigned ld1(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, AUX);
}
template <int AUX>
__device__ inline void st1(unsigned v, __amdgpu_buffer_rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, 0, AUX);
}

constexpr int RING = 8, TILE_BYTES = 16384;

// LD/ST: aux bits of the data accesses; FL/FS: of the flag accesses (1 = sc0, 16 = sc1, 17 = both)
template <int LD, int ST, int FL, int FS>
__global__ __launch_bounds__(256) void chain(unsigned* data, unsigned* flags, int nstage, int ntiles, int cross,
                                             unsigned* bad, int* err, int work) {
    // same-XCD chain: stages are the workgroups with blockIdx % 8 == 0; cross: stage = blockIdx
    int k;
    if (cross) { k = blockIdx.x; } else { if (blockIdx.x & 7) return; k = blockIdx.x >> 3; }
    if (k >= nstage) return;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)data, 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc((void*)flags, 0, 4096 * 64, 0x00020000);
    unsigned nbad = 0;
    float sink = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        if (threadIdx.x == 0) {
            long spins = 0;
            if (k > 0)
                while (ld1<FL>(rf, (k - 1) * 256) <= (unsigned)t) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000) { *err = 1; break; }
                }
            if (k + 1 < nstage && t >= RING)
                while (ld1<FL>(rf, (k + 1) * 256) <= (unsigned)(t - RING)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 4000000) { *err = 2; break; }
                }
        }
        __syncthreads();
        if (*err) return;
        const int slot = t % RING;
        u4 acc = (u4){0, 0, 0, 0};
        if (k > 0) {
            const int base = ((k - 1) * RING + slot) * TILE_BYTES + threadIdx.x * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u4 v = ld4<LD>(rd, base + i * 4096);
                const unsigned want = (unsigned)((k - 1) * 1000003 + t * 17 + i);
                if (v[0] != want || v[3] != want + threadIdx.x) ++nbad;
                acc += v;
            }
        }
        // stand-in for the tile's arithmetic
        float f = (float)acc[1];
        for (int w = 0; w < work; ++w) f = __builtin_fmaf(f, 1.0001f, 0.5f);
        sink += f;
        const int ob = (k * RING + slot) * TILE_BYTES + threadIdx.x * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned tag = (unsigned)(k * 1000003 + t * 17 + i);
            st4<ST>((u4){tag, acc[1], acc[2], tag + threadIdx.x}, rd, ob + i * 4096);
        }
        __builtin_amdgcn_s_waitcnt(0);          // stores acknowledged (vmcnt = 0)
        __syncthreads();
        if (threadIdx.x == 0) st1<FS>((unsigned)(t + 1), rf, k * 256);
    }
    if (nbad) atomicAdd(bad, nbad);
    if (sink == 123.456f) *err = 9;
}

template <int LD, int ST, int FL, int FS>
void run(const char* name, int nstage, int cross, int work) {
    unsigned *data, *flags, *bad; int* err;
    hipMalloc(&data, (size_t)256 * RING * TILE_BYTES); hipMalloc(&flags, 4096 * 64); hipMalloc(&bad, 4); hipMalloc(&err, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float t1 = 0, t2 = 0;
    unsigned hbad = 0; int herr = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int ntiles = pass == 0 ? 200 : 2200;
        hipMemset(data, 0xff, (size_t)256 * RING * TILE_BYTES); hipMemset(flags, 0, 4096 * 64); hipMemset(bad, 0, 4); hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        void* args[] = {&data, &flags, &nstage, &ntiles, &cross, &bad, &err, &work};
        hipEventRecord(e0);
        hipLaunchCooperativeKernel((const void*)chain<LD, ST, FL, FS>, dim3(256), dim3(256), args, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        (pass == 0 ? t1 : t2) = ms;
        unsigned b; int e; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        hbad += b; herr |= e;
    }
    const double per_tile = (t2 - t1) * 1e3 / 2000.0;            // steady state
    const double fill = (t1 * 1e3 - 200 * per_tile) / nstage;      // per-hop latency
    printf("%-34s stages %3d %s work %5d: %.2f us/tile steady, %.2f us/hop fill, mismatches %u err %d\n", name, nstage,
           cross ? "cross-XCD" : "same-XCD ", work, per_tile, fill, hbad, herr);
    hipFree(data); hipFree(flags); hipFree(bad); hipFree(err);
}

int main() {
    for (int work : {0, 2000}) {
        run<1, 0, 1, 0>("data ld sc0 / st default, flag sc0", 32, 0, work);
        run<0, 0, 1, 0>("data ld default (L1 may be stale)", 32, 0, work);
        run<16, 16, 16, 16>("all sc1", 32, 0, work);
        run<16, 16, 16, 16>("all sc1", 32, 1, work);
        run<17, 17, 17, 17>("all sc0|sc1", 32, 1, work);
        run<1, 0, 1, 0>("sc0 only across XCDs (expect bad)", 32, 1, work);
    }
    return 0;
}
