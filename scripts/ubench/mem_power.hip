// Micro-benchmark: what a byte costs in ENERGY at each level of the memory system.  A streaming read (and a streaming
// write) over a working set that lives in HBM (4 GB), in the 256 MB Infinity Cache (96 MB) or in the L2s (16 MB), sustained
// for seconds with the GPU's hwmon power / clock nodes sampled beside it; pJ per byte = (package power - power of the idle,
// clocked chip) / bytes per second.  The generation call runs at the package power cap: every joule that is
// not an MFMA is a candidate.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mem_power mem_power.hip     run: ./mem_power <hwmon dir> [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rd(const f4* __restrict__ p, size_t n, float* out) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const f4 v = __builtin_nontemporal_load(p + i);
        acc += v;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void rd_cached(const f4* __restrict__ p, size_t n, float* out) {
    f4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void wr(f4* __restrict__ p, size_t n, float v) {
    const f4 x = {v, v + 1, v + 2, v + 3};
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(x, p + i);
}
__global__ __launch_bounds__(256) void wr_cached(f4* __restrict__ p, size_t n, float v) {
    const f4 x = {v, v + 1, v + 2, v + 3};
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = x;
}
__global__ void idle_spin(float* out, int iters) {      // the chip clocked and busy with nothing: s_sleep loops
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    if (iters < 0) out[0] = 1.f;
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
    const double seconds = argc > 2 ? atof(argv[2]) : 2.5;
    const size_t big = (size_t)4 << 30;
    f4* buf;
    float* out;
    hipMalloc(&buf, big);
    hipMalloc(&out, 64);
    hipMemset(buf, 0x3c, big);
    double p_idle = 0;
    auto run = [&](const char* tag, int kind, size_t bytes) {
        const size_t n = bytes / 16;
        auto launch = [&]() {
            if (kind == 0) hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, buf, n, out);
            else if (kind == 1) hipLaunchKernelGGL(rd_cached, dim3(2048), dim3(256), 0, 0, buf, n, out);
            else if (kind == 2) hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, buf, n, 1.f);
            else hipLaunchKernelGGL(idle_spin, dim3(1024), dim3(256), 0, 0, out, 2000);
        };
        for (int w = 0; w < 20; ++w) launch();
        hipDeviceSynchronize();
        double psum = 0, fsum = 0;
        int ns = 0;
        long launches = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            const int per = bytes > ((size_t)1 << 30) ? 4 : 200;
            for (int w = 0; w < per; ++w) launch();
            launches += per;
            const long p = read_long(hw + "/power1_input"), f = read_long(hw + "/freq1_input");
            if (p > 0 && f > 0) { psum += p * 1e-6; fsum += f * 1e-6; ++ns; }
            hipDeviceSynchronize();
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double P = ns ? psum / ns : 0, F = ns ? fsum / ns : 0;
        if (kind == 3) { p_idle = P; printf("%-28s %7.1f W  %6.0f MHz\n", tag, P, F); return; }
        const double bps = (double)bytes * launches / dt;
        printf("%-28s %7.2f TB/s  %7.1f W  %6.0f MHz  -> %6.1f pJ/B above the busy-idle chip\n", tag, bps / 1e12, P, F, (P - p_idle) / bps * 1e12);
    };
    // producer -> consumer through the memory side: a window of W MB is written by one launch and read by the next, the window moving
    // through the 4 GB buffer so that nothing is left from an earlier pass.  `same`: the reader reads the window just written (could it
    // still be in the Infinity Cache?); `far`: it reads a window 2 GB away (certainly in HBM).  Same bytes both ways: the difference in
    // power and in the reader's time is what producing C right in front of its consumer could save.
    auto pc = [&](const char* tag, size_t wbytes, bool same, bool nt_w, bool nt_r) {
        const size_t n = wbytes / 16, nwin = big / wbytes;
        hipEvent_t e0, e1, e2;
        hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        double psum = 0, fsum = 0, wms = 0, rms = 0;
        int ns = 0;
        long pairs = 0;
        size_t w = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            for (int k = 0; k < 8; ++k, ++w) {
                f4* wp = buf + (w % nwin) * n;
                f4* rp = same ? wp : buf + ((w + nwin / 2) % nwin) * n;
                const bool ev = k == 7;
                if (ev) hipEventRecord(e0, 0);
                if (nt_w) hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, wp, n, 1.f);
                else hipLaunchKernelGGL(wr_cached, dim3(2048), dim3(256), 0, 0, wp, n, 1.f);
                if (ev) hipEventRecord(e1, 0);
                if (nt_r) hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, rp, n, out);
                else hipLaunchKernelGGL(rd_cached, dim3(2048), dim3(256), 0, 0, rp, n, out);
                if (ev) hipEventRecord(e2, 0);
            }
            const long p = read_long(hw + "/power1_input"), f = read_long(hw + "/freq1_input");
            if (p > 0 && f > 0) { psum += p * 1e-6; fsum += f * 1e-6; ++ns; }
            hipDeviceSynchronize();
            float a = 0, b = 0;
            hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
            wms += a; rms += b; ++pairs;
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double P = ns ? psum / ns : 0, F = ns ? fsum / ns : 0;
        const double bps = 2.0 * wbytes * (double)w / dt;
        printf("%-44s write %6.1f us  read %6.1f us  %6.2f TB/s  %7.1f W  %6.0f MHz  -> %6.1f pJ/B (both directions)\n", tag, wms / pairs * 1e3,
               rms / pairs * 1e3, bps / 1e12, P, F, (P - p_idle) / bps * 1e12);
    };
    run("busy-idle (s_sleep loops)", 3, 0);
    if (argc > 3) {      // ./mem_power <hwmon> <seconds> pc
        for (size_t mb : {(size_t)64, (size_t)105, (size_t)210}) {
            char t[96];
            for (int v = 0; v < 4; ++v) {
                const bool ntw = v & 1, ntr = v & 2;
                snprintf(t, sizeof t, "%3zu MB %s-write %s-read, same window", mb, ntw ? "nt" : "pl", ntr ? "nt" : "pl");
                pc(t, mb << 20, true, ntw, ntr);
                snprintf(t, sizeof t, "%3zu MB %s-write %s-read, far window", mb, ntw ? "nt" : "pl", ntr ? "nt" : "pl");
                pc(t, mb << 20, false, ntw, ntr);
            }
        }
        return 0;
    }
    run("read  4 GB   (HBM, nt)", 0, big);
    run("read  4 GB   (HBM)", 1, big);
    run("read  96 MB  (Infinity Cache)", 1, (size_t)96 << 20);
    run("read  16 MB  (L2)", 1, (size_t)16 << 20);
    run("write 4 GB   (HBM, nt)", 2, big);
    run("write 96 MB  (nt)", 2, (size_t)96 << 20);
    return 0;
}
