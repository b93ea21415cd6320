// Mel-conditioning upsampler: the transposed-conv stack
//   wavenet/wavenet.py:46-73 (_deconv_stack), wavenet/masked.py:235-291 (trans_conv1d)
// as fp32 MFMA GEMMs on gfx950.
//
// conv2d_transpose(SAME, stride S, filter [1,K,Cout,Cin]) in closed form:
//   y[n,co] = bias[co] + sum_i sum_ci x[i,ci] * W[n + pL - i*S, co, ci],  pL = (K-S)/2.
// With n + pL = S*q + r this is, per phase r, a dense GEMM over k = (j, ci):
//   y[S*q + r - pL, co] = bias[co] + sum_{j<K/S} sum_ci W[S*j + r, co, ci] * x[q - j, ci]
// so  A = W_r [Cout x (K/S*Cin)]  (weights, pre-packed in MFMA A-fragment order)
//     B = X   [(j,ci) x q]        (activations, channel-major rows, time contiguous)
//     D = Y_r [Cout x q]
// Activations are kept CHANNEL-MAJOR ([C][time]) everywhere on the device so that
// the B operand of v_mfma_f32_16x16x4_f32 (lane = (k>>?, n)) is a coalesced
// time-contiguous load and dilation / tap shifts are plain column offsets.
#include <algorithm>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_pack_h.h"
#include "wn_mfma_h.h"


namespace {

constexpr int DC_NT = 4;             // 16-column MFMA tiles per wave (dwordx4 loads)
constexpr int DC_QT = 16 * DC_NT;    // q columns per wave
constexpr int DC_XOFF = 8;           // zero columns left of sample 0 in every input row
constexpr int DC_QW = 256;           // q columns per workgroup of the split-fp16 GEMM (4 waves x DC_QT)
constexpr int DH_KC = 4;             // K-steps of weights staged in LDS per pipeline stage

__host__ __device__ inline int dc_row_stride(int L) {
    // columns: [xoff zeros][L samples][zeros up to the rounded q range + slack]
    return DC_XOFF + ((L + 5 + DC_QW - 1) / DC_QW) * DC_QW + 8;
}

// mel [B,F,C] (reference layout) -> channel-major padded rows [B][C][xs]
__global__ void mel_to_cm_kernel(const float* __restrict__ mel, float* __restrict__ out,
                                 int F, int C, int xs) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= xs) return;
    const int f = col - DC_XOFF;
    float v = 0.f;
    if (f >= 0 && f < F) v = mel[((size_t)b * F + f) * C + c];
    out[((size_t)b * C + c) * xs + col] = v;
}

__device__ inline float apply_act(float v, int act) {
    if (act == WN_ACT_LEAKY_RELU) return fmaxf(v, 0.4f * v);   // masked.py:33-34
    if (act == WN_ACT_RELU) return fmaxf(v, 0.f);
    return tanhf(v);
}

// One workgroup = 4 waves; wave w owns output channels [64*(4*zc + w), +64) as 4
// MFMA row blocks; all waves share the same 64 q columns (their B loads hit L1).
// The GEMM result of phase r is written PHASE-MAJOR, yp[b][r][co][q] (16 contiguous bytes
// per lane, full lines per row); deconv_interleave_kernel then weaves the S phases into
// time order.  Writing y[co][S*q + r - pL] directly from here would be a 4-byte scatter at
// stride S*4 bytes -- measured 8.8x HBM write amplification (rocprofv3 WRITE_SIZE).
__global__ __launch_bounds__(256) void deconv_mfma_kernel(
    const float* __restrict__ x, int cin, int xs, const float* __restrict__ wp,
    float* __restrict__ yp, int cout, int Qp, int S, int taps, int zc_count) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int r = blockIdx.y;
    const int b = blockIdx.z / zc_count, zc = blockIdx.z % zc_count;
    const int cg = zc * 4 + wave;                 // 64-channel group
    if (cg * 64 >= cout) return;
    const int q0 = blockIdx.x * DC_QT;
    const int nmb = cout / 16;
    const int cblk = cin / 16;                    // K-step groups (of 4 K-steps) per tap
    const int nks4 = taps * cblk;

    f4 acc[4][DC_NT];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int e = 0; e < DC_NT; ++e) acc[mb][e] = (f4){0.f, 0.f, 0.f, 0.f};

    const f4* wpr = reinterpret_cast<const f4*>(wp) + ((size_t)r * nks4 * nmb + cg * 4) * 64 + lane;
    const float* xb = x + (size_t)b * cin * xs + DC_XOFF + q0 + DC_NT * n;

    // operands of K-group `it` (tap j = it / cblk, channel block c4 = it % cblk), loaded one
    // group ahead of the 64 MFMAs that consume them
    auto load = [&](int it, f4 (&a)[4], f4 (&bv)[4]) {
        const int j = it / cblk, c4 = it - j * cblk;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[mb] = wpr[((size_t)it * nmb + mb) * 64];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            bv[jj] = *reinterpret_cast<const f4u*>(xb + (size_t)(16 * c4 + 4 * jj + kq) * xs - j);
    };
    f4 a1[4], b1[4];
    load(0, a1, b1);
    for (int it = 0; it < nks4; ++it) {
        f4 a[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = a1[i]; bv[i] = b1[i]; }
        if (it + 1 < nks4) load(it + 1, a1, b1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int e = 0; e < DC_NT; ++e)
                    acc[mb][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][jj], bv[jj][e], acc[mb][e], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int co = 64 * cg + 16 * mb + 4 * kq + rr;
            float* row = yp + (((size_t)b * S + r) * cout + co) * Qp + q0 + DC_NT * n;
            *reinterpret_cast<f4*>(row) = (f4){acc[mb][0][rr], acc[mb][1][rr], acc[mb][2][rr], acc[mb][3][rr]};
        }
}

// yp[b][r][co][q] -> y[b][co][yoff + S*q + r - pL] with bias + activation.  One workgroup =
// one channel x 64 q columns = S*64 consecutive output samples, transposed through LDS so
// that both the reads (64 contiguous q per phase row) and the writes are full lines.
constexpr int DI_MAXS = 32;
__global__ __launch_bounds__(256) void deconv_interleave_kernel(
    const float* __restrict__ yp, const float* __restrict__ bias, float* __restrict__ y,
    int cout, int Qp, int64_t ys, int yoff, int L, int S, int pL, int act) {
    __shared__ float tile[DI_MAXS][DC_QT + 1];
    const int co = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * DC_QT;
    for (int i = threadIdx.x; i < S * DC_QT; i += 256) {
        const int r = i / DC_QT, q = i - r * DC_QT;
        tile[r][q] = yp[(((size_t)b * S + r) * cout + co) * Qp + q0 + q];
    }
    __syncthreads();
    const float bco = bias[co];
    const int64_t SL = (int64_t)S * L;
    float* yr = y + ((size_t)b * cout + co) * ys + yoff;
    const int64_t n0 = (int64_t)S * q0 - pL;
    for (int i = threadIdx.x; i < S * DC_QT; i += 256) {
        const int q = i / S, r = i - q * S;
        const int64_t nn = n0 + i;
        if (nn >= 0 && nn < SL) yr[nn] = apply_act(tile[r][q] + bco, act);
    }
}

// mel [B,F,C] -> split-fp16 pair planes [B][2][C/2][xs] (zero padded)
__global__ void mel_to_split_kernel(const float* __restrict__ mel, unsigned* __restrict__ out,
                                    int F, int C, int xs, unsigned* __restrict__ status) {
    const int b = blockIdx.z, cp = blockIdx.y;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= xs) return;
    const int f = col - DC_XOFF;
    unsigned hw = 0, lw = 0;
    if (f >= 0 && f < F) {
        const float* m = mel + ((size_t)b * F + f) * C + 2 * cp;
        float amax = 0.f;
        wn_split_pair_t(m[0], m[1], hw, lw, amax);
        wn_range_flag(amax, status);
    }
    out[((size_t)b * C + cp) * xs + col] = hw;
    out[((size_t)b * C + C / 2 + cp) * xs + col] = lw;
}

// Split-fp16 version of deconv_mfma_kernel (v_mfma_f32_16x16x32_f16, 3 MFMAs per product).
// One workgroup = one phase r x 64 output channels x 256 q columns; wave w owns 64 of the
// columns (4 MFMA column tiles, dwordx4 operand loads).  All four waves use the SAME weight
// fragments, so those are staged through LDS in chunks of DH_KC K-steps (double buffered):
// without the sharing the fp16 MFMA rate would ask the L2 for > 12 TB/s of weight reads.
// The flattened K axis is (tap j, 16-channel block); one K-step = two consecutive blocks,
// k-slot (kg, e) <-> channel 16*blk(e>>2) + 4*kg + (e&3) of tap j(e>>2).
// G4IN: the input is in the 4-row-interleaved G4 layout (needs cin % 32 == 0, so that both
// 16-channel blocks of a K-step belong to one tap): one 16-byte load per column is the whole
// operand, no register transposes.  Otherwise planar pair planes (mel, cin = 80).
template <bool G4IN>
__global__ __launch_bounds__(256, 2) void deconv_mfma_h_kernel(
    const unsigned* __restrict__ x, int cin, int xs, const unsigned* __restrict__ wp,
    float* __restrict__ yp, int cout, int Qp, int S, int taps, float inv_scale) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2][DH_KC * 4 * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4;
    const int r = blockIdx.y;
    const int ncg = cout / 64;
    const int b = blockIdx.z / ncg, cg = blockIdx.z % ncg;
    const int q0 = blockIdx.x * DC_QW + wave * DC_QT;
    const int nmb = cout / 16;
    const int nb16 = cin / 16;                     // 16-channel blocks per tap
    const int nks = taps * nb16 / 2;               // K-steps of 32
    const int nchunk = (nks + DH_KC - 1) / DH_KC;

    f4 acc[4][DC_NT];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int e = 0; e < DC_NT; ++e) acc[mb][e] = (f4){0.f, 0.f, 0.f, 0.f};

    const wn_u4* wsrc = reinterpret_cast<const wn_u4*>(wp);       // [r][ks][mb][plane][lane] u4
    auto stage = [&](int chunk, int buf) {
        // DH_KC K-steps x (4 row blocks x 2 planes x 64 lanes) u4 = DH_KC x 512 u4; 2 per thread per K-step
#pragma unroll
        for (int kl = 0; kl < DH_KC; ++kl) {
            const int ks = chunk * DH_KC + kl;
            if (ks < nks) {
                const wn_u4* src = wsrc + (((size_t)r * nks + ks) * nmb + cg * 4) * 128;
                wn_u4* dst = reinterpret_cast<wn_u4*>(lds[buf]) + kl * 512;
                dst[threadIdx.x] = src[threadIdx.x];
                dst[threadIdx.x + 256] = src[threadIdx.x + 256];
            }
        }
    };
    const unsigned* xb = x + (size_t)b * cin * xs;
    const size_t lo_plane = (size_t)(cin / 2) * xs;
    // operand words of K-step ks for the DC_NT columns of this lane: vh[e], vl[e]
    auto loadB = [&](int ks, wn_u4 (&vh)[DC_NT], wn_u4 (&vl)[DC_NT]) {
        if (G4IN) {
            const int nb32 = cin / 32;
            const int j = ks / nb32, c = ks - j * nb32;
            const unsigned* p = xb + ((size_t)(4 * c + kg) * xs + DC_XOFF + q0 + DC_NT * n - j) * 4;
#pragma unroll
            for (int e = 0; e < DC_NT; ++e) {
                vh[e] = *reinterpret_cast<const wn_u4*>(p + 4 * e);
                vl[e] = *reinterpret_cast<const wn_u4*>(p + lo_plane + 4 * e);
            }
        } else {
            wn_u4 bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = 2 * ks + (i >> 1);               // flattened 16-channel block
                const int j = g / nb16, blk = g - j * nb16;
                const unsigned* p = xb + (size_t)(8 * blk + 2 * kg + (i & 1)) * xs + DC_XOFF + q0 + DC_NT * n - j;
                bh[i] = *reinterpret_cast<const wn_u4 __attribute__((aligned(4)))*>(p);
                bl[i] = *reinterpret_cast<const wn_u4 __attribute__((aligned(4)))*>(p + lo_plane);
            }
#pragma unroll
            for (int e = 0; e < DC_NT; ++e) {
                vh[e] = (wn_u4){bh[0][e], bh[1][e], bh[2][e], bh[3][e]};
                vl[e] = (wn_u4){bl[0][e], bl[1][e], bl[2][e], bl[3][e]};
            }
        }
    };

    stage(0, 0);
    wn_u4 b1h[DC_NT], b1l[DC_NT];      // operands one K-step ahead (two waves per SIMD hide the rest)
    loadB(0, b1h, b1l);
    __syncthreads();
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunk) stage(chunk + 1, buf ^ 1);
        const wn_u4* Al = reinterpret_cast<const wn_u4*>(lds[buf]) + lane;
#pragma unroll
        for (int kl = 0; kl < DH_KC; ++kl) {
            const int ks = chunk * DH_KC + kl;
            if (ks >= nks) break;
            wn_u4 vh[DC_NT], vl[DC_NT];
#pragma unroll
            for (int e = 0; e < DC_NT; ++e) { vh[e] = b1h[e]; vl[e] = b1l[e]; }
            if (ks + 1 < nks) loadB(ks + 1, b1h, b1l);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const wn_u4 ah = Al[(kl * 4 + mb) * 128], al = Al[(kl * 4 + mb) * 128 + 64];
#pragma unroll
                for (int e = 0; e < DC_NT; ++e) {
                    f4 c = acc[mb][e];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, ah), __builtin_bit_cast(wn_h8, vh[e]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, ah), __builtin_bit_cast(wn_h8, vl[e]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, al), __builtin_bit_cast(wn_h8, vh[e]), c, 0, 0, 0);
                    acc[mb][e] = c;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int co = 64 * cg + 16 * mb + 4 * kg + rr;
            float* row = yp + (((size_t)b * S + r) * cout + co) * Qp + q0 + DC_NT * n;
            *reinterpret_cast<f4*>(row) = (f4){acc[mb][0][rr] * inv_scale, acc[mb][1][rr] * inv_scale,
                                               acc[mb][2][rr] * inv_scale, acc[mb][3][rr] * inv_scale};
        }
}

// G4-input variant that loads every activation column ONCE per 32-channel block: tap j of the
// transposed conv reads the same rows one column to the left of tap j-1, and a lane owns DC_NT
// consecutive columns, so the shifted operand is a register renaming (e -> e+1) plus one
// row_shr:1 DPP move per word for e = 0, with the column left of the wave's range ("halo")
// patched into lane 0 of each 16-lane row.  K is walked block-major (all taps of a block, then
// the next block); the A fragments of those K-steps are staged together.  4x less operand
// traffic than reloading per tap (the plain kernel is bound by it, not by the MFMA pipe).
__device__ inline unsigned dpp_shr1(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x111, 0xf, 0xf, false);   // row_shr:1
}

template <int TAPS>
__global__ __launch_bounds__(256, 2) void deconv_mfma_hs_kernel(
    const unsigned* __restrict__ x, int cin, int xs, const unsigned* __restrict__ wp,
    float* __restrict__ yp, int cout, int Qp, int S, float inv_scale) {
    constexpr int taps = TAPS, NH = TAPS > 1 ? TAPS - 1 : 1;
    __shared__ __attribute__((aligned(16))) unsigned lds[2][DH_KC * 4 * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kg = lane >> 4;
    const int r = blockIdx.y;
    const int ncg = cout / 64;
    const int b = blockIdx.z / ncg, cg = blockIdx.z % ncg;
    const int q0 = blockIdx.x * DC_QW + wave * DC_QT;
    const int nmb = cout / 16;
    const int nb32 = cin / 32;                     // 32-channel blocks
    const int nks = taps * nb32;                   // K-step ks = tap * nb32 + block (pack order)
    constexpr int ntg = (taps + DH_KC - 1) / DH_KC;   // tap groups per block (one LDS stage each)
    const int nchunk = nb32 * ntg;

    f4 acc[4][DC_NT];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int e = 0; e < DC_NT; ++e) acc[mb][e] = (f4){0.f, 0.f, 0.f, 0.f};

    const wn_u4* wsrc = reinterpret_cast<const wn_u4*>(wp);       // [r][ks][mb][plane][lane] u4
    // weight fragments of a chunk: global -> registers at the start of the previous chunk, registers ->
    // LDS at its end, so the L2 latency is covered by that chunk's MFMAs
    wn_u4 wt[DH_KC][2];
    auto stage_load = [&](int chunk) {
        const int c = chunk / ntg, tg = chunk - c * ntg;
#pragma unroll
        for (int kl = 0; kl < DH_KC; ++kl) {
            const int j = tg * DH_KC + kl;
            if (j < taps) {
                const wn_u4* src = wsrc + (((size_t)r * nks + j * nb32 + c) * nmb + cg * 4) * 128;
                wt[kl][0] = src[threadIdx.x];
                wt[kl][1] = src[threadIdx.x + 256];
            }
        }
    };
    auto stage_store = [&](int chunk, int buf) {
        const int c = chunk / ntg, tg = chunk - c * ntg;
#pragma unroll
        for (int kl = 0; kl < DH_KC; ++kl) {
            if (tg * DH_KC + kl < taps) {
                wn_u4* dst = reinterpret_cast<wn_u4*>(lds[buf]) + kl * 512;
                dst[threadIdx.x] = wt[kl][0];
                dst[threadIdx.x + 256] = wt[kl][1];
            }
        }
    };
    const unsigned* xb = x + (size_t)b * cin * xs;
    const size_t lo_plane = (size_t)(cin / 2) * xs;
    // tap-0 operands of block c (DC_NT consecutive columns per lane) and the halo columns q0-1 .. q0-7
    auto loadB = [&](int c, wn_u4 (&vh)[DC_NT], wn_u4 (&vl)[DC_NT], wn_u4 (&hh)[NH], wn_u4 (&hl)[NH]) {
        const unsigned* row = xb + ((size_t)(4 * c + kg) * xs + DC_XOFF + q0) * 4;
        const unsigned* p = row + DC_NT * n * 4;
#pragma unroll
        for (int e = 0; e < DC_NT; ++e) {
            vh[e] = *reinterpret_cast<const wn_u4*>(p + 4 * e);
            vl[e] = *reinterpret_cast<const wn_u4*>(p + lo_plane + 4 * e);
        }
#pragma unroll
        for (int k = 0; k + 1 < taps; ++k) {
            hh[k] = *reinterpret_cast<const wn_u4*>(row - 4 * (k + 1));
            hl[k] = *reinterpret_cast<const wn_u4*>(row + lo_plane - 4 * (k + 1));
        }
    };

    stage_load(0);
    stage_store(0, 0);
    wn_u4 nh[DC_NT], nl[DC_NT], nhh[NH], nhl[NH];    // next block's operands, one block ahead
    loadB(0, nh, nl, nhh, nhl);
    __syncthreads();
    wn_u4 vh[DC_NT], vl[DC_NT], hh[NH], hl[NH];
    for (int c = 0; c < nb32; ++c) {
#pragma unroll
        for (int e = 0; e < DC_NT; ++e) { vh[e] = nh[e]; vl[e] = nl[e]; }
#pragma unroll
        for (int k = 0; k < NH; ++k) { hh[k] = nhh[k]; hl[k] = nhl[k]; }
        if (c + 1 < nb32) loadB(c + 1, nh, nl, nhh, nhl);
#pragma unroll
      for (int tg = 0; tg < ntg; ++tg) {
        const int chunk = c * ntg + tg;
        const int buf = chunk & 1;
        if (chunk + 1 < nchunk) stage_load(chunk + 1);
        const wn_u4* Al = reinterpret_cast<const wn_u4*>(lds[buf]) + lane;
#pragma unroll
        for (int kl = 0; kl < DH_KC; ++kl) {
            const int j = tg * DH_KC + kl;
            if (j < taps) {                       // (a guard, not a break: keeps the loop fully unrolled)
            if (j > 0) {
                // operands one column to the left: e -> e+1, e = 0 from the neighbour lane / halo
                const wn_u4 th = vh[DC_NT - 1], tl = vl[DC_NT - 1];
#pragma unroll
                for (int e = DC_NT - 1; e > 0; --e) { vh[e] = vh[e - 1]; vl[e] = vl[e - 1]; }
                const wn_u4 oh = hh[j > 0 ? j - 1 : 0], ol = hl[j > 0 ? j - 1 : 0];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vh[0][i] = dpp_shr1(oh[i], th[i]);
                    vl[0][i] = dpp_shr1(ol[i], tl[i]);
                }
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const wn_u4 ah = Al[(kl * 4 + mb) * 128], al = Al[(kl * 4 + mb) * 128 + 64];
#pragma unroll
                for (int e = 0; e < DC_NT; ++e) {
                    f4 cc = acc[mb][e];
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, ah), __builtin_bit_cast(wn_h8, vh[e]), cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, ah), __builtin_bit_cast(wn_h8, vl[e]), cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, al), __builtin_bit_cast(wn_h8, vh[e]), cc, 0, 0, 0);
                    acc[mb][e] = cc;
                }
            }
            }
        }
        if (chunk + 1 < nchunk) stage_store(chunk + 1, buf ^ 1);
        __syncthreads();
      }
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int co = 64 * cg + 16 * mb + 4 * kg + rr;
            float* row = yp + (((size_t)b * S + r) * cout + co) * Qp + q0 + DC_NT * n;
            *reinterpret_cast<f4*>(row) = (f4){acc[mb][0][rr] * inv_scale, acc[mb][1][rr] * inv_scale,
                                               acc[mb][2][rr] * inv_scale, acc[mb][3][rr] * inv_scale};
        }
}

// ---------------------------------------------------------------------------
// Phase-group GEMM (round 5): G4 words in, G4 words out -- no phase-major buffer, no interleave launch.
//
// Output sample t = S f + p (frame f, phase p) of a transposed conv is  y[t] = b + sum_j W[S j + r] . x[f + d - j]  with
// r = (p + pL) % S, d = (p + pL) / S: every phase is a GEMM over (tap, channel) on the SAME frame axis f in [0, L), shifted
// by d.  A wave tile is 8 MFMA row blocks x 5 column blocks of frames, the row blocks chosen so that a lane's accumulators
// ARE finished G4 words at consecutive samples:
//   full group  -- FOUR consecutive phases x 32 channels: rows 4 kg .. 4 kg + 3 of two 16-channel blocks = the four pair
//                  slots of G4 group 4 s + kg, at four consecutive samples: four 16-byte stores per plane = 64 contiguous bytes;
//   half group  -- TWO consecutive phases x 64 channels (where the phases of one d do not fill a group of four: with
//                  S = 20, pL = 30 the phases split 10 | 10 = 4 + 4 + 2 each): two G4 groups x 32 contiguous bytes.
// Every group has ONE d, so all row groups walk the same K = taps x cin and do the same work: (4 x 8 + 2 x 4) = 40 row
// groups x 6 frame tiles of 640 = 240 workgroups at 4.8 s -- one round of one workgroup per CU, no tail.  Bias, activation,
// fp16 split and the stores happen in the epilogue: the phase-major round trip (2 x 78.6 MB per utterance) and the
// interleave launch (34 us) are gone.
// Eight waves per workgroup (two per SIMD).  A lane owns 5 consecutive frames and loads, per 32-channel input block, the 8
// words f - 3 .. f + 4 of its rows ONCE: tap m of frame e is register e + 3 - m, no shuffles.  The A fragments of a row
// group (512 KB) stream through LDS by LDS-DMA, four K-steps (one input block) per stage, double buffered, shared by the
// eight waves; row groups are laid out so that the tiles of one group run on one XCD (its L2 holds the group's fragments).
constexpr int PG_TP = 4;                                 // taps (input columns per output frame and block)
constexpr int PG_NT = 5;                                 // column blocks of 16 frames per wave
constexpr int PG_WAVES = 8;
constexpr int PG_FRAMES = PG_WAVES * 16 * PG_NT;         // 640 frames per workgroup
constexpr int PG_KS_WORDS = 8 * 2 * 64 * 4;              // one K-step: 8 row blocks x 2 planes x 64 lanes x 4 words = 16 KB
constexpr int PG_RG_WORDS_PER_BLOCK = PG_TP * PG_KS_WORDS;  // one LDS stage: the four taps of one input block = 64 KB
constexpr int PG_LDS_BYTES = 2 * PG_RG_WORDS_PER_BLOCK * 4; // 128 KB
constexpr int PG_MAXG = 8;                               // phase groups per layer

struct PgArgs {
    const unsigned* x;        // input, G4 rows [B][2][cin / 8][xs] x 16 B, frame f at column DC_XOFF + f
    int cin, xs, L;
    const unsigned* wp;       // phase-group pack (wn_pack_deconv): [row group][input block][tap][mb8][plane][lane][4]
    const float* bias;
    float inv_scale;
    unsigned* y;              // output, G4 rows [B][2][cout / 8][ys] x 16 B, sample t at column yoff + t
    int cout, S, act;
    int64_t ys;
    int yoff;
    unsigned* status;
    int ntiles;               // frame tiles per row group
    int ngroups, nrg;         // phase groups, row groups
    int g_p0[PG_MAXG], g_nph[PG_MAXG], g_d[PG_MAXG], g_rg0[PG_MAXG];   // first phase, phases (4 | 2), column offset, first row group
};

__device__ inline void pg_dma16(const unsigned* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

__global__ __launch_bounds__(PG_WAVES * 64, 1) void deconv_pg_kernel(const PgArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned pg_lds[];
    char* lds = reinterpret_cast<char*>(pg_lds);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    constexpr int NT = PG_NT, TP = PG_TP, NR = NT + TP - 1;
    // blockIdx.x = (rg % 8) + 8 * ((rg / 8) * ntiles + tile): the tiles of a row group share an XCD
    int rg, tile;
    if ((A.nrg & 7) == 0) {
        const int k = blockIdx.x >> 3;
        rg = (blockIdx.x & 7) + 8 * (k / A.ntiles);
        tile = k % A.ntiles;
    } else {
        rg = blockIdx.x / A.ntiles;
        tile = blockIdx.x % A.ntiles;
    }
    int gi = 0;
    for (int k = 1; k < A.ngroups; ++k) if (rg >= A.g_rg0[k]) gi = k;
    const int p0 = A.g_p0[gi], nph = A.g_nph[gi], dcol = A.g_d[gi], sub = rg - A.g_rg0[gi];   // sub: 32-channel (full) / 64-channel (half) slice
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int nb32 = A.cin / 32, ngi = A.cin / 8, ngo = A.cout / 8;
    const int f0 = tile * PG_FRAMES + wave * (16 * NT);           // first frame of the wave
    const unsigned* wsrc = A.wp + (size_t)rg * nb32 * PG_RG_WORDS_PER_BLOCK;

    f4 acc[8][NT];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
        for (int e = 0; e < NT; ++e) acc[mb][e] = (f4){0.f, 0.f, 0.f, 0.f};

    // stage input block c (four K-steps = 64 KB) into buffer `buf`: 8 requests of 1 KB per wave
    auto stage = [&](int c, int buf) {
        const unsigned* src = wsrc + (size_t)c * PG_RG_WORDS_PER_BLOCK;
#pragma unroll
        for (int i = 0; i < PG_RG_WORDS_PER_BLOCK / 256 / PG_WAVES; ++i) {
            const int piece = wave * (PG_RG_WORDS_PER_BLOCK / 256 / PG_WAVES) + i;
            pg_dma16(src + (size_t)(piece * 64 + lane) * 4, lds_base + buf * (PG_RG_WORDS_PER_BLOCK * 4) + piece * 1024);
        }
    };
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(A.x + (size_t)b * 2 * ngi * A.xs * 4), 0, 2 * ngi * A.xs * 16, 0x00020000);
    const int lo_plane = ngi * A.xs * 16;
    // input words of block c: frames f + d - (TP - 1) .. f + d + NT - 1 of the lane's first frame f
    wn_u4 Rh[NR], Rl[NR];
    auto loadB = [&](int c) {
        const int vo = (4 * c + kg) * A.xs * 16 + (DC_XOFF + f0 + NT * n + dcol - (TP - 1)) * 16;
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            Rh[k] = __builtin_bit_cast(wn_u4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + k * 16, 0, 0));
            Rl[k] = __builtin_bit_cast(wn_u4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo + k * 16, lo_plane, 0));
        }
    };

    stage(0, 0);
    loadB(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < nb32; ++c) {
        const int buf = c & 1;
        if (c + 1 < nb32) stage(c + 1, buf ^ 1);                               // (that buffer's readers passed the barrier below)
#pragma unroll
        for (int m = 0; m < TP; ++m) {
            const wn_u4* Al = reinterpret_cast<const wn_u4*>(lds + buf * (PG_RG_WORDS_PER_BLOCK * 4)) + m * (PG_KS_WORDS / 4) + lane;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const wn_u4 ah = Al[mb * 128], al = Al[mb * 128 + 64];
#pragma unroll
                for (int e = 0; e < NT; ++e) {
                    const wn_u4 bh = Rh[e + TP - 1 - m], bl = Rl[e + TP - 1 - m];
                    f4 cc = acc[mb][e];
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, ah), __builtin_bit_cast(wn_h8, bh), cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, ah), __builtin_bit_cast(wn_h8, bl), cc, 0, 0, 0);
                    cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wn_h8, al), __builtin_bit_cast(wn_h8, bh), cc, 0, 0, 0);
                    acc[mb][e] = cc;
                }
            }
        }
        if (c + 1 < nb32) loadB(c + 1);                                        // (two waves per SIMD cover each other's wait)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's share of the next stage has landed
        __syncthreads();
    }
    // ---- epilogue: bias, activation, fp16 split, G4 words of consecutive samples ----
    // row block mb8 holds phase p0 + ph and 16-channel block cb16:  full: ph = mb8 >> 1, cb16 = 2 sub + (mb8 & 1);
    // half: ph = mb8 >> 2, cb16 = 4 sub + (mb8 & 3).  G4 group g = 4 (cb16 >> 1) + kg takes blocks cb16 = 2 s', 2 s' + 1.
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(A.y + (size_t)b * 2 * ngo * A.ys * 4), 0, (int)(2 * ngo * A.ys * 16), 0x00020000);
    const int lo_out = (int)(ngo * A.ys * 16);
    float amax = 0.f;
    const bool half = nph == 2;
    const int cb_first = half ? 4 * sub : 2 * sub;
    float bs[4][4];                                                            // [cb16 - cb_first][row] (full: two blocks)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) bs[k][rr] = (half || k < 2) ? A.bias[16 * (cb_first + k) + 4 * kg + rr] : 0.f;
    // The words leave through an 8 KB LDS patch per wave (the fragment stages are free now), transposed so that FOUR
    // CONSECUTIVE LANES hold the four words of one frame: a store instruction then writes 16 runs of 64 contiguous bytes
    // (half groups: 32 runs of 32) instead of 64 scattered 16-byte pieces -- measured 39 us of a 134 us launch for the
    // scattered form (profiles/r05_upsampler_variants.txt).
    char* patch = lds + wave * 8192;
    const int fr = f0 + NT * (lane >> 2);                  // frame of this lane in the store pass (+ e)
    const int w2r = lane & 3;                              // ... and its word
    const int phr = half ? (w2r & 1) : w2r, sgr = half ? (w2r >> 1) : 0;
#pragma unroll
    for (int e = 0; e < NT; ++e) {
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            // word w2 of this frame: full -- phase w2 of the one G4 group; half -- phase w2 & 1 of G4 group w2 >> 1
            wn_u4 hw, lw;
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                const int blk = slot >> 1, r0 = (slot & 1) * 2;
                // (compile-time register indices: both variants are evaluated and selected)
                const float a0 = half ? acc[(4 * (w2 & 1) + 2 * (w2 >> 1) + blk) & 7][e][r0] : acc[(2 * w2 + blk) & 7][e][r0];
                const float a1 = half ? acc[(4 * (w2 & 1) + 2 * (w2 >> 1) + blk) & 7][e][r0 + 1] : acc[(2 * w2 + blk) & 7][e][r0 + 1];
                const float b0 = half ? bs[2 * (w2 >> 1) + blk][r0] : bs[blk][r0];
                const float b1 = half ? bs[2 * (w2 >> 1) + blk][r0 + 1] : bs[blk][r0 + 1];
                const float v0 = apply_act(fmaf(a0, A.inv_scale, b0), A.act);
                const float v1 = apply_act(fmaf(a1, A.inv_scale, b1), A.act);
                unsigned a, c2;
                wn_split_pair_t(v0, v1, a, c2, amax);
                hw[slot] = a;
                lw[slot] = c2;
            }
            *reinterpret_cast<wn_u4*>(patch + ((kg * 16 + n) * 4 + w2) * 16) = hw;
            *reinterpret_cast<wn_u4*>(patch + 4096 + ((kg * 16 + n) * 4 + w2) * 16) = lw;
        }
        const int f = fr + e;
        const int64_t tcol = A.yoff + (int64_t)A.S * f + p0 + phr;
#pragma unroll
        for (int kr = 0; kr < 4; ++kr) {
            const int g = 4 * ((cb_first >> 1) + sgr) + kr;
            const int vo = f < A.L ? (int)(((int64_t)g * A.ys + tcol) * 16) : (int)0x80000000;
            const wn_u4 hw = *reinterpret_cast<const wn_u4*>(patch + (kr * 64 + lane) * 16);
            const wn_u4 lw = *reinterpret_cast<const wn_u4*>(patch + 4096 + (kr * 64 + lane) * 16);
            buf_st4(hw, ry, vo, 0);
            buf_st4(lw, ry, vo, lo_out);
        }
    }
    wn_range_flag(amax, A.status);
}

// channel-major [B][C][T] (row stride cs) -> reference layout [B][T][C]
__global__ void cm_to_tm_kernel(const float* __restrict__ in, float* __restrict__ out,
                                int C, int64_t T, int64_t cs) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int64_t t0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i;
        const int64_t t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? in[((size_t)b * C + c) * cs + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int64_t t = t0 + i;
        const int c = c0 + tx;
        if (t < T && c < C) out[((size_t)b * T + t) * C + c] = tile[tx][i];
    }
}

// Same, but the output is written as split-fp16 pair planes (wn_codec.h): one workgroup =
// one channel PAIR x 64 q columns; hi words to row cp, lo words to row cout/2 + cp.
__global__ __launch_bounds__(256) void deconv_interleave_split_kernel(
    const float* __restrict__ yp, const float* __restrict__ bias, unsigned* __restrict__ y,
    int cout, int Qp, int64_t ys, int yoff, int L, int S, int pL, int act, unsigned* __restrict__ status) {
    __shared__ float tile[2][DI_MAXS][DC_QT + 1];
    const int cp = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * DC_QT;
    for (int i = threadIdx.x; i < 2 * S * DC_QT; i += 256) {
        const int h = i / (S * DC_QT), j = i - h * S * DC_QT;
        const int r = j / DC_QT, q = j - r * DC_QT;
        tile[h][r][q] = yp[(((size_t)b * S + r) * cout + 2 * cp + h) * Qp + q0 + q];
    }
    __syncthreads();
    const float b0 = bias[2 * cp], b1 = bias[2 * cp + 1];
    const int64_t SL = (int64_t)S * L;
    unsigned* yh = y + ((size_t)b * cout + cp) * ys + yoff;
    unsigned* yl = yh + (size_t)(cout / 2) * ys;
    const int64_t n0 = (int64_t)S * q0 - pL;
    float amax = 0.f;
    for (int i = threadIdx.x; i < S * DC_QT; i += 256) {
        const int q = i / S, r = i - q * S;
        const int64_t nn = n0 + i;
        if (nn >= 0 && nn < SL) {
            unsigned hw, lw;
            wn_split_pair_t(apply_act(tile[0][r][q] + b0, act), apply_act(tile[1][r][q] + b1, act), hw, lw, amax);
            yh[nn] = hw;
            yl[nn] = lw;
        }
    }
    wn_range_flag(amax, status);
}

// Last layer of the split-fp16 path: weave the phases AND emit the G4 layout of wn_iaf_h.hip
// (4 pair rows interleaved per 16-byte word group).  One workgroup = one group (8 channels) x
// 32 q columns = 32*S consecutive output samples, 16-byte stores.
// SC: the stride as a compile-time constant (0 = run-time value): the index arithmetic of both loops divides by it
constexpr int DG_Q = 32;
template <int SC>
__global__ __launch_bounds__(256) void deconv_interleave_g4_kernel(
    const float* __restrict__ yp, const float* __restrict__ bias, unsigned* __restrict__ y,
    int cout, int Qp, int64_t ys, int yoff, int L, int S_rt, int pL, int act, unsigned* __restrict__ status,
    int zero_pads) {
    const int S = SC ? SC : S_rt;
    __shared__ float tile[8][DI_MAXS][DG_Q + 1];
    const int g = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * DG_Q;
    if (zero_pads && (blockIdx.x == 0 || blockIdx.x + 1 == gridDim.x)) {
        // an intermediate layer: the next GEMM reads yoff zero words left of sample 0 and zeros right of sample S L up
        // to the row stride -- written here instead of by a memset of the whole buffer in front of the launch
        wn_u4* rh = reinterpret_cast<wn_u4*>(y + ((size_t)b * cout * ys)) + (size_t)g * ys;
        wn_u4* rl = rh + (size_t)(cout / 8) * ys;
        const int64_t lo = blockIdx.x == 0 ? 0 : yoff + (int64_t)S * L, hi = blockIdx.x == 0 ? yoff : ys;
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) rh[i] = rl[i] = (wn_u4){0u, 0u, 0u, 0u};
    }
    const int prbase = 16 * (g >> 2) + 2 * (g & 3);
    // 16-byte loads: a (channel, phase) row of the tile is 128 contiguous bytes of the phase-major buffer
    for (int i = threadIdx.x; i < 8 * S * (DG_Q / 4); i += 256) {
        const int c8 = i / (S * (DG_Q / 4)), j = i - c8 * S * (DG_Q / 4);
        const int r = j / (DG_Q / 4), q = 4 * (j - r * (DG_Q / 4));
        const int slot = c8 >> 1;
        const int ch = 2 * (prbase + 8 * (slot >> 1) + (slot & 1)) + (c8 & 1);
        const f4 v = *reinterpret_cast<const f4*>(yp + (((size_t)b * S + r) * cout + ch) * Qp + q0 + q);
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[c8][r][q + k] = v[k];
    }
    __syncthreads();
    float bs[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int slot = c8 >> 1;
        bs[c8] = bias[2 * (prbase + 8 * (slot >> 1) + (slot & 1)) + (c8 & 1)];
    }
    const int ng = cout / 8;
    const int64_t SL = (int64_t)S * L;
    wn_u4* yh = reinterpret_cast<wn_u4*>(y + ((size_t)b * cout * ys)) + (size_t)g * ys + yoff;
    wn_u4* yl = yh + (size_t)ng * ys;
    const int64_t n0 = (int64_t)S * q0 - pL;
    float amax = 0.f;
    for (int i = threadIdx.x; i < S * DG_Q; i += 256) {
        const int q = i / S, r = i - q * S;
        const int64_t nn = n0 + i;
        if (nn >= 0 && nn < SL) {
            wn_u4 hw, lw;
#pragma unroll
            for (int slot = 0; slot < 4; ++slot) {
                unsigned a, c2;
                wn_split_pair_t(apply_act(tile[2 * slot][r][q] + bs[2 * slot], act),
                                apply_act(tile[2 * slot + 1][r][q] + bs[2 * slot + 1], act), a, c2, amax);
                hw[slot] = a;
                lw[slot] = c2;
            }
            yh[nn] = hw;
            yl[nn] = lw;
        }
    }
    wn_range_flag(amax, status);
}

}  // namespace

// ---------------------------------------------------------------------------
int wn_pack_deconv(wn_handle* h, std::vector<float>& blob) {
    const wn_config& c = h->cfg;
    for (auto& sp : h->stacks) {
        int cin = c.n_mel;
        for (int j = 0; j < c.n_deconv; ++j) {
            DeconvLayerPack lp;
            lp.cin = cin;
            lp.cout = c.deconv_width;
            lp.K = c.deconv_filter[j];
            lp.S = c.deconv_stride[j];
            lp.pL = (lp.K - lp.S) / 2;
            lp.taps = lp.K / lp.S;
            if (cin % 16) return wn_fail(h, WN_EINVAL, "deconv: input channels %d not a multiple of 16", cin);
            std::string scope = (sp.prefix.empty() ? std::string() : sp.prefix + "/") +
                                (c.use_resize_conv ? "resize_conv_" : "trans_conv_") + std::to_string(j + 1);
            std::vector<float> W;                                              // [K][cout][cin]
            if (c.use_resize_conv) {
                // masked.py:294-322: u[t] = x[t / S] (nearest neighbour), y[t] = b + sum_k u[t+k-pl] Wr[k],
                // pl = (fl-1)/2 (SAME).  With t + pr = S q + r (pr = fl-1-pl) this is the phase GEMM
                //   y = sum_j x[q-j] . Weff[S j + r],   Weff[S j + r] = sum_{k: -floor((r+k-(fl-1))/S) = j} Wr[k]
                // i.e. a transposed conv with the summed kernel Weff, crop offset pr, zeros outside x.
                const int fl = lp.K, S = lp.S;
                std::vector<float> Wr = wn_get_kernel(h, scope, "W", false);   // [fl][cin][cout]
                int taps = (fl - 1 + S - 1) / S + 1;
                if ((taps * (cin / 16)) % 2) ++taps;                           // even K-step count for the fp16 GEMM
                lp.taps = taps;
                lp.K = S * taps;
                lp.pL = fl - 1 - (fl - 1) / 2;
                W.assign((size_t)lp.K * lp.cout * cin, 0.f);
                for (int r = 0; r < S; ++r)
                    for (int k = 0; k < fl; ++k) {
                        const int m = r + k - (fl - 1);                        // <= S-1
                        const int jt = m >= 0 ? 0 : (-m + S - 1) / S;          // -floor(m/S)
                        for (int co = 0; co < lp.cout; ++co)
                            for (int ci = 0; ci < cin; ++ci)
                                W[((size_t)(S * jt + r) * lp.cout + co) * cin + ci] +=
                                    Wr[((size_t)k * cin + ci) * lp.cout + co];
                    }
            } else {
                W = wn_get_kernel(h, scope, "kernel", true);
            }
            const std::vector<float>& bias = h->vars.at(scope + (c.use_resize_conv ? "/biases" : "/bias")).data;
            const int nmb = lp.cout / 16, cblk = cin / 16, nks4 = lp.taps * cblk;
            blob.resize(align_up(blob.size(), 64));
            lp.w_off = blob.size();
            blob.resize(blob.size() + (size_t)lp.S * nks4 * nmb * 256);
            float* P = blob.data() + lp.w_off;
            for (int r = 0; r < lp.S; ++r)
                for (int ks4 = 0; ks4 < nks4; ++ks4) {
                    const int tap = ks4 / cblk, c4 = ks4 % cblk;
                    for (int mb = 0; mb < nmb; ++mb)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int jj = 0; jj < 4; ++jj) {
                                const int i = lane & 15, kq = lane >> 4;
                                const int co = 16 * mb + i, ci = 16 * c4 + 4 * jj + kq;
                                const int k = lp.S * tap + r;
                                P[((((size_t)r * nks4 + ks4) * nmb + mb) * 64 + lane) * 4 + jj] =
                                    W[((size_t)k * lp.cout + co) * cin + ci];
                            }
                }
            lp.b_off = blob.size();
            blob.insert(blob.end(), bias.begin(), bias.end());
            // split-fp16 A fragments: [S][ks][mb][plane][lane][4], K-step = two 16-channel blocks of
            // the flattened (tap, block) axis
            lp.w_off_h = 0;
            lp.inv_scale_h = 1.f;
            if ((lp.taps * (cin / 16)) % 2 == 0 && cin % 16 == 0) {
                const int nb16 = cin / 16, nks = lp.taps * nb16 / 2;
                const float sc = pick_scale(W.data(), W.size());
                lp.inv_scale_h = 1.0f / sc;
                blob.resize(align_up(blob.size(), 64));
                lp.w_off_h = blob.size();
                blob.resize(blob.size() + (size_t)lp.S * nks * nmb * 512);
                unsigned* PH = reinterpret_cast<unsigned*>(blob.data() + lp.w_off_h);
                for (int r = 0; r < lp.S; ++r)
                    for (int ks = 0; ks < nks; ++ks)
                        for (int mb = 0; mb < nmb; ++mb)
                            pack_afrag(PH + (((size_t)r * nks + ks) * nmb + mb) * 512, [&](int e, int kg, int i16) {
                                const int g = 2 * ks + (e >> 2);
                                const int tap = g / nb16, blk = g - tap * nb16;
                                const int ci = 16 * blk + 4 * kg + (e & 3);
                                return sc * W[((size_t)(lp.S * tap + r) * lp.cout + 16 * mb + i16) * cin + ci];
                            });
            }
            // phase-group pack (deconv_pg_kernel): [row group][input block c][tap m][row block mb8][plane][lane][4].
            // Phase p uses kernel offset r = (p + pL) % S and input frame f + d - m, d = (p + pL) / S.  The phases of one d
            // are cut into groups of four (full: 4 phases x 32 channels per row group) and a remainder of two (half:
            // 2 phases x 64 channels per row group), so every row group has one d and 8 row blocks.
            if (lp.w_off_h && !c.use_resize_conv && lp.taps == 4 && cin % 32 == 0 && lp.cout % 64 == 0) {
                std::vector<int> gp0, gnph;
                bool ok = true;
                for (int p = 0; p < lp.S;) {
                    const int d = (p + lp.pL) / lp.S;
                    int q = p;
                    while (q < lp.S && (q + lp.pL) / lp.S == d) ++q;          // phases [p, q) share d
                    if ((q - p) & 1) { ok = false; break; }
                    for (; q - p >= 4; p += 4) { gp0.push_back(p); gnph.push_back(4); }
                    if (q - p == 2) { gp0.push_back(p); gnph.push_back(2); p += 2; }
                }
                if (ok && (int)gp0.size() <= 8) {
                    const int nb32 = cin / 32;
                    const float sc = 1.0f / lp.inv_scale_h;
                    lp.pg_n = (int)gp0.size();
                    int nrg = 0;
                    for (int g = 0; g < lp.pg_n; ++g) {
                        lp.pg_p0[g] = gp0[g]; lp.pg_nph[g] = gnph[g]; lp.pg_d[g] = (gp0[g] + lp.pL) / lp.S; lp.pg_rg0[g] = nrg;
                        nrg += gnph[g] == 4 ? lp.cout / 32 : lp.cout / 64;
                    }
                    lp.pg_nrg = nrg;
                    blob.resize(align_up(blob.size(), 64));
                    lp.w_off_pg = blob.size();
                    blob.resize(blob.size() + (size_t)nrg * nb32 * 4 * 4096);
                    unsigned* PG = reinterpret_cast<unsigned*>(blob.data() + lp.w_off_pg);
                    for (int g = 0; g < lp.pg_n; ++g) {
                        const int nsub = gnph[g] == 4 ? lp.cout / 32 : lp.cout / 64;
                        for (int sub = 0; sub < nsub; ++sub)
                            for (int cb = 0; cb < nb32; ++cb)
                                for (int m = 0; m < 4; ++m)
                                    for (int mb8 = 0; mb8 < 8; ++mb8) {
                                        const int ph = gnph[g] == 4 ? mb8 >> 1 : mb8 >> 2;
                                        const int cb16 = gnph[g] == 4 ? 2 * sub + (mb8 & 1) : 4 * sub + (mb8 & 3);
                                        const int r = (gp0[g] + ph + lp.pL) % lp.S;
                                        pack_afrag(PG + (((((size_t)(lp.pg_rg0[g] + sub)) * nb32 + cb) * 4 + m) * 8 + mb8) * 512,
                                                   [&](int e, int kg, int i16) {
                                                       const int ci = 32 * cb + 16 * (e >> 2) + 4 * kg + (e & 3);
                                                       return sc * W[((size_t)(lp.S * m + r) * lp.cout + 16 * cb16 + i16) * cin + ci];
                                                   });
                                    }
                    }
                }
            }
            // fp32 frame-axis GEMM (wn_iaf_f.hip): output phase p = kernel offset r = (p + pL) % S on input frames f + d - j,
            // d = (p + pL) / S; one row block per (phase, 64-channel group) inside the fp32 pack above
            if (!c.use_resize_conv && cin == 256 && lp.taps >= 1 && lp.taps <= 5 && lp.cout % 64 == 0 && (lp.pL + lp.S - 1) / lp.S <= 4) {
                const int ncg = lp.cout / 64;
                std::vector<unsigned> tab;
                for (int p = 0; p < lp.S; ++p)
                    for (int cg = 0; cg < ncg; ++cg) {
                        const int r = (p + lp.pL) % lp.S, d = (p + lp.pL) / lp.S;
                        tab.push_back((unsigned)(lp.w_off + ((size_t)r * nks4 * nmb + (size_t)cg * 4) * 256));
                        tab.push_back(((unsigned)p << 8) | (unsigned)d);
                    }
                blob.resize(align_up(blob.size(), 64));
                lp.tab_f_off = blob.size();
                lp.tab_f_R = (int)tab.size() / 2;
                blob.resize(blob.size() + align_up(tab.size(), 4));
                memcpy(blob.data() + lp.tab_f_off, tab.data(), tab.size() * sizeof(unsigned));
            }
            sp.layers.push_back(lp);
            cin = lp.cout;
        }
    }
    return WN_OK;
}

int wn_deconv_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(deconv_pg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  PG_LDS_BYTES));
    return wn_iaf_f_set_attrs(h);       // the fp32 GEMMs with the activation tile in LDS (students and fp32 teachers' upsampler)
}

// phase-major GEMM output of the largest layer: [B][S][cout][Qp]
static size_t dc_phase_floats(const wn_handle* h, int B, int F) {
    const wn_config& c = h->cfg;
    size_t mx = 0;
    int64_t L = F;
    for (int j = 0; j < c.n_deconv; ++j) {
        const size_t Qp = (size_t)((L + 5 + DC_QW - 1) / DC_QW) * DC_QW;   // Q <= L + 5 for <= 8 taps
        mx = std::max(mx, (size_t)B * c.deconv_stride[j] * c.deconv_width * Qp);
        L *= c.deconv_stride[j];
    }
    return mx;
}

// scratch: channel-major mel + every intermediate layer output + the phase-major buffer
size_t wn_deconv_scratch_bytes(const wn_handle* h, int B, int F) {
    const wn_config& c = h->cfg;
    size_t fl = (size_t)B * c.n_mel * dc_row_stride(F);
    int64_t L = F;
    for (int j = 0; j + 1 < c.n_deconv; ++j) {
        L *= c.deconv_stride[j];
        fl += (size_t)B * c.deconv_width * dc_row_stride((int)L);
    }
    fl += dc_phase_floats(h, B, F);
    return align_up(fl * sizeof(float), 256);
}

int wn_run_deconv(wn_handle* h, int si, const float* mel, int B, int F, float* enc_cm,
                  int64_t enc_stride, void* scratch, hipStream_t st, bool split_out, unsigned* status, int prec) {
    const wn_config& c = h->cfg;
    const DeconvStackPack& sp = h->stacks[si];
    // split-fp16 GEMMs when the call runs in f16x3 mode and every layer's shape supports them
    bool h_gemm = (prec < 0 ? c.precision : prec) == WN_PREC_F16X3;
    for (const DeconvLayerPack& lp : sp.layers) h_gemm = h_gemm && lp.w_off_h != 0;
    float* buf = reinterpret_cast<float*>(scratch);
    int xs = dc_row_stride(F);
    if (h_gemm) {
        dim3 g((xs + 255) / 256, c.n_mel / 2, B);
        hipLaunchKernelGGL(mel_to_split_kernel, g, dim3(256), 0, st, mel, reinterpret_cast<unsigned*>(buf), F,
                           c.n_mel, xs, status);
    } else {
        dim3 g((xs + 255) / 256, c.n_mel, B);
        hipLaunchKernelGGL(mel_to_cm_kernel, g, dim3(256), 0, st, mel, buf, F, c.n_mel, xs);
    }
    const float* x = buf;
    float* next = buf + (size_t)B * c.n_mel * xs;
    // the phase-major buffer sits at the end of the scratch area
    float* phase = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + wn_deconv_scratch_bytes(h, B, F)) -
                   dc_phase_floats(h, B, F);
    int L = F;
    bool in_g4 = false;
    for (int j = 0; j < c.n_deconv; ++j) {
        const DeconvLayerPack& lp = sp.layers[j];
        const bool last = (j + 1 == c.n_deconv);
        const int Lout = L * lp.S;
        float* y;
        int64_t ys;
        int yoff;
        if (last) {
            y = enc_cm; ys = enc_stride; yoff = 0;
        } else {
            y = next; ys = dc_row_stride(Lout); yoff = DC_XOFF;
        }
        const int Q = L + (lp.pL + lp.S - 1) / lp.S + 1;      // phase columns q = (t + pL) / S
        const int Qp = ((Q + DC_QW - 1) / DC_QW) * DC_QW;
        // the fp16 GEMM of the NEXT layer reads G4 words when its input width allows it
        const bool next_g4 = h_gemm && !last && (lp.cout % 32 == 0);
        // zero pads of an intermediate output: the G4 interleave writes them itself, the other forms get a memset
        if (!last && !next_g4) WN_HIP(h, hipMemsetAsync(y, 0, (size_t)B * lp.cout * ys * sizeof(float), st));
        // (h->dc_no_pg: WN_DC_NO_PG=1 at wn_create keeps the phase-major GEMM + interleave of rounds 1-4: A/B runs, cross-form test)
        if (h_gemm && in_g4 && last && split_out && lp.w_off_pg && !h->dc_no_pg) {
            PgArgs A{};
            A.x = reinterpret_cast<const unsigned*>(x);
            A.cin = lp.cin; A.xs = xs; A.L = L;
            A.wp = reinterpret_cast<const unsigned*>(h->d_blob + lp.w_off_pg);
            A.bias = h->d_blob + lp.b_off;
            A.inv_scale = lp.inv_scale_h;
            A.y = reinterpret_cast<unsigned*>(y);
            A.cout = lp.cout; A.S = lp.S; A.act = c.upsample_act;
            A.ys = ys; A.yoff = yoff; A.status = status;
            A.ngroups = lp.pg_n; A.nrg = lp.pg_nrg;
            for (int g = 0; g < lp.pg_n; ++g) {
                A.g_p0[g] = lp.pg_p0[g]; A.g_nph[g] = lp.pg_nph[g]; A.g_d[g] = lp.pg_d[g]; A.g_rg0[g] = lp.pg_rg0[g];
            }
            A.ntiles = (L + PG_FRAMES - 1) / PG_FRAMES;
            hipLaunchKernelGGL(deconv_pg_kernel, dim3(lp.pg_nrg * A.ntiles, B), dim3(PG_WAVES * 64), PG_LDS_BYTES, st, A);
            in_g4 = false;
            x = y; xs = (int)ys; next = y + (size_t)B * lp.cout * ys; L = Lout;
            continue;
        }
        if (h_gemm) {
            dim3 g(Qp / DC_QW, lp.S, B * (lp.cout / 64));
            const unsigned* xin = reinterpret_cast<const unsigned*>(x);
            const unsigned* wfr = reinterpret_cast<const unsigned*>(h->d_blob + lp.w_off_h);
            // G4 input: every column is loaded once per channel block and shifted in registers for the
            // other taps (instantiated for the tap counts of the supported configs)
            if (in_g4 && lp.taps == 4)
                hipLaunchKernelGGL(deconv_mfma_hs_kernel<4>, g, dim3(256), 0, st, xin, lp.cin, xs, wfr, phase, lp.cout,
                                   Qp, lp.S, lp.inv_scale_h);
            else if (in_g4 && lp.taps == 5)
                hipLaunchKernelGGL(deconv_mfma_hs_kernel<5>, g, dim3(256), 0, st, xin, lp.cin, xs, wfr, phase, lp.cout,
                                   Qp, lp.S, lp.inv_scale_h);
            else if (in_g4 && lp.taps == 6)
                hipLaunchKernelGGL(deconv_mfma_hs_kernel<6>, g, dim3(256), 0, st, xin, lp.cin, xs, wfr, phase, lp.cout,
                                   Qp, lp.S, lp.inv_scale_h);
            else {
                auto kern = in_g4 ? deconv_mfma_h_kernel<true> : deconv_mfma_h_kernel<false>;
                hipLaunchKernelGGL(kern, g, dim3(256), 0, st, xin, lp.cin, xs, wfr, phase, lp.cout, Qp, lp.S, lp.taps,
                                   lp.inv_scale_h);
            }
        } else if (lp.tab_f_off && !h->dc_no_pg) {
            // fp32, 256 input channels: the frame-axis GEMM with the input tile in LDS (wn_iaf_f.hip); its phase-major output
            // is indexed by (output phase, frame), so the interleave below runs with pL = 0
            const int Lp = (L + 63) / 64 * 64;
            wn_deconv_f_gemm(h, x, xs, DC_XOFF, reinterpret_cast<const unsigned*>(h->d_blob + lp.tab_f_off), lp.tab_f_R, lp.taps,
                             (lp.cout / 16) * 256, phase, lp.S, lp.cout, L, Lp, B, st);
            dim3 gi(Lp / DC_QT, lp.cout, B);
            hipLaunchKernelGGL(deconv_interleave_kernel, gi, dim3(256), 0, st, phase, h->d_blob + lp.b_off, y,
                               lp.cout, Lp, ys, yoff, L, lp.S, 0, c.upsample_act);
            in_g4 = false;
            x = y;
            xs = (int)ys;
            next = y + (size_t)B * lp.cout * ys;
            L = Lout;
            continue;
        } else {
            const int zc = (lp.cout + 255) / 256;
            dim3 g(Qp / DC_QT, lp.S, B * zc);
            hipLaunchKernelGGL(deconv_mfma_kernel, g, dim3(256), 0, st, x, lp.cin, xs, h->d_blob + lp.w_off, phase,
                               lp.cout, Qp, lp.S, lp.taps, zc);
        }
        // intermediate outputs feed the next GEMM; the last one is split (G4) only when the caller
        // consumes the split layout
        if ((last && split_out) || next_g4) {
            dim3 gi(Qp / DG_Q, lp.cout / 8, B);
            auto ik = lp.S == 20 ? deconv_interleave_g4_kernel<20> : lp.S == 10 ? deconv_interleave_g4_kernel<10>
                                                                                 : deconv_interleave_g4_kernel<0>;
            hipLaunchKernelGGL(ik, gi, dim3(256), 0, st, phase, h->d_blob + lp.b_off,
                               reinterpret_cast<unsigned*>(y), lp.cout, Qp, ys, yoff, L, lp.S, lp.pL, c.upsample_act,
                               status, last ? 0 : 1);
        } else if (!last && h_gemm) {
            dim3 gi(Qp / DC_QT, lp.cout / 2, B);
            hipLaunchKernelGGL(deconv_interleave_split_kernel, gi, dim3(256), 0, st, phase, h->d_blob + lp.b_off,
                               reinterpret_cast<unsigned*>(y), lp.cout, Qp, ys, yoff, L, lp.S, lp.pL,
                               c.upsample_act, status);
        } else {
            dim3 gi(Qp / DC_QT, lp.cout, B);
            hipLaunchKernelGGL(deconv_interleave_kernel, gi, dim3(256), 0, st, phase, h->d_blob + lp.b_off, y,
                               lp.cout, Qp, ys, yoff, L, lp.S, lp.pL, c.upsample_act);
        }
        in_g4 = next_g4;
        x = y;
        xs = (int)ys;
        next = y + (size_t)B * lp.cout * ys;
        L = Lout;
    }
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

extern "C" int wn_deconv(wn_handle* h, const char* scope, const float* mel, int B, int F, float* enc,
                         void* ws, size_t ws_bytes, void* stream) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_deconv: null handle");
    if (!h->finalized) return wn_fail(h, WN_ESTATE, "wn_deconv: call wn_finalize first");
    if (!mel || !enc || !ws || B < 1 || F < 1) return wn_fail(h, WN_EINVAL, "wn_deconv: bad argument");
    const WnWork work(h);
    int si = -1;
    for (size_t i = 0; i < h->stacks.size(); ++i)
        if (h->stacks[i].prefix == (scope ? scope : "")) si = (int)i;
    if (si < 0) return wn_fail(h, WN_ENOENT, "wn_deconv: no deconv stack with scope '%s'", scope ? scope : "");
    const int64_t Tn = (int64_t)F * h->frame_shift;
    const size_t cm_bytes = align_up((size_t)B * h->cfg.deconv_width * Tn * sizeof(float), 256);
    // The first WN_WS_HEAD bytes of a workspace are the range-guard words of the generate calls made on it
    // (wn_iaf_range_status*): a student's wn_deconv on the SAME buffer must leave them alone.
    const size_t need = WN_WS_HEAD + cm_bytes + wn_deconv_scratch_bytes(h, B, F);
    if (ws_bytes < need)
        return wn_fail(h, WN_ENOMEM, "wn_deconv: workspace %zu < %zu bytes", ws_bytes, need);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* cm = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + WN_WS_HEAD);
    int rc = wn_run_deconv(h, si, mel, B, F, cm, Tn, reinterpret_cast<char*>(ws) + WN_WS_HEAD + cm_bytes, st);
    if (rc) return rc;
    dim3 g((unsigned)((Tn + 31) / 32), (h->cfg.deconv_width + 31) / 32, B);
    hipLaunchKernelGGL(cm_to_tm_kernel, g, dim3(256), 0, st, cm, enc, h->cfg.deconv_width, Tn, Tn);
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}
