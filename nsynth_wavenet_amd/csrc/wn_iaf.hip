// Parallel-WaveNet (IAF) student generation on gfx950:
//   wavenet/parallel_wavenet.py:200-287 (_create_iaf), :289-345 (feed_forward),
//   :347-359 (_clip_quant_scale), :172-184 (noise sources).
//
// Device data layout (all fp32, caller-provided workspace):
//   l   ping/pong  [B][64][LP + T]   residual stream, CHANNEL-MAJOR rows, LP zero
//                                    columns on the left so t-d / t-2d never branch
//   enc            [B][256][TE]      upsampled mel (wn_deconv.hip), channel-major
//   x              [B][XP + T]       flow input/output, XP zero columns on the left
//   x0, M, S       [B][T]            noise, mean_tot, scale_tot
//
// One residual layer (parallel_wavenet.py:227-254) is ONE kernel:
//   h[o,t]  = bd+bc + sum_k Wd[tap k] l[:, t-(2-k)d] + Wc enc[:, t+c0]     (K = 448)
//   g[c,t]  = sigmoid(h[c,t]) * tanh(h[c+32,t])
//   l'[o,t] = l[o,t] + br + Wr g[:, t]                                       (K = 32)
// as v_mfma_f32_16x16x4_f32 with the WEIGHTS as the A operand (M = output channel)
// and the ACTIVATIONS as the B operand (N = time).  With that orientation
//   * B-operand loads are time-contiguous (coalesced) rows, dilation is a column
//     offset, the centre crop (wavenet.py:76-85) is a pointer offset;
//   * the accumulator holds, per lane, one time step and channels {16mb+4q+r}:
//     sigmoid channel c and tanh channel c+32 sit in the same lane and register
//     index (row blocks mb and mb+2), so the gate is lane-local;
//   * the gated registers ARE the B operand of the residual 1x1 (K order is
//     chosen to match the accumulator layout), and the tap-t loads ARE its C-in.
// Weights are packed on the host in A-fragment order and staged once per
// workgroup in LDS (120.5 KB); workgroups are persistent over 64-sample tiles.
#include "wn_internal.h"
#include "wn_codec.h"

namespace {

constexpr float EXP_M9 = 1.2340980408667956e-4f;
constexpr float EXP_7 = 1096.6331584284585f;

__device__ inline f4 mfma4(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ inline float sigmoidf_(float a) { return __builtin_amdgcn_rcpf(1.f + __expf(-a)); }
__device__ inline float tanhf_(float a) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * a) + 1.f); }

// ---------------- noise (parallel_wavenet.py:172-184) ----------------
// 4 samples per thread.  logistic: log u - log(1-u), u ~ U(1e-5, 1-1e-5); gauss: Box-Muller.
__device__ inline void iaf_noise_body(float* __restrict__ x0, float* __restrict__ x, int64_t T, int XR,
                                      uint64_t seed, int gauss, int bx, int b) {
    const int64_t i4 = (int64_t)bx * blockDim.x + threadIdx.x;
    if (i4 * 4 >= T) return;
    uint32_t c[4] = {(uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)b, 0x49414630u};
    wn_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    float v[4];
    if (!gauss) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u01 = (float)(c[e] >> 8) * (1.0f / 16777216.0f);
            const float u = u01 * (1.f - 2e-5f) + 1e-5f;
            v[e] = logf(u) - logf(1.f - u);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            const float u1 = ((float)(c[e] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
            const float u2 = (float)(c[e + 1] >> 8) * (1.0f / 16777216.0f);
            const float rr = sqrtf(-2.f * logf(u1));
            v[e] = rr * cosf(6.283185307179586f * u2);
            v[e + 1] = rr * sinf(6.283185307179586f * u2);
        }
    }
    *reinterpret_cast<f4*>(x0 + (size_t)b * T + i4 * 4) = (f4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f4*>(x + (size_t)b * XR + IAF_XP + i4 * 4) = (f4){v[0], v[1], v[2], v[3]};
}

__global__ void iaf_noise_kernel(float* __restrict__ x0, float* __restrict__ x, int64_t T, int XR,
                                 uint64_t seed, int gauss) {
    iaf_noise_body(x0, x, T, XR, seed, gauss, blockIdx.x, blockIdx.y);
}

__device__ inline void iaf_copy_noise_body(const float* __restrict__ noise, float* __restrict__ x, int64_t T, int XR,
                                           int bx, int b) {
    const int64_t i4 = (int64_t)bx * blockDim.x + threadIdx.x;
    if (i4 * 4 >= T) return;
    *reinterpret_cast<f4*>(x + (size_t)b * XR + IAF_XP + i4 * 4) =
        *reinterpret_cast<const f4*>(noise + (size_t)b * T + i4 * 4);
}

__global__ void iaf_copy_noise_kernel(const float* __restrict__ noise, float* __restrict__ x, int64_t T, int XR) {
    iaf_copy_noise_body(noise, x, T, XR, blockIdx.x, blockIdx.y);
}

// zero the left pads of `rows` rows (row stride rs floats, pad floats each)
// zero left pads of both activation buffers and of the flow input, one launch (bz picks)
// dl_rj > 0: lB holds the DL layout of wn_iaf_g.hip (32 residue rows of dl_rj 16-byte words per group row, 64 zero
// words in front of each) instead of one left pad per row
__device__ inline void zero_pads_body(float* __restrict__ lA, float* __restrict__ lB, int64_t rs, int pad, int rows,
                                      float* __restrict__ x, int64_t xrs, int xpad, int xrows, unsigned* __restrict__ status,
                                      int dl_rj, int bx, int by, int bz, int gy) {
    const int c = bx * blockDim.x + threadIdx.x;
    if (by == 0 && c == 0 && bz == 0) *status = 0u;  // range-guard word of this call (wn_codec.h)
    // grid.y is capped (65 535 rows per launch dimension): rows are walked
    for (int row = by; row < max(rows, xrows); row += gy) {
        if (bz < 2) {
            float* p = bz ? lB : lA;
            if (bz == 1 && dl_rj > 0) {
                // 32 x 64 words x 4 floats = pad floats again: float c -> residue c / 256, float c % 256 of its pad
                if (row < rows && c < pad) p[(size_t)row * rs + (size_t)(c >> 8) * dl_rj * 4 + (c & 255)] = 0.f;
            } else if (row < rows && c < pad) p[(size_t)row * rs + c] = 0.f;
        } else if (row < xrows && c < xpad) {
            x[(size_t)row * xrs + c] = 0.f;
        }
    }
}

__global__ void zero_pads_kernel(float* __restrict__ lA, float* __restrict__ lB, int64_t rs, int pad, int rows,
                                 float* __restrict__ x, int64_t xrs, int xpad, int xrows, unsigned* __restrict__ status,
                                 int dl_rj) {
    zero_pads_body(lA, lB, rs, pad, rows, x, xrs, xpad, xrows, status, dl_rj, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}

// The call's prologue as ONE launch (round 5): the zero pads and the flow input (noise drawn here, or the caller's copied)
// are independent element-wise jobs of a few microseconds each; a 1-D grid holds the blocks of both, pads first.
struct PrologueArgs {
    float *lA, *lB, *x, *x0g;
    const float* noise;
    unsigned* status;
    int64_t rs, xrs, T;
    uint64_t seed;
    int pad, rows, xpad, xrows, dl_rj, pgx, pgy, npad, ngx, XR, gauss;
};
__global__ void iaf_prologue_kernel(const PrologueArgs A) {
    const int bid = blockIdx.x;
    if (bid < A.npad) {
        const int bx = bid % A.pgx, by = (bid / A.pgx) % A.pgy, bz = bid / (A.pgx * A.pgy);
        zero_pads_body(A.lA, A.lB, A.rs, A.pad, A.rows, A.x, A.xrs, A.xpad, A.xrows, A.status, A.dl_rj, bx, by, bz, A.pgy);
    } else {
        const int nb = bid - A.npad, bx = nb % A.ngx, b = nb / A.ngx;
        if (A.noise) iaf_copy_noise_body(A.noise, A.x, A.T, A.XR, bx, b);
        else iaf_noise_body(A.x0g, A.x, A.T, A.XR, A.seed, A.gauss, bx, b);
    }
}

// ---------------- start conv (parallel_wavenet.py:222-225; masked.py:39-52) ----------------
// l[c,t] = b[c] + W0[c] x[t-3] + W1[c] x[t-2] + W2[c] x[t-1]   (shift_right folded into the taps)
__global__ void iaf_start_kernel(const float* __restrict__ x, const float* __restrict__ wb,
                                 float* __restrict__ l, int64_t T, int XR, int64_t RS) {
    const int b = blockIdx.y;
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (t >= T) return;
    const float* xp = x + (size_t)b * XR + IAF_XP + t;
    float xv[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) xv[i] = xp[i - 3];
    float* lp = l + (size_t)b * IAF_W * RS + IAF_LP + t;
    for (int c = 0; c < IAF_W; ++c) {
        const float w0 = wb[c], w1 = wb[IAF_W + c], w2 = wb[2 * IAF_W + c], bb = wb[3 * IAF_W + c];
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bb + w0 * xv[e] + w1 * xv[e + 1] + w2 * xv[e + 2];
        *reinterpret_cast<f4*>(lp + (size_t)c * RS) = o;
    }
}

// The same into the "Q4" rows of the hoisted fp32 form: l[b][group g = c / 4][column][4 channels] -- one 16-byte word is
// the MFMA B operand of a lane for a whole K-group, so a layer issues 12 operand loads per block instead of 48 (the vector
// memory instructions of the planar form cost the K loop 16 % and the launch's first 4 000 cycles: r06 cycle stamps).
__global__ void iaf_start_q4_kernel(const float* __restrict__ x, const float* __restrict__ wb,
                                    float* __restrict__ l, int64_t T, int XR, int64_t RS) {
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float* xp = x + (size_t)b * XR + IAF_XP + t;
    const float x0 = xp[-3], x1 = xp[-2], x2 = xp[-1];
    f4* lp = reinterpret_cast<f4*>(l + (size_t)b * IAF_W * RS) + IAF_LP + t;
    for (int g = 0; g < IAF_W / 4; ++g) {
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * g + e;
            o[e] = wb[3 * IAF_W + c] + wb[c] * x0 + wb[IAF_W + c] * x1 + wb[2 * IAF_W + c] * x2;
        }
        lp[(size_t)g * RS] = o;
    }
}

// ---------------- fused residual layer ----------------
// Buffer-addressed loads: base pointer + size live in a scalar resource descriptor, the
// per-lane part of the address is ONE VGPR per tap that stays constant for a tile, and the
// row (channel) offset is a scalar soffset -- no per-load 64-bit VALU address arithmetic.
__device__ inline float buf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// NOTE: pass the value as a plain float.  hipcc (ROCm 7.2) miscompiles
// raw_buffer_store_b32(__builtin_bit_cast(unsigned, vec[r]), ...) on an MFMA accumulator
// vector: every r stores element 0 (seen in the ISA as four stores of a0).
__device__ inline void buf_st(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

// 16-byte accesses of the hoisted fp32 form's residual stream ("Q4" rows: four channels of one sample per word).  The store
// goes through the same guard as every wide residual-stream store of the library (wn_mfma_h.h: buf_st4): gfx950 loses lanes
// of a buffer_store_dwordx4 whose data a VALU instruction overwrites right behind it.
__device__ inline f4 buf_ld16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
template <int AUX = 0>
__device__ inline void buf_st16(f4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u4_t w = __builtin_bit_cast(u4_t, v);
    __builtin_amdgcn_raw_buffer_store_b128(w, r, voff, soff, AUX);
    asm volatile("s_nop 1" ::"v"(w));
}

// One-shot staging of a packed weight image into LDS: ALL loads of a thread are issued
// before the first LDS store (a load->store loop serialises ~30 L2 round trips per thread,
// which cost more than the layer's MFMA time).
template <int NFLOATS>
__device__ inline void stage_weights(const float* __restrict__ wpack, float* lds) {
    constexpr int NF4 = NFLOATS / 4, NCHUNK = NF4 / 256, REM = NF4 - NCHUNK * 256;
    const f4* src = reinterpret_cast<const f4*>(wpack) + threadIdx.x;
    f4* dst = reinterpret_cast<f4*>(lds) + threadIdx.x;
    f4 tmp[NCHUNK + 1];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) tmp[k] = src[k * 256];
    if (REM && (int)threadIdx.x < REM) tmp[NCHUNK] = src[NCHUNK * 256];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) dst[k * 256] = tmp[k];
    if (REM && (int)threadIdx.x < REM) dst[NCHUNK * 256] = tmp[NCHUNK];
    __syncthreads();
}

// two pieces of a pack (NA floats from srcA, NB from srcB) back to back into LDS, all loads before the first store
template <int NA, int NB, int NT = 256, bool SYNC = true>
__device__ inline void stage_weights_part(const float* __restrict__ srcA, const float* __restrict__ srcB, float* lds) {
    constexpr int NF4 = (NA + NB) / 4, NA4 = NA / 4, NCHUNK = (NF4 + NT - 1) / NT;
    static_assert(NA % 4 == 0 && NB % 4 == 0, "16-byte pieces");
    f4 tmp[NCHUNK];
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        const int i = k * NT + (int)threadIdx.x;
        if (i < NF4) tmp[k] = i < NA4 ? reinterpret_cast<const f4*>(srcA)[i] : reinterpret_cast<const f4*>(srcB)[i - NA4];
    }
#pragma unroll
    for (int k = 0; k < NCHUNK; ++k) {
        const int i = k * NT + (int)threadIdx.x;
        if (i < NF4) reinterpret_cast<f4*>(lds)[i] = tmp[k];
    }
    if (SYNC) __syncthreads();
}

// LDS-DMA the compiler does not see (16 B per lane: lane i's bytes land at lds_byte_addr + 16 i), issued from inline asm:
// through the builtin every later LDS access of the wave would be preceded by s_waitcnt vmcnt(0) (wn_iaf_g.hip has the
// same helper and the measurement).  Completion is the caller's business: s_waitcnt vmcnt(0) before the barrier.
__device__ inline void f_dma16(const float* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}
// `floats` (a multiple of 4) from src to LDS byte address dst, 1 KB per instruction, spread over the workgroup's waves
__device__ inline void f_dma_image(const float* src, unsigned dst, int floats, int wave, int lane, int nwaves) {
    const int full = floats >> 8, rem = (floats & 255) >> 2;
    for (int i = wave; i < full; i += nwaves) f_dma16(src + (size_t)(i * 64 + lane) * 4, dst + i * 1024);
    if (rem && wave == full % nwaves && lane < rem) f_dma16(src + (size_t)(full * 64 + lane) * 4, dst + full * 1024);
}

struct TileSrc {                      // where one wave finds the B operands of one tile
    __amdgpu_buffer_rsrc_t rl, re;    // residual stream rows / upsampled-mel rows of the batch element
    int vo[3];                        // per-lane byte offset for taps t-2d, t-d, t
    int ve;                           // per-lane byte offset into enc
};

// HOIST (round 6): the conditioning 1x1 of every layer and head is evaluated by ONE fp32 GEMM per deconv stack
// (gemm_f32_kernel, wn_iaf_f.hip) and arrives as the term C in the accumulator layout, bias included -- the kernel then
// walks the 12 K-groups of the dilated conv only (K = 192 instead of 448: 53 % of a layer's MACs leave the 60 per-layer
// launches for a GEMM that keeps the matrix pipe busy), its weight image shrinks to 57 KB and it no longer reads enc.
constexpr int IAF_LAYER_F_FLOATS = 12 * 1024 + IAF_PR_FLOATS + 64 + 64;   // hoisted form: dilated-conv K-groups | PR | bgate | bres
constexpr int IAF_HEAD_F_FLOATS = 4 * 1024 + 64 * 3 + 4;   // hoisted form: out1 K-groups | bias, wmean, wscale, bmean / bscale
__device__ inline float softplus_tf(float p);

// LAST (hoisted form): the layer closes a flow and the flow head (parallel_wavenet.py:256-277, :319-324) runs in its epilogue.
// The head is column-local -- out1 over relu(l), relu, two 64-wide dots, softplus, x <- x s + mean, mean_tot / scale_tot --
// and the layer's output words ARE its MFMA B operand as they stand in the registers (channel 16 cg + 4 q + jj of lane group q:
// the K order of the packs): the flow's last l is neither written nor read back and the flow loses a launch (16.8 us each).
struct HeadF {
    const float* whead;      // head pack (out1 K-groups at 0, tail at IAF_PH_FLOATS)
    const float* Ch;         // hoisted term of the head's row block (batch row 0)
    float* x;
    float* Mt;
    float* St;
    int XR;
    int64_t T;
    int first;
};

// Hoisted form: EIGHT waves per workgroup, two per SIMD, one weight image.  The two waves of a SIMD (w and w + 4) take the
// workgroup's tiles alternately: with one wave per SIMD the gate (transcendentals), the dependent start of the residual
// 1x1 and the stores of every tile sat bare between two K loops; now one wave's epilogue runs under the other's MFMAs.
// One utterance is 1 200 tiles on 256 workgroups = 4.69 per SIMD: five rounds either way (3 + 2 per wave pair).
#ifndef WN_F32_NH
#define WN_F32_NH 2
#endif
constexpr int iaf_layer_threads(bool hoist) { return hoist ? 256 * WN_F32_NH : 256; }
// Cache policy of the hoisted form's residual-stream stores: 16 = sc1, written through.  A launch writes 19.7 MB per
// utterance that the NEXT launch reads from every XCD: left dirty in the L2s they are flushed at the kernel boundary, with
// the matrix pipes idle (timing-only ablation: a launch without its stores is 2.65 us shorter); written through as they are
// produced the launch is 1.9 us shorter (25.96 -> 24.03 us under rocprofv3; 3.21 -> 3.07 ms per call, A/B on one box).  The
// same policy on the f16x3 group kernels (WN_G_ST_AUX) costs 4.5 %: their stores are a burst at the end of a long launch.
#ifndef WN_F32_ST_AUX
#define WN_F32_ST_AUX 16
#endif

// START (hoisted form): the layer opens a flow (dilation 1) and computes its own input instead of loading it -- l0 =
// start_conv(shift_right(x)) (parallel_wavenet.py:222-225): l0[c][t'] = b[c] + w0[c] x[t'-3] + w1[c] x[t'-2] + w2[c] x[t'-1],
// 0 left of the utterance -- from five x values per column and the lane's 16 x 4 coefficients held in registers: the flow's first
// l is neither written (iaf_start_q4_kernel, a launch per flow) nor read back (12 of the block's 16 operand loads).
struct StartF {
    const float* x;          // flow input rows [B][XR], IAF_XP zero columns in front
    const float* w;          // start conv: w0[64] | w1[64] | w2[64] | b[64]
    int XR;
};

template <bool HOIST, bool LAST = false, bool START = false>
__global__ __launch_bounds__(iaf_layer_threads(HOIST), 1) void iaf_layer_kernel(
    const float* __restrict__ lin, float* __restrict__ lout, const float* __restrict__ enc,
    const float* __restrict__ wpack, int64_t RS, int64_t TE, int d, int tiles_per_row, int ntiles,
    const float* __restrict__ C, int64_t c_bstride, const HeadF hd, const StartF sf) {
    static_assert(HOIST || !LAST, "the head runs in the epilogue of the hoisted form only");
    static_assert((HOIST && !LAST) || !START, "the start conv runs in front of a plain hoisted layer only");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NG = HOIST ? 12 : 28;                 // K-groups of four K-steps
    constexpr int P_FLOATS = NG * 1024;
    constexpr int NH = HOIST ? WN_F32_NH : 1;           // wave sets that walk the tiles alternately
    // wave-uniform values through readfirstlane: a tile index the compiler takes for divergent makes every buffer
    // descriptor "divergent" and wraps each operand load in a waterfall loop (28 -> 37 us per launch when that happened)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3),
              half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const int n = lane & 15, q = lane >> 4;
    const f4* Pl = reinterpret_cast<const f4*>(lds) + lane;
    const f4* PRl = Pl + P_FLOATS / 4;
    const float* bg = lds + P_FLOATS + IAF_PR_FLOATS + q * 16;
    const float* br = bg + 64;
    // fused form: planar rows [64][RS] (4 bytes per column); hoisted form: Q4 rows [16 groups][RS][4] (16 bytes per column)
    constexpr int CB = HOIST ? 16 : 4;                   // bytes per column of a row
    const int RS4 = (int)RS * CB, TE4 = (int)TE * 4;
    const int lane_l = ((HOIST ? q : 4 * q) * (int)RS + wave * 16 + n + IAF_LP) * CB;
    const int lane_e = (4 * q * (int)TE + wave * 16 + n) * 4;
    const int tile0 = (int)blockIdx.x + half * (int)gridDim.x, tstep = NH * (int)gridDim.x;

    auto tile_src = [&](int tile) -> TileSrc {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * 64;
        TileSrc s;
        s.rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lin + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        s.re = __builtin_amdgcn_make_buffer_rsrc((void*)(enc + (size_t)b * IAF_CD * TE), 0, 0x7ffffff0, 0x00020000);
        s.vo[0] = lane_l + (tt - 2 * d) * CB;
        s.vo[1] = lane_l + (tt - d) * CB;
        s.vo[2] = lane_l + tt * CB;
        s.ve = lane_e + tt * 4;
        return s;
    };
    // hoisted term of this wave's column block of a tile: [row block][t / 16][mb][lane][4], one 16-byte load per mb
    auto load_c = [&](int tile, f4 (&cv)[4], const float* Cb = nullptr) {
        const int b = tile / tiles_per_row;
        const int cb = (tile - b * tiles_per_row) * 4 + wave;
        const f4* cp = reinterpret_cast<const f4*>((Cb ? Cb : C) + (size_t)b * c_bstride) + (size_t)cb * 256 + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) cv[mb] = __builtin_nontemporal_load(cp + mb * 64);
    };
    // 28 K-groups of 4 K-steps (16 MFMAs each): groups 0-11 = the three causal taps t-2d,
    // t-d, t of the dilated conv (4 groups of 16 channels each), groups 12-27 = the
    // conditioning 1x1 over the 256 upsampled-mel channels (not in the hoisted form).
    auto loadB = [&](const TileSrc& s, int g) -> f4 {
        f4 v;
        if (HOIST) return buf_ld16(s.rl, s.vo[g >> 2], 4 * (g & 3) * RS4);     // group row 4 cg + q: one word = the K-group's operand
        if (g < 12) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = buf_ld(s.rl, s.vo[g >> 2], (16 * (g & 3) + jj) * RS4);
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = buf_ld(s.re, s.ve, (16 * (g - 12) + jj) * TE4);
        }
        return v;
    };
    // START: the lane's coefficients (channels 16 cg + 4 q + jj) and the operand of K-group g (tap t - 2 + g / 4) from
    // xv[j] = x[t - 5 + j]; the expression is iaf_start_q4_kernel's, term for term
    f4 sw0[4], sw1[4], sw2[4], sbs[4];
    if (START) {
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int c = 16 * cg + 4 * q + jj;
                sw0[cg][jj] = sf.w[c];
                sw1[cg][jj] = sf.w[IAF_W + c];
                sw2[cg][jj] = sf.w[2 * IAF_W + c];
                sbs[cg][jj] = sf.w[3 * IAF_W + c];
            }
    }
    struct XV { float v[5]; int t; };
    auto load_x = [&](int tile) -> XV {
        const int b = tile / tiles_per_row;
        const int t = (tile - b * tiles_per_row) * 64 + wave * 16 + n;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(sf.x + (size_t)b * sf.XR), 0, sf.XR * 4, 0x00020000);
        XV r;
        r.t = t;
#pragma unroll
        for (int j = 0; j < 5; ++j) r.v[j] = buf_ld(rx, (IAF_XP + t - 5 + j) * 4, 0);
        return r;
    };
    auto start_op = [&](const XV& xv, int g) -> f4 {
        const int k = g >> 2, cg = g & 3;
        f4 o;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            o[jj] = sbs[cg][jj] + sw0[cg][jj] * xv.v[k] + sw1[cg][jj] * xv.v[k + 1] + sw2[cg][jj] * xv.v[k + 2];
        return xv.t - 2 + k >= 0 ? o : (f4){0.f, 0.f, 0.f, 0.f};
    };
    // The f32 MFMA is slow (32 cycles per instruction per SIMD) and this kernel runs one wave
    // per SIMD, so memory latency is hidden by distance, not occupancy: as soon as K-group g of
    // tile i has been multiplied, the B operands of group g of the wave's NEXT tile are loaded
    // into the same registers -- every load is issued one whole tile (~15k cycles) before its
    // use, with a single 112-register operand set and no copies.  Weights (A) are read from
    // LDS one K-group ahead into the other half of a register double buffer.
    f4 bcur[NG], ccur[4], hcur[4];
    if (HOIST) {
        // the weight image by LDS-DMA, requested FIRST (no registers, no wait): the image of the hoisted form skips the 16
        // conditioning K-groups of the pack; LAST: the head's image behind it.  It lands while the first tile's operands load.
        const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        constexpr int NWV = 4 * NH;
        f_dma_image(wpack, lds_base, 12 * 1024, wv, lane, NWV);
        f_dma_image(wpack + IAF_P_FLOATS, lds_base + 12 * 1024 * 4, IAF_PR_FLOATS + 128, wv, lane, NWV);
        if (LAST) {
            f_dma_image(hd.whead, lds_base + IAF_LAYER_F_FLOATS * 4, 4 * 1024, wv, lane, NWV);
            f_dma_image(hd.whead + IAF_PH_FLOATS, lds_base + (IAF_LAYER_F_FLOATS + 4 * 1024) * 4, 64 * 3 + 4, wv, lane, NWV);
        }
    }
    XV xn{};
    if (tile0 < ntiles) {
        const TileSrc s0 = tile_src(tile0);
        if (START) {
            xn = load_x(tile0);
#pragma unroll
            for (int g = 0; g < NG; ++g) bcur[g] = start_op(xn, g);
        } else {
#pragma unroll
            for (int g = 0; g < NG; ++g) bcur[g] = loadB(s0, g);
        }
        if (HOIST) load_c(tile0, ccur);
        if (LAST) load_c(tile0, hcur, hd.Ch);
    }
    if (HOIST) {
        // the DMA requests are older than the tile loads and vmcnt retires in order: waiting for everything is the wait
        // for the tile's operands, which the first K-group needs anyway
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        // the weight image AFTER the first tile's operand loads are in flight (the two latencies overlap)
        stage_weights<IAF_LAYER_FLOATS>(wpack, lds);
    }
    for (int tile = tile0; tile < ntiles; tile += tstep) {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * 64;
        const int next = tile + tstep;
        const bool has_next = next < ntiles;
        const TileSrc sn = tile_src(has_next ? next : tile);

        f4 acc[4], cur[4], a[2][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = HOIST ? ccur[mb] : *reinterpret_cast<const f4*>(bg + mb * 4);
        if (HOIST && has_next) load_c(next, ccur);
        if (START) xn = load_x(has_next ? next : tile);
        f4 hacc[4];
        if (LAST) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) hacc[mb] = hcur[mb];
            if (has_next) load_c(next, hcur, hd.Ch);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[0][mb] = Pl[mb * 64];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) a[(g + 1) & 1][mb] = Pl[((g + 1) * 4 + mb) * 64];
            }
            if (g >= 8 && g < 12) cur[g - 8] = bcur[g];      // tap t is also the residual C-in
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma4(a[g & 1][mb][jj], bcur[g][jj], acc[mb]);
            // pin the order inside the group: next group's 4 LDS weight reads FIRST (so their
            // latency hides under this group's 16 MFMAs), then the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            // START: the next tile's operands from its x values, four groups behind the loads issued at the top of this tile
            // (their round trip) -- group g - 4 here, the last four behind the loop
            if (START) { if (g >= 4) bcur[g - 4] = start_op(xn, g - 4); }
            else if (has_next) bcur[g] = loadB(sn, g);
            // keep the schedule group-by-group: without this fence the scheduler hoists every
            // LDS weight read of the unrolled loop to the top and spills hundreds of registers
            __builtin_amdgcn_sched_barrier(0);
        }
        if (START) {
#pragma unroll
            for (int g = NG - 4; g < NG; ++g) bcur[g] = start_op(xn, g);
        }
        // gate: sigmoid(first half) * tanh(second half)  (parallel_wavenet.py:246-250)
        f4 gt[2];
#pragma unroll
        for (int mg = 0; mg < 2; ++mg)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gt[mg][r] = sigmoidf_(acc[mg][r]) * tanhf_(acc[mg + 2][r]);
        // residual 1x1 accumulated onto l: the tap-t operands (groups 8-11) are the C-in
        f4 d2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) d2[mb] = cur[mb] + *reinterpret_cast<const f4*>(br + mb * 4);
#pragma unroll
        for (int j4 = 0; j4 < 2; ++j4) {
            f4 ar[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) ar[mb] = PRl[(j4 * 4 + mb) * 64];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) d2[mb] = mfma4(ar[mb][jj], gt[j4][jj], d2[mb]);
        }
        if (LAST) {
            // flow head on the registers: out1 over relu(l') on top of the head's hoisted tile, then the two projections
            const f4* PHl = reinterpret_cast<const f4*>(lds + IAF_LAYER_F_FLOATS) + lane;
            const float* ht = lds + IAF_LAYER_F_FLOATS + 4 * 1024;
            const float* wm = ht + 64 + q * 16;
            const float* wsc = wm + 64;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) {
                f4 ah[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) ah[mb] = PHl[(cg * 4 + mb) * 64];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float bv = fmaxf(d2[cg][jj], 0.f);                           // relu(l), :256
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) hacc[mb] = mfma4(ah[mb][jj], bv, hacc[mb]);
                }
            }
            float pm = 0.f, ps = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = fmaxf(hacc[mb][r], 0.f);
                    pm = fmaf(wm[mb * 4 + r], o, pm);
                    ps = fmaf(wsc[mb * 4 + r], o, ps);
                }
            pm += __shfl_xor(pm, 16);
            ps += __shfl_xor(ps, 16);
            pm += __shfl_xor(pm, 32);
            ps += __shfl_xor(ps, 32);
            if (q == 0) {
                const int64_t t = tt + wave * 16 + n;
                const float mean = pm + ht[192];
                const float sc = fminf(fmaxf(softplus_tf(ps + ht[193]), EXP_M9), EXP_7);   // :105-114
                float* xp = hd.x + (size_t)b * hd.XR + IAF_XP + t;
                *xp = *xp * sc + mean;                                                  // :277
                float* mp = hd.Mt + (size_t)b * hd.T + t;
                float* sp = hd.St + (size_t)b * hd.T + t;
                if (hd.first) { *mp = mean; *sp = sc; }
                else { *mp = mean + *mp * sc; *sp = *sp * sc; }                         // :322-323
            }
            continue;
        }
        const __amdgpu_buffer_rsrc_t ro =
            __builtin_amdgcn_make_buffer_rsrc((void*)(lout + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        const int vo_out = lane_l + tt * CB;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            if (HOIST) {                                   // channels 16 mb + 4 q + (0..3) = one word of group row 4 mb + q
                buf_st16<WN_F32_ST_AUX>(d2[mb], ro, vo_out, 4 * mb * RS4);
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) buf_st(d2[mb][r], ro, vo_out, (16 * mb + r) * RS4);
        }
    }
}

// ---------------- flow head (parallel_wavenet.py:256-277, :319-324) ----------------
__device__ inline float softplus_tf(float p) {
    // tf.nn.softplus (Eigen): threshold = log(eps) + 2
    const float thr = -13.942384719848633f;
    if (p > -thr) return p;
    if (p < thr) return expf(p);
    return log1pf(expf(p));
}

template <bool HOIST>
__global__ __launch_bounds__(256, 1) void iaf_head_kernel(
    const float* __restrict__ lin, const float* __restrict__ enc, const float* __restrict__ wpack,
    float* __restrict__ x, float* __restrict__ Mt, float* __restrict__ St,
    int64_t RS, int64_t TE, int XR, int64_t T, int first, int tiles_per_row, int ntiles,
    const float* __restrict__ C, int64_t c_bstride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NG = HOIST ? 4 : 20;
    constexpr int PH_FLOATS = NG * 1024;
    if (HOIST) stage_weights_part<4 * 1024, 64 * 3 + 4>(wpack, wpack + IAF_PH_FLOATS, lds);
    else stage_weights<IAF_HEAD_FLOATS>(wpack, lds);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const f4* Pl = reinterpret_cast<const f4*>(lds) + lane;
    const float* bo = lds + PH_FLOATS + q * 16;
    const float* wm = bo + 64;
    const float* wsc = wm + 64;
    const float bmean = lds[PH_FLOATS + 192], bscale = lds[PH_FLOATS + 193];
    constexpr int CB = HOIST ? 16 : 4;                   // bytes per column of an l row (Q4 rows in the hoisted form)
    const int RS4 = (int)RS * CB, TE4 = (int)TE * 4;
    const int lane_l = ((HOIST ? q : 4 * q) * (int)RS + wave * 16 + n + IAF_LP) * CB;
    const int lane_e = (4 * q * (int)TE + wave * 16 + n) * 4;

    // 20 K-groups: 0-3 = out1 over relu(l), 4-19 = mel_cond_out1 over the 256 enc channels (hoisted form: those arrive
    // in C, bias included); same one-tile-ahead operand prefetch as iaf_layer_kernel.
    auto tile_src = [&](int tile) -> TileSrc {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * 64;
        TileSrc s;
        s.rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lin + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        s.re = __builtin_amdgcn_make_buffer_rsrc((void*)(enc + (size_t)b * IAF_CD * TE), 0, 0x7ffffff0, 0x00020000);
        s.vo[0] = s.vo[1] = s.vo[2] = lane_l + tt * CB;
        s.ve = lane_e + tt * 4;
        return s;
    };
    auto load_c = [&](int tile, f4 (&cv)[4]) {
        const int b = tile / tiles_per_row;
        const int cb = (tile - b * tiles_per_row) * 4 + wave;
        const f4* cp = reinterpret_cast<const f4*>(C + (size_t)b * c_bstride) + (size_t)cb * 256 + lane;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) cv[mb] = __builtin_nontemporal_load(cp + mb * 64);
    };
    auto loadB = [&](const TileSrc& s, int g) -> f4 {
        f4 v;
        if (HOIST) return buf_ld16(s.rl, s.vo[2], 4 * g * RS4);
        if (g < 4) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = buf_ld(s.rl, s.vo[2], (16 * g + jj) * RS4);
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) v[jj] = buf_ld(s.re, s.ve, (16 * (g - 4) + jj) * TE4);
        }
        return v;
    };
    f4 bcur[NG], ccur[4];
    if ((int)blockIdx.x < ntiles) {
        const TileSrc s0 = tile_src(blockIdx.x);
#pragma unroll
        for (int g = 0; g < NG; ++g) bcur[g] = loadB(s0, g);
        if (HOIST) load_c(blockIdx.x, ccur);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_row;
        const int t0 = (tile - b * tiles_per_row) * 64 + wave * 16;
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        const TileSrc sn = tile_src(has_next ? next : tile);
        f4 acc[4], a[2][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = HOIST ? ccur[mb] : *reinterpret_cast<const f4*>(bo + mb * 4);
        if (HOIST && has_next) load_c(next, ccur);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) a[0][mb] = Pl[mb * 64];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) a[(g + 1) & 1][mb] = Pl[((g + 1) * 4 + mb) * 64];
            }
            f4 bv = bcur[g];
            if (g < 4) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bv[jj] = fmaxf(bv[jj], 0.f);        // relu(l), :256
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma4(a[g & 1][mb][jj], bv[jj], acc[mb]);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            if (has_next) bcur[g] = loadB(sn, g);
            __builtin_amdgcn_sched_barrier(0);
        }
        float pm = 0.f, ps = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float o = fmaxf(acc[mb][r], 0.f);
                pm = fmaf(wm[mb * 4 + r], o, pm);
                ps = fmaf(wsc[mb * 4 + r], o, ps);
            }
        pm += __shfl_xor(pm, 16);
        ps += __shfl_xor(ps, 16);
        pm += __shfl_xor(pm, 32);
        ps += __shfl_xor(ps, 32);
        if (q == 0) {
            const int64_t t = t0 + n;
            const float mean = pm + bmean;
            const float s = fminf(fmaxf(softplus_tf(ps + bscale), EXP_M9), EXP_7);   // :105-114
            float* xp = x + (size_t)b * XR + IAF_XP + t;
            *xp = *xp * s + mean;                                                    // :277
            float* mp = Mt + (size_t)b * T + t;
            float* sp = St + (size_t)b * T + t;
            if (first) { *mp = mean; *sp = s; }
            else { *mp = mean + *mp * s; *sp = *sp * s; }                            // :322-323
        }
    }
}

// ---------------- final affine + _clip_quant_scale (parallel_wavenet.py:326-330,347-359) ----------------
__device__ inline void clip_quant_one(float y, int Q, int mu, float& wav, int& qi) {
    qi = wn_clip_quantize(y, Q);
    wav = wn_dequant(qi, Q, mu);
}

__global__ void iaf_final_kernel(const float* __restrict__ x0, const float* __restrict__ Mt,
                                 const float* __restrict__ St, int64_t n, int Q, int mu,
                                 float* __restrict__ wav, int* __restrict__ idx, float* __restrict__ xraw,
                                 float* __restrict__ mean_tot, float* __restrict__ scale_tot,
                                 unsigned* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // A split-fp16 operand left the fp16 range somewhere in this call (wn_codec.h): whatever the flows computed after
    // that is not the reference's result.  Never hand it out as audio -- every float output becomes NaN, the index 0;
    // wn_iaf_range_status() reports WN_ERANGE and the caller re-runs the call with wn_iaf_generate_prec(.., 1 = fp32).
    const bool bad = *status != 0u;
    // status[1] accumulates over the calls made on this workspace (zero_pads_kernel clears status[0] only): a caller
    // that keeps a run of calls asynchronous asks ONCE at the end (wn_iaf_range_status_since_reset)
    if (bad && i == 0) atomicOr(status + 1, *status);
    float s = fminf(St[i], EXP_7);                          // :327
    float m = Mt[i];
    float y = x0[i] * s + m;                                // :330
    float w; int qi;
    clip_quant_one(y, Q, mu, w, qi);
    if (bad) { w = y = m = s = __builtin_nanf(""); qi = 0; }
    wav[i] = w;
    if (idx) idx[i] = qi;
    if (xraw) xraw[i] = y;
    if (mean_tot) mean_tot[i] = m;
    if (scale_tot) scale_tot[i] = s;
}

__global__ void clip_quant_kernel(const float* __restrict__ x, int64_t n, int Q, int mu,
                                  float* __restrict__ wav, int* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w; int qi;
    clip_quant_one(x[i], Q, mu, w, qi);
    if (wav) wav[i] = w;
    if (idx) idx[i] = qi;
}

struct IafLayout {
    int64_t T, TE, RS;
    int XR, c0;
    size_t status, enc, lA, lB, x, x0, M, S, C, scratch, total;   // byte offsets
    int64_t c_bstride;                                    // floats of hoisted conditioning per batch row
    int form;                                             // WN_COND_FUSED / WN_COND_HOISTED for this call
};

IafLayout iaf_layout(const wn_handle* h, int B, int F, int form) {
    IafLayout L;
    L.T = wn_iaf_length(h, F);
    L.TE = (int64_t)F * h->frame_shift;
    L.c0 = (int)((L.TE - L.T) / 2);                        // wavenet.py:76-85
    L.RS = IAF_LP + L.T;
    L.XR = (int)(IAF_XP + L.T);
    size_t o = 0;
    auto carve = [&](size_t floats) { size_t r = o; o += align_up(floats * sizeof(float), 256); return r; };
    L.form = wn_iaf_form(h, B, L.T, form);
    L.status = carve(64);                                  // range-guard word: first bytes of the workspace
    if (h->generic_student) {                              // fp32 rows of the model's own widths (wn_iaf_x.hip)
        const size_t W = h->cfg.width, Cd = h->cfg.deconv_width;
        L.form = WN_COND_FUSED;
        L.enc = carve((size_t)B * Cd * L.TE + 64);
        L.lA = carve((size_t)B * W * L.RS);
        L.lB = carve((size_t)B * W * L.RS);
        L.x = carve((size_t)2 * B * L.XR);
        L.x0 = carve((size_t)B * L.T);
        L.M = carve((size_t)B * L.T);
        L.S = carve((size_t)B * L.T);
        L.c_bstride = 0;
        L.C = o;
        L.scratch = o;
        o += wn_deconv_scratch_bytes(h, B, F);
        L.total = o;
        return L;
    }
    L.enc = carve((size_t)B * IAF_CD * L.TE + 64);
    L.lA = carve((size_t)B * IAF_W * L.RS);
    L.lB = carve((size_t)B * IAF_W * L.RS);
    L.x = carve((size_t)2 * B * L.XR);      // flow input and, for flows that are ONE layer group, a second copy (wn_iaf_g.hip)
    L.x0 = carve((size_t)B * L.T);
    L.M = carve((size_t)B * L.T);
    L.S = carve((size_t)B * L.T);
    const bool hoist = L.form == WN_COND_HOISTED;
    L.c_bstride = hoist ? (int64_t)wn_iaf_c_floats(h->cfg.share_deconv ? h->cond_rows : 0, L.T) : 0;
    if (hoist && !h->cfg.share_deconv) {        // private deconv stacks: one flow's rows at a time
        int mx = 0;
        for (const IafFlowPack& fp : h->flows) mx = std::max(mx, (int)fp.layers.size() + 1);
        L.c_bstride = (int64_t)wn_iaf_c_floats(mx, L.T);
    }
    L.C = carve((size_t)B * L.c_bstride);
    L.scratch = o;
    o += wn_deconv_scratch_bytes(h, B, F);
    L.total = o;
    return L;
}

}  // namespace

// ---------------------------------------------------------------------------
int wn_pack_iaf(wn_handle* h, std::vector<float>& blob) {
    const wn_config& c = h->cfg;
    auto var = [&](const std::string& nme) -> const std::vector<float>& { return h->vars.at(nme).data; };
    for (int k = 0; k < c.n_flows; ++k) {
        const std::string p = "iaf_" + std::to_string(k + 1);
        IafFlowPack fp;
        fp.deconv_stack = c.share_deconv ? 0 : k;
        // start conv: W [1,3,1,64] -> w[tap][c], then bias
        blob.resize(align_up(blob.size(), 64));
        fp.start_off = blob.size();
        {
            std::vector<float> W = wn_get_kernel(h, p + "/start_conv", "W", false);
            blob.insert(blob.end(), W.begin(), W.end());
            const auto& b = var(p + "/start_conv/biases");
            blob.insert(blob.end(), b.begin(), b.end());
        }
        for (int i = 0; i < c.iaf_layers[k]; ++i) {
            const std::string s = std::to_string(i + 1);
            std::vector<float> Wd = wn_get_kernel(h, p + "/dilated_conv_" + s, "W", false);   // [3][64][64]
            std::vector<float> Wc = wn_get_kernel(h, p + "/mel_cond_" + s, "W", false);       // [256][64]
            std::vector<float> Wr = wn_get_kernel(h, p + "/res_" + s, "W", false);            // [32][64]
            const auto& bd = var(p + "/dilated_conv_" + s + "/biases");
            const auto& bc = var(p + "/mel_cond_" + s + "/biases");
            const auto& br = var(p + "/res_" + s + "/biases");
            IafLayerPack lp;
            lp.dilation = 1 << (i % c.num_stages);          // parallel_wavenet.py:228
            blob.resize(align_up(blob.size(), 64));
            lp.off = blob.size();
            blob.resize(blob.size() + IAF_LAYER_FLOATS);
            float* P = blob.data() + lp.off;
            float* PR = P + IAF_P_FLOATS;
            float* bg = PR + IAF_PR_FLOATS;
            float* brp = bg + 64;
            for (int ks = 0; ks < 112; ++ks) {
                const int seg = ks / 16, j = ks % 16;
                for (int mb = 0; mb < 4; ++mb)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i16 = lane & 15, q = lane >> 4;
                        const int o = 16 * mb + i16;
                        const int kch = 16 * (j >> 2) + 4 * q + (j & 3);
                        const float v = seg < 3 ? Wd[((size_t)seg * 64 + kch) * 64 + o]
                                                : Wc[((size_t)(seg - 3) * 64 + kch) * 64 + o];
                        P[((((ks >> 2) * 4 + mb) * 64) + lane) * 4 + (ks & 3)] = v;
                    }
            }
            for (int j = 0; j < 8; ++j)
                for (int mb = 0; mb < 4; ++mb)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i16 = lane & 15, q = lane >> 4;
                        const int o = 16 * mb + i16;
                        const int cch = 16 * (j >> 2) + 4 * q + (j & 3);
                        PR[((((j >> 2) * 4 + mb) * 64) + lane) * 4 + (j & 3)] = Wr[(size_t)cch * 64 + o];
                    }
            for (int q = 0; q < 4; ++q)
                for (int mb = 0; mb < 4; ++mb)
                    for (int r = 0; r < 4; ++r) {
                        const int o = 16 * mb + 4 * q + r;
                        bg[q * 16 + mb * 4 + r] = bd[o] + bc[o];
                        brp[q * 16 + mb * 4 + r] = br[o];
                    }
            fp.layers.push_back(lp);
        }
        // head
        {
            std::vector<float> Wo = wn_get_kernel(h, p + "/out1", "W", false);            // [64][64]
            std::vector<float> Wco = wn_get_kernel(h, p + "/mel_cond_out1", "W", false);  // [256][64]
            std::vector<float> Wm = wn_get_kernel(h, p + "/out2_mean", "W", false);       // [64][1]
            std::vector<float> Ws = wn_get_kernel(h, p + "/out2_scale", "W", false);
            const auto& bo = var(p + "/out1/biases");
            const auto& bco = var(p + "/mel_cond_out1/biases");
            blob.resize(align_up(blob.size(), 64));
            fp.head_off = blob.size();
            blob.resize(blob.size() + IAF_HEAD_FLOATS);
            float* P = blob.data() + fp.head_off;
            float* tb = P + IAF_PH_FLOATS;
            for (int ks = 0; ks < 80; ++ks) {
                const int seg = ks / 16, j = ks % 16;
                for (int mb = 0; mb < 4; ++mb)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i16 = lane & 15, q = lane >> 4;
                        const int o = 16 * mb + i16;
                        const int kch = 16 * (j >> 2) + 4 * q + (j & 3);
                        const float v = seg == 0 ? Wo[(size_t)kch * 64 + o]
                                                 : Wco[((size_t)(seg - 1) * 64 + kch) * 64 + o];
                        P[((((ks >> 2) * 4 + mb) * 64) + lane) * 4 + (ks & 3)] = v;
                    }
            }
            for (int q = 0; q < 4; ++q)
                for (int mb = 0; mb < 4; ++mb)
                    for (int r = 0; r < 4; ++r) {
                        const int o = 16 * mb + 4 * q + r;
                        tb[q * 16 + mb * 4 + r] = bo[o] + bco[o];
                        tb[64 + q * 16 + mb * 4 + r] = Wm[o];
                        tb[128 + q * 16 + mb * 4 + r] = Ws[o];
                    }
            tb[192] = var(p + "/out2_mean/biases")[0];
            tb[193] = var(p + "/out2_scale/biases")[0];
            tb[194] = tb[195] = 0.f;
        }
        h->flows.push_back(fp);
    }
    // row-block table of the fp32 conditioning GEMM (wn_iaf_f.hip): the 16 cond K-groups of each pack (layer: K-groups
    // 12-27, head: 4-19) are contiguous 16 x 1024 floats; the biases (dilated + cond / out1 + cond) sit in lane order
    // behind the fragments
    {
        std::vector<unsigned> tab;
        for (const IafFlowPack& fp : h->flows) {
            for (const IafLayerPack& lp : fp.layers) {
                tab.push_back((unsigned)(lp.off + 12 * 1024));
                tab.push_back((unsigned)(lp.off + IAF_P_FLOATS + IAF_PR_FLOATS));
            }
            tab.push_back((unsigned)(fp.head_off + 4 * 1024));
            tab.push_back((unsigned)(fp.head_off + IAF_PH_FLOATS));
        }
        if (blob.size() >= 0x1ff00000u) return wn_fail(h, WN_EINVAL, "weight blob too large");
        blob.resize(align_up(blob.size(), 64));
        h->cond_tab_f_off = blob.size();
        blob.resize(blob.size() + align_up(tab.size(), 4));
        memcpy(blob.data() + h->cond_tab_f_off, tab.data(), tab.size() * sizeof(unsigned));
    }
    return WN_OK;
}

// A student whose widths the MFMA kernels are not specialised for: the same call on the generic fp32 kernels of
// wn_iaf_x.hip (fp32 upsampler GEMM, one launch per start conv / layer / head, fp32 rows, no fp16 anywhere).
static int iaf_generate_generic(wn_handle* h, const IafLayout& L, const float* mel, int B, int F, const float* noise,
                                uint64_t seed, float* wav, int32_t* idx, float* x_raw, float* mean_tot, float* scale_tot,
                                float* rand_out, char* base, hipStream_t st) {
    const wn_config& c = h->cfg;
    float* enc = reinterpret_cast<float*>(base + L.enc);
    float* lA = reinterpret_cast<float*>(base + L.lA);
    float* lB = reinterpret_cast<float*>(base + L.lB);
    float* x = reinterpret_cast<float*>(base + L.x);
    float* x0g = reinterpret_cast<float*>(base + L.x0);
    float* Mt = reinterpret_cast<float*>(base + L.M);
    float* St = reinterpret_cast<float*>(base + L.S);
    unsigned* status = reinterpret_cast<unsigned*>(base + L.status);
    void* scratch = base + L.scratch;
    {
        const int rows = B * c.width;
        dim3 g((IAF_LP + 255) / 256, std::min(rows, 32768), 3);
        hipLaunchKernelGGL(zero_pads_kernel, g, dim3(256), 0, st, lA, lB, L.RS, IAF_LP, rows, x, (int64_t)L.XR, IAF_XP, 2 * B,
                           status, 0);
    }
    const float* x0 = noise;
    {
        dim3 g((unsigned)((L.T / 4 + 255) / 256), B);
        if (noise) {
            hipLaunchKernelGGL(iaf_copy_noise_kernel, g, dim3(256), 0, st, noise, x, L.T, L.XR);
        } else {
            hipLaunchKernelGGL(iaf_noise_kernel, g, dim3(256), 0, st, x0g, x, L.T, L.XR, seed,
                               c.loss_type == WN_LOSS_GAUSS ? 1 : 0);
            x0 = x0g;
        }
    }
    if (c.share_deconv)
        if (int rc = wn_run_deconv(h, 0, mel, B, F, enc, L.TE, scratch, st, false, nullptr, WN_PREC_F32)) return rc;
    for (int k = 0; k < c.n_flows; ++k) {
        const IafFlowX& fx = h->flows_x[k];
        if (!c.share_deconv)
            if (int rc = wn_run_deconv(h, fx.deconv_stack, mel, B, F, enc, L.TE, scratch, st, false, nullptr, WN_PREC_F32)) return rc;
        wn_iaf_x_start(h, fx, x, lA, L.T, L.XR, L.RS, B, st);
        float* lin = lA;
        float* lout = lB;
        for (const IafLayerX& lx : fx.layers) {
            wn_iaf_x_layer(h, lx, lin, lout, enc, L.RS, L.TE, L.c0, B, L.T, st);
            std::swap(lin, lout);
        }
        wn_iaf_x_head(h, fx, lin, enc, x, Mt, St, L.RS, L.TE, L.c0, L.XR, L.T, k == 0 ? 1 : 0, B, st);
    }
    const int64_t nn = (int64_t)B * L.T;
    const int Q = c.use_mu_law ? 256 : 65536;
    hipLaunchKernelGGL(iaf_final_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, x0, Mt, St, nn, Q, c.use_mu_law,
                       wav, idx, x_raw, mean_tot, scale_tot, status);
    if (rand_out && rand_out != x0) WN_HIP(h, hipMemcpyAsync(rand_out, x0, nn * sizeof(float), hipMemcpyDeviceToDevice, st));
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

extern "C" int wn_iaf_generate(wn_handle* h, const float* mel, int B, int F, const float* noise,
                               uint64_t seed, float* wav, int32_t* idx, float* x_raw, float* mean_tot,
                               float* scale_tot, float* rand_out, void* ws, size_t ws_bytes, void* stream) {
    return wn_iaf_generate_form(h, WN_FORM_DEFAULT, mel, B, F, noise, seed, wav, idx, x_raw, mean_tot, scale_tot, rand_out,
                                ws, ws_bytes, stream);
}

extern "C" int wn_iaf_generate_form(wn_handle* h, int form, const float* mel, int B, int F, const float* noise,
                                    uint64_t seed, float* wav, int32_t* idx, float* x_raw, float* mean_tot,
                                    float* scale_tot, float* rand_out, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_iaf_generate: null handle");
    if (form < WN_FORM_DEFAULT || form > WN_FORM_F16X3_FUSED)
        return wn_fail(h, WN_EINVAL, "wn_iaf_generate_form: unknown form %d", form);
    if (!h->finalized) return wn_fail(h, WN_ESTATE, "wn_iaf_generate: call wn_finalize first");
    if (h->cfg.kind != WN_KIND_STUDENT)
        return wn_fail(h, WN_EINVAL, "wn_iaf_generate: handle is not a ParallelWavenet student");
    if (B < 1 || F < 1) return wn_fail(h, WN_EINVAL, "wn_iaf_generate: B and F must be >= 1");
    // inside the library from here on: the switches (wn_iaf_set_groups, wn_profile_*) are refused until the call returns,
    // and each of them is read ONCE, here
    const WnWork work(h);
    const bool prof_on = h->prof_on, parts_on = h->parts_on;
    const int pmask = parts_on ? h->parts_mask : 0xf;   // a restriction exists only inside a parts session (wn_profile_parts_only)
    const IafLayout L = iaf_layout(h, B, F, form);
    if (L.T == 0) return WN_OK;   // fewer frames than one max-dilation block: empty output
    if (!mel || !wav || !ws) return wn_fail(h, WN_EINVAL, "wn_iaf_generate: null mel/wav/workspace");
    if (ws_bytes < L.total)
        return wn_fail(h, WN_ENOMEM, "wn_iaf_generate: workspace %zu < %zu bytes", ws_bytes, L.total);
    if ((int64_t)B * L.T / 64 > 0x7fffffff) return wn_fail(h, WN_EINVAL, "wn_iaf_generate: batch too large");
    if (L.TE > 2000000)   // 32-bit buffer offsets inside one batch element (960 * TE bytes)
        return wn_fail(h, WN_EINVAL, "wn_iaf_generate: %lld samples per utterance exceeds the 2,000,000 "
                       "(125 s) limit of the 32-bit row offsets; split the utterance", (long long)L.TE);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(ws);
    float* enc = reinterpret_cast<float*>(base + L.enc);
    float* lA = reinterpret_cast<float*>(base + L.lA);
    float* lB = reinterpret_cast<float*>(base + L.lB);
    float* x = reinterpret_cast<float*>(base + L.x);
    float* const xbuf0 = x;
    float* x0g = reinterpret_cast<float*>(base + L.x0);
    float* Mt = reinterpret_cast<float*>(base + L.M);
    float* St = reinterpret_cast<float*>(base + L.S);
    float* Cc = reinterpret_cast<float*>(base + L.C);
    unsigned* status = reinterpret_cast<unsigned*>(base + L.status);
    void* scratch = base + L.scratch;
    const wn_config& c = h->cfg;

    if (h->generic_student)
        return iaf_generate_generic(h, L, mel, B, F, noise, seed, wav, idx, x_raw, mean_tot, scale_tot, rand_out, base, st);
    const int prec = wn_form_precision(h, form);
    const bool f16x3 = prec == WN_PREC_F16X3;

    const bool use_groups = f16x3 && wn_iaf_use_groups(h, B, L.T, L.form);
    // measurement aid (wn_profile_parts_begin): an event where a part of the call begins.  Parts: 0 prologue / epilogue
    // (pads, noise, final), 1 upsampler, 2 conditioning GEMM, 3 residual stack (start convs, layers, heads)
    int part_now = -2;
    constexpr size_t MAX_PART_EVENTS = 4096;            // a power loop of thousands of calls stops recording, not allocating
    auto record = [&](std::vector<hipEvent_t>& list, int tag) -> int {
        hipEvent_t ev;
        WN_HIP(h, hipEventCreate(&ev));
        {
            std::lock_guard<std::mutex> g(h->list_mu);
            list.push_back(ev);
            if (&list == &h->part_events) h->part_tags.push_back(tag);
        }
        WN_HIP(h, hipEventRecord(ev, st));
        return WN_OK;
    };
    auto part = [&](int tag) -> int {
        if (!parts_on || tag == part_now) return WN_OK;
        part_now = tag;
        {
            std::lock_guard<std::mutex> g(h->list_mu);
            if (h->part_events.size() >= MAX_PART_EVENTS) return WN_OK;
        }
        return record(h->part_events, tag);
    };
    if (int rc = part(0)) return rc;
    // prologue: zero left pads + the flow input (drawn on the device, or the caller's noise), one launch
    const float* x0 = noise ? noise : x0g;
    {
        PrologueArgs A{};
        // G4 layout (f16x3) and the Q4 rows of the hoisted fp32 form: 16 interleaved group rows per batch element, each
        // 4*(LP+T) words; fused fp32 form: 64 planar rows
        const bool rows16 = f16x3 || L.form == WN_COND_HOISTED;
        A.rows = rows16 ? B * 16 : B * IAF_W;
        A.rs = rows16 ? 4 * L.RS : L.RS;
        A.pad = rows16 ? 4 * IAF_LP : IAF_LP;
        A.lA = lA; A.lB = lB; A.x = x; A.x0g = x0g; A.noise = noise; A.status = status;
        A.xrs = (int64_t)L.XR; A.xpad = IAF_XP; A.xrows = 2 * B;
        A.dl_rj = use_groups ? 64 + (int)(L.T / 32) : 0;
        A.pgx = (A.pad + 255) / 256;
        A.pgy = std::min(std::max(A.rows, 2 * B), 32768);
        A.npad = A.pgx * A.pgy * 3;
        A.T = L.T; A.XR = L.XR; A.seed = seed; A.gauss = c.loss_type == WN_LOSS_GAUSS ? 1 : 0;
        A.ngx = (int)((L.T / 4 + 255) / 256);
        hipLaunchKernelGGL(iaf_prologue_kernel, dim3((unsigned)(A.npad + A.ngx * B)), dim3(256), 0, st, A);
    }
    const bool hoist = L.form == WN_COND_HOISTED;
    auto blob_u = [&](size_t off) { return reinterpret_cast<const unsigned*>(h->d_blob + off); };
    const unsigned* cond_tab = blob_u(h->cond_tab_off);
    const size_t rb_floats = (size_t)(L.T / 16) * 1024;     // C floats of one row block
    if (c.share_deconv) {
        if (int rc = part(1)) return rc;
        int rc = (pmask & 2) ? wn_run_deconv(h, 0, mel, B, F, enc, L.TE, scratch, st, f16x3, status, prec) : WN_OK;
        if (rc) return rc;
        if (hoist) if (int rc2 = part(2)) return rc2;
        if (hoist && (pmask & 4)) {
            if (f16x3)
                wn_iaf_c_cond(enc, h->d_blob, cond_tab, blob_u(use_groups ? h->order_all_off : h->order_id_off),
                              use_groups ? h->n_nat_all : h->cond_rows, Cc, L.c_bstride, L.TE, L.c0, h->cond_rows, B, L.T,
                              h->num_cu, st);
            else
                wn_iaf_f_cond(h, enc, blob_u(h->cond_tab_f_off), h->cond_rows, Cc, L.c_bstride, L.TE, L.c0, B, L.T, st);
        }
    }
    const int tiles_per_row = (int)(L.T / 64);
    const int ntiles = B * tiles_per_row;
    const int grid = ntiles < h->num_cu ? ntiles : h->num_cu;
    const float* encc = enc + L.c0;
    for (int k = 0; k < c.n_flows; ++k) {
        const IafFlowPack& fp = h->flows[k];
        if (!c.share_deconv) {
            if (int rc = part(1)) return rc;
            int rc = (pmask & 2) ? wn_run_deconv(h, fp.deconv_stack, mel, B, F, enc, L.TE, scratch, st, f16x3, status, prec) : WN_OK;
            if (rc) return rc;
            if (hoist) if (int rc2 = part(2)) return rc2;
            if (hoist && (pmask & 4)) {
                if (f16x3)
                    wn_iaf_c_cond(enc, h->d_blob, cond_tab + fp.rb_base,
                                  use_groups ? blob_u(h->order_flow_off) + fp.rb_base : blob_u(h->order_id_off),
                                  use_groups ? h->n_nat_flow[k] : (int)fp.layers.size() + 1, Cc, L.c_bstride, L.TE, L.c0,
                                  (int)fp.layers.size() + 1, B, L.T, h->num_cu, st);
                else
                    wn_iaf_f_cond(h, enc, blob_u(h->cond_tab_f_off) + 2 * fp.rb_base, (int)fp.layers.size() + 1, Cc,
                                  L.c_bstride, L.TE, L.c0, B, L.T, st);
            }
        }
        if (int rc = part(3)) return rc;
        if (!(pmask & 8)) continue;
        // row blocks of this flow inside C (all flows when the deconv stack is shared)
        const float* Cf = Cc + (c.share_deconv ? (size_t)fp.rb_base * rb_floats : 0);
        if (use_groups) {
            // natural / decimated layer groups in LDS: lA natural, lB DL; the first group computes the start conv
            // from x, the last one runs the flow head
            float* gin = lA;
            float* gout = lB;
            if (prof_on) if (int rc = record(h->prof_events, 0)) return rc;
            for (size_t gi = 0; gi < fp.groups.size(); ++gi) {
                const WnGroup& g = fp.groups[gi];
                const bool lastg = gi + 1 == fp.groups.size();
                // a flow that is ONE group reads x (start conv, with its halo) and writes x (head) in the same launch:
                // the new x goes to the other copy, which becomes the flow input from then on
                float* xnew = (gi == 0 && lastg) ? (x == xbuf0 ? xbuf0 + (size_t)B * L.XR : xbuf0) : x;
                wn_iaf_g_run(h, g, fp.layers.data(), Cf + (size_t)g.begin * rb_floats, rb_floats, L.c_bstride, gin, gout,
                             L.RS, lastg ? 0 : fp.groups[gi + 1].kind, B, L.T, gi == 0 ? x : nullptr, L.XR,
                             h->d_blob + fp.start_off, lastg, h->d_blob + fp.head_off_h, x, xnew, Mt, St, k == 0 ? 1 : 0,
                             status, st);
                x = xnew;
                std::swap(gin, gout);
            }
            if (prof_on) {
                if (int rc = record(h->prof_events, 0)) return rc;
                std::lock_guard<std::mutex> g(h->list_mu);
                h->prof_launches += (int64_t)fp.groups.size();
            }
            continue;
        }
        // fused f16x3 form: the start conv runs inside the first layer kernel of the flow (dilation 1)
        const bool fuse_start = f16x3 && !hoist && !fp.layers.empty() && fp.layers[0].dilation == 1;
        // hoisted form: layer pairs with small dilations run as one launch (wn_iaf_c_pair); when the flow
        // starts with such a pair the start conv runs inside it as well
        const bool pair_start = hoist && f16x3 && fp.layers.size() >= 2 && fp.layers[0].dilation == 1 &&
                                wn_iaf_c_pair_ok(fp.layers[0].dilation, fp.layers[1].dilation);
        // fp32 hoisted form: the start conv runs in front of the first layer (dilation 1, at least two layers: the last one carries the head)
        static const bool no_start_fuse = getenv("WN_F32_NO_STARTFUSE") != nullptr;
        const bool start_in_layer = hoist && !f16x3 && !no_start_fuse && fp.layers.size() >= 2 && fp.layers[0].dilation == 1;
        if (fuse_start || pair_start || start_in_layer) {
        } else if (f16x3) {
            wn_iaf_h_start(x, h->d_blob + fp.start_off, lA, L.T, L.XR, L.RS, B, st, status);
        } else if (hoist) {
            dim3 g((unsigned)((L.T + 255) / 256), B);
            hipLaunchKernelGGL(iaf_start_q4_kernel, g, dim3(256), 0, st, x, h->d_blob + fp.start_off, lA, L.T,
                               L.XR, L.RS);
        } else {
            dim3 g((unsigned)((L.T / 4 + 255) / 256), B);
            hipLaunchKernelGGL(iaf_start_kernel, g, dim3(256), 0, st, x, h->d_blob + fp.start_off, lA, L.T,
                               L.XR, L.RS);
        }
        float* lin = lA;
        float* lout = lB;
        // bench.py measurement aid: event pairs around every maximal run of single-layer launches
        // (the dominant kernel); two-layer launches and heads stay outside the brackets
        bool prof_open = false;
        auto prof_mark = [&](bool open, int launches) -> int {
            if (!prof_on) return WN_OK;
            if (open != prof_open) {
                if (int rc = record(h->prof_events, 0)) return rc;
                prof_open = open;
            }
            if (open) {
                std::lock_guard<std::mutex> g(h->list_mu);
                h->prof_launches += launches;
            }
            return WN_OK;
        };
        size_t li = 0;
        bool head_done = false;
        for (size_t i = 0; i < fp.layers.size(); ++i) {
            const IafLayerPack& lp = fp.layers[i];
            if (hoist && f16x3 && i + 1 < fp.layers.size() && wn_iaf_c_pair_ok(lp.dilation, fp.layers[i + 1].dilation)) {
                const IafLayerPack& lq = fp.layers[i + 1];
                if (int rc = prof_mark(false, 0)) return rc;
                wn_iaf_c_pair(lin, lout, Cf + li * rb_floats, Cf + (li + 1) * rb_floats, L.c_bstride,
                              h->d_blob + lp.off_h, h->d_blob + lq.off_h, L.RS, lp.dilation, lq.dilation, B, L.T,
                              h->num_cu, st, (i == 0 && pair_start) ? x : nullptr, L.XR, h->d_blob + fp.start_off, status);
                float* t = lin; lin = lout; lout = t;
                li += 2;
                ++i;
                continue;
            }
            if (hoist && f16x3 && i + 1 == fp.layers.size() && wn_iaf_c_last_ok()) {
                // last layer of the flow: the head runs in its epilogue
                if (int rc = prof_mark(false, 0)) return rc;
                wn_iaf_c_layer_head(lin, Cf + li * rb_floats, Cf + (li + 1) * rb_floats, L.c_bstride,
                                    h->d_blob + lp.off_h, h->d_blob + fp.head_off_h, x, Mt, St, L.RS, L.XR, lp.dilation,
                                    k == 0 ? 1 : 0, B, L.T, h->num_cu, st, status);
                head_done = true;
                ++li;
                continue;
            }
            if (int rc = prof_mark(true, 1)) return rc;
            if (hoist && f16x3)
                wn_iaf_c_layer(lin, lout, Cf + li * rb_floats, L.c_bstride, h->d_blob + lp.off_h, L.RS, lp.dilation, B,
                               L.T, h->num_cu, st, status);
            else if (hoist && i + 1 == fp.layers.size()) {
                // fp32 hoisted form, last layer of the flow: the head runs in its epilogue (iaf_layer_kernel<true, true>)
                if (int rc = prof_mark(false, 0)) return rc;
                const HeadF hd{h->d_blob + fp.head_off, Cf + (li + 1) * rb_floats, x, Mt, St, L.XR, L.T, k == 0 ? 1 : 0};
                hipLaunchKernelGGL((iaf_layer_kernel<true, true>), dim3(grid), dim3(iaf_layer_threads(true)),
                                   (IAF_LAYER_F_FLOATS + IAF_HEAD_F_FLOATS) * sizeof(float), st, lin, lout, encc,
                                   h->d_blob + lp.off, L.RS, L.TE, lp.dilation, tiles_per_row, ntiles, Cf + li * rb_floats,
                                   L.c_bstride, hd, StartF{});
                head_done = true;
                ++li;
                continue;
            } else if (hoist && i == 0 && start_in_layer)
                // fp32 hoisted form, first layer of the flow: the start conv runs in front of it (iaf_layer_kernel<true, false, true>)
                hipLaunchKernelGGL((iaf_layer_kernel<true, false, true>), dim3(grid), dim3(iaf_layer_threads(true)), IAF_LAYER_F_FLOATS * sizeof(float), st,
                                   lin, lout, encc, h->d_blob + lp.off, L.RS, L.TE, lp.dilation, tiles_per_row, ntiles,
                                   Cf + li * rb_floats, L.c_bstride, HeadF{}, StartF{x, h->d_blob + fp.start_off, L.XR});
            else if (hoist)
                hipLaunchKernelGGL((iaf_layer_kernel<true, false>), dim3(grid), dim3(iaf_layer_threads(true)), IAF_LAYER_F_FLOATS * sizeof(float), st,
                                   lin, lout, encc, h->d_blob + lp.off, L.RS, L.TE, lp.dilation, tiles_per_row, ntiles,
                                   Cf + li * rb_floats, L.c_bstride, HeadF{}, StartF{});
            else if (f16x3)
                wn_iaf_h_layer(lin, lout, enc, h->d_blob + lp.off_h, L.RS, L.TE, L.c0, lp.dilation, B, L.T, h->num_cu, st, status,
                               (li == 0 && fuse_start) ? x : nullptr, L.XR, h->d_blob + fp.start_off);
            else
                hipLaunchKernelGGL((iaf_layer_kernel<false, false>), dim3(grid), dim3(256), IAF_LAYER_FLOATS * sizeof(float), st,
                                   lin, lout, encc, h->d_blob + lp.off, L.RS, L.TE, lp.dilation, tiles_per_row,
                                   ntiles, (const float*)nullptr, (int64_t)0, HeadF{}, StartF{});
            float* t = lin; lin = lout; lout = t;
            ++li;
        }
        if (int rc = prof_mark(false, 0)) return rc;
        if (head_done) {
        } else if (hoist && !f16x3)
            hipLaunchKernelGGL(iaf_head_kernel<true>, dim3(grid), dim3(256), IAF_HEAD_F_FLOATS * sizeof(float), st, lin,
                               encc, h->d_blob + fp.head_off, x, Mt, St, L.RS, L.TE, L.XR, L.T, k == 0 ? 1 : 0,
                               tiles_per_row, ntiles, Cf + li * rb_floats, L.c_bstride);
        else if (hoist)
            wn_iaf_c_head(lin, Cf + li * rb_floats, L.c_bstride, h->d_blob + fp.head_off_h, x, Mt, St, L.RS, L.XR, L.T,
                          k == 0 ? 1 : 0, B, h->num_cu, st);
        else if (f16x3)
            wn_iaf_h_head(lin, enc, h->d_blob + fp.head_off_h, x, Mt, St, L.RS, L.TE, L.c0, L.XR, L.T, k == 0 ? 1 : 0, B,
                          h->num_cu, st);
        else
            hipLaunchKernelGGL(iaf_head_kernel<false>, dim3(grid), dim3(256), IAF_HEAD_FLOATS * sizeof(float), st, lin,
                               encc, h->d_blob + fp.head_off, x, Mt, St, L.RS, L.TE, L.XR, L.T, k == 0 ? 1 : 0,
                               tiles_per_row, ntiles, (const float*)nullptr, (int64_t)0);
    }
    if (int rc = part(0)) return rc;
    {
        const int64_t nn = (int64_t)B * L.T;
        const int Q = c.use_mu_law ? 256 : 65536;
        hipLaunchKernelGGL(iaf_final_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, x0, Mt, St, nn,
                           Q, c.use_mu_law, wav, idx, x_raw, mean_tot, scale_tot, status);
        if (rand_out && rand_out != x0)
            WN_HIP(h, hipMemcpyAsync(rand_out, x0, nn * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (int rc = part(-1)) return rc;
    if (parts_on) {
        std::lock_guard<std::mutex> g(h->list_mu);
        ++h->part_calls;
    }
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

extern "C" int wn_clip_quant(wn_handle* h, const float* x, int64_t n, float* wav, int32_t* idx, void* stream) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_clip_quant: null handle");
    if (n < 0 || (n > 0 && !x)) return wn_fail(h, WN_EINVAL, "wn_clip_quant: bad argument");
    if (n == 0) return WN_OK;
    const int Q = h->cfg.use_mu_law ? 256 : 65536;
    hipLaunchKernelGGL(clip_quant_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, n, Q, h->cfg.use_mu_law, wav, idx);
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

// Where the per-layer conditioning 1x1s run (f16x3 only).  Hoisted (default): one GEMM per deconv
// stack writes every layer's projection, the layer kernels stream 768 B/sample (+ 256 B written by
// the GEMM) and the small-dilation layers run two per launch (wn_iaf_c_pair).  Fused
// (cond_mode / WN_COND=fused): every layer kernel streams enc itself (1536 B/sample/layer, no extra
// workspace).  Measured on MI355X (M samples/s, hoisted / fused): 50.7 / 46.8 at one utterance of
// 4.8 s, 56 / 49 at two, 61 / 42 at eight.
int wn_form_precision(const wn_handle* h, int form) {
    return form == WN_FORM_F32 ? WN_PREC_F32 : form == WN_FORM_DEFAULT ? h->cfg.precision : WN_PREC_F16X3;
}

int wn_iaf_set_attrs(wn_handle* h) {
    if (h->generic_student) return wn_iaf_x_set_attrs(h);
    int rc = wn_iaf_h_set_attrs(h);
    if (rc) return rc;
    rc = wn_iaf_c_set_attrs(h);
    if (rc) return rc;
    rc = wn_iaf_g_set_attrs(h);
    if (rc) return rc;
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_LAYER_FLOATS * sizeof(float)));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_head_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_HEAD_FLOATS * sizeof(float)));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_LAYER_F_FLOATS * sizeof(float)));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_kernel<true, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_LAYER_F_FLOATS * sizeof(float)));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_kernel<true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (IAF_LAYER_F_FLOATS + IAF_HEAD_F_FLOATS) * sizeof(float)));
    return WN_OK;
}

int wn_iaf_form(const wn_handle* h, int B, int64_t T, int form) {
    if (h->generic_student) return WN_COND_FUSED;               // fp32 kernels of wn_iaf_x.hip: no projected term
    if (form == WN_FORM_F16X3_FUSED) return WN_COND_FUSED;
    // fp32 form (round 6): the same placement policy; its conditioning GEMM walks 128-column tiles
    if (wn_form_precision(h, form) != WN_PREC_F16X3 && T % 128 != 0) return WN_COND_FUSED;
    int mode = h->cfg.cond_mode;
    if (mode == WN_COND_AUTO) mode = h->cond_env_mode;           // WN_COND, resolved once in wn_create
    if (mode == WN_COND_FUSED) return WN_COND_FUSED;
    if (mode == WN_COND_HOISTED) return WN_COND_HOISTED;
    return wn_iaf_hoisted(h, B, T) ? WN_COND_HOISTED : WN_COND_FUSED;
}

// Default placement (cond_mode 0, no WN_COND): hoisted while the projected term (256 B per sample and row
// block) stays below a third of the device memory, else the fused form, which needs no such workspace.
bool wn_iaf_hoisted(const wn_handle* h, int B, int64_t T) {
    int rows = h->cond_rows;
    if (!h->cfg.share_deconv) {
        rows = 0;
        for (const IafFlowPack& fp : h->flows) rows = std::max(rows, (int)fp.layers.size() + 1);
    }
    return (double)B * (double)T * 256.0 * rows <= h->hoist_limit_bytes;
}

// The layer-group kernel (wn_iaf_g.hip) runs the hoisted form whenever every flow has a group plan and the decimated
// view exists (T a multiple of 32 * 16); lA then keeps the natural layout and lB the DL layout for the whole call.
// Where it pays (round 5, profiles/r05_batch_sweep.txt; configs[1] utterances on one box, ms per call groups | per-layer
// launches): 1.19 | 1.43 at one utterance, 4.71 | 4.74 at four, 6.99 | 6.92 at six, 9.30 | 9.29 at eight, 13.91 | 14.27 at twelve,
// 18.35 | 18.78 at sixteen.  From two utterances on BOTH forms run at the package power cap, so what decides
// is energy per sample: the group form moves the residual stream through the fabric twice per ten layers instead of eight
// times, the per-layer form has no halo recompute; the groups are 2.3-2.5 % ahead from twelve utterances and within +-1 % of
// the per-layer form below (second sweep, same file: the per-layer form 0.6-1.5 % ahead at five to seven utterances, the
// groups 0.5 % ahead at eight, 3.5 % at ten).  Default: groups, except between 4.5 and 7 segments per CU.  wn_iaf_set_groups
// (or WN_GROUPS=1 / WN_NO_GROUPS=1 at wn_create) forces either form (A/B measurements, cross-form tests).
bool wn_iaf_use_groups(const wn_handle* h, int B, int64_t T, int form) {
    if (form != WN_COND_HOISTED || !h->groups_ok || T % 512 != 0) return false;
    if (h->groups_env) return h->groups_env > 0;                // wn_iaf_set_groups; WN_NO_GROUPS / WN_GROUPS set its initial value in wn_create
    // five to seven utterances' worth of segments: the per-layer launches are 0.6-1.5 % ahead there (r05_batch_sweep.txt)
    const int64_t segs = (int64_t)B * ((T / 16 + 19) / 20);
    return segs <= 4 * (int64_t)h->num_cu + h->num_cu / 2 || segs > 7 * (int64_t)h->num_cu;
}

extern "C" int wn_iaf_set_groups(wn_handle* h, int mode) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_iaf_set_groups: null handle");
    if (mode < -1 || mode > 1) return wn_fail(h, WN_EINVAL, "wn_iaf_set_groups: mode must be -1 (never), 0 (as created) or 1 (always), got %d", mode);
    WN_SWITCH(h, "wn_iaf_set_groups");
    h->groups_env = mode ? mode : h->groups_env0;      // 0: what wn_create resolved (the size policy, or WN_GROUPS / WN_NO_GROUPS)
    return WN_OK;
}

extern "C" int wn_iaf_layer_groups(const wn_handle* h, int B, int F) {
    if (!h || h->cfg.kind != WN_KIND_STUDENT || B < 1 || F < 1 || h->cfg.precision != WN_PREC_F16X3) return 0;
    const int64_t T = wn_iaf_length(h, F);
    return wn_iaf_use_groups(h, B, T, wn_iaf_form(h, B, T, WN_FORM_DEFAULT)) ? 1 : 0;
}

extern "C" int wn_iaf_cond_hoisted(const wn_handle* h, int B, int F) {
    if (!h || h->cfg.kind != WN_KIND_STUDENT || B < 1 || F < 1) return 0;
    return wn_iaf_form(h, B, wn_iaf_length(h, F), WN_FORM_DEFAULT) == WN_COND_HOISTED ? 1 : 0;
}

// default form: room for the fp32 re-run of a call that left the fp16 range as well (it needs less: no projected term)
size_t wn_iaf_workspace_bytes(const wn_handle* h, int B, int F, int form) {
    size_t n = iaf_layout(h, B, F, form).total;
    if (form == WN_FORM_DEFAULT) n = std::max(n, iaf_layout(h, B, F, WN_FORM_F32).total);
    return n;
}

static int range_report(wn_handle* h, unsigned flag) {
    if (flag)
        return wn_fail(h, WN_ERANGE, "wn_iaf_generate: an activation left the fp16 range of the split-fp16 arithmetic "
                       "(|value| >= 65504); the outputs of that call are NaN -- re-run it with "
                       "wn_iaf_generate_form(h, WN_FORM_F32, ...)");
    return WN_OK;
}

extern "C" int wn_iaf_range_status(wn_handle* h, const void* ws, void* stream) {
    if (!h || !ws) return wn_fail(h, WN_EINVAL, "wn_iaf_range_status: null argument");
    unsigned flag = 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WN_HIP(h, hipMemcpyAsync(&flag, ws, sizeof(flag), hipMemcpyDeviceToHost, st));      // the status word leads the workspace
    WN_HIP(h, hipStreamSynchronize(st));
    return range_report(h, flag);
}

extern "C" int wn_iaf_range_reset(wn_handle* h, void* ws, void* stream) {
    if (!h || !ws) return wn_fail(h, WN_EINVAL, "wn_iaf_range_reset: null argument");
    WN_HIP(h, hipMemsetAsync(ws, 0, 64, reinterpret_cast<hipStream_t>(stream)));
    return WN_OK;
}

extern "C" int wn_iaf_range_status_since_reset(wn_handle* h, void* ws, void* stream) {
    if (!h || !ws) return wn_fail(h, WN_EINVAL, "wn_iaf_range_status_since_reset: null argument");
    unsigned flags[2] = {0, 0};
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    WN_HIP(h, hipMemcpyAsync(flags, ws, sizeof(flags), hipMemcpyDeviceToHost, st));
    WN_HIP(h, hipStreamSynchronize(st));
    if (flags[0] | flags[1]) WN_HIP(h, hipMemsetAsync(ws, 0, 64, st));
    return range_report(h, flags[0] | flags[1]);
}

extern "C" int wn_profile_begin(wn_handle* h) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_profile_begin: null handle");
    WN_SWITCH(h, "wn_profile_begin");
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    h->prof_events.clear();
    h->prof_launches = 0;
    h->prof_on = true;
    return WN_OK;
}

extern "C" int wn_profile_parts_begin(wn_handle* h) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_profile_parts_begin: null handle");
    WN_SWITCH(h, "wn_profile_parts_begin");
    for (hipEvent_t e : h->part_events) (void)hipEventDestroy(e);
    h->part_events.clear();
    h->part_tags.clear();
    h->part_calls = 0;
    h->parts_mask = 0xf;
    h->parts_on = true;
    return WN_OK;
}

extern "C" int wn_profile_parts_only(wn_handle* h, int mask) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_profile_parts_only: null handle");
    if (mask < 1 || mask > 0xf) return wn_fail(h, WN_EINVAL, "wn_profile_parts_only: mask must be in 1..15, got %d", mask);
    WN_SWITCH(h, "wn_profile_parts_only");
    if (!h->parts_on)      // a restricted call skips work: the switch lives and dies with a measurement session
        return wn_fail(h, WN_ESTATE, "wn_profile_parts_only: only between wn_profile_parts_begin and wn_profile_parts_end");
    h->parts_mask = mask;
    return WN_OK;
}

extern "C" int wn_profile_parts_end(wn_handle* h, double* part_ms, int64_t* calls) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_profile_parts_end: null handle");
    WN_SWITCH(h, "wn_profile_parts_end");
    h->parts_on = false;
    h->parts_mask = 0xf;                                           // never left armed
    double ms[WN_PROFILE_PARTS] = {0.0};
    hipError_t bad = hipSuccess;
    for (size_t i = 0; i + 1 < h->part_events.size() && bad == hipSuccess; ++i) {
        const int tag = h->part_tags[i];
        if (tag < 0 || tag >= WN_PROFILE_PARTS) continue;          // -1: between two calls
        float f = 0.f;
        bad = hipEventSynchronize(h->part_events[i + 1]);
        if (bad == hipSuccess) bad = hipEventElapsedTime(&f, h->part_events[i], h->part_events[i + 1]);
        ms[tag] += f;
    }
    for (hipEvent_t e : h->part_events) (void)hipEventDestroy(e);   // (on the error path as well)
    h->part_events.clear();
    h->part_tags.clear();
    if (bad != hipSuccess) return wn_fail(h, WN_EIO, "wn_profile_parts_end: %s", hipGetErrorString(bad));
    if (part_ms) for (int i = 0; i < WN_PROFILE_PARTS; ++i) part_ms[i] = ms[i];
    if (calls) *calls = h->part_calls;
    return WN_OK;
}

extern "C" int wn_profile_pause(wn_handle* h, int paused) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_profile_pause: null handle");
    WN_SWITCH(h, "wn_profile_pause");
    h->prof_on = !paused;
    return WN_OK;
}

extern "C" int wn_profile_end(wn_handle* h, double* layer_ms, int64_t* layer_launches) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_profile_end: null handle");
    WN_SWITCH(h, "wn_profile_end");
    h->prof_on = false;
    double ms = 0.0;
    for (size_t i = 0; i + 1 < h->prof_events.size(); i += 2) {
        float f = 0.f;
        WN_HIP(h, hipEventSynchronize(h->prof_events[i + 1]));
        WN_HIP(h, hipEventElapsedTime(&f, h->prof_events[i], h->prof_events[i + 1]));
        ms += f;
    }
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    h->prof_events.clear();
    if (layer_ms) *layer_ms = ms;
    if (layer_launches) *layer_launches = h->prof_launches;
    return WN_OK;
}
