// Autoregressive "fastgen" path on gfx950:
//   wavenet/wavenet.py:379-514 (Fastgen.sample), wavenet/masked.py:328-405
//   (causal_linear / linear), wavenet/loss_func.py:140-206 (sampling heads),
//   wavenet/fastgen.py:128-169 (per-sample driver loop).
//
// The reference runs ONE python->TF->device round trip per audio sample with two
// CPU FIFOQueues per causal layer.  Here the queues are device-resident rings
// (slot = step mod 2*rate holds the layer INPUT of that step, masked.py:357-359),
// the step counter lives in device memory, and every pointer a step needs is
// derived on the device from that counter -- so one step is a static sequence of
// GEMV kernels that is captured once in a hipGraph (AR_GRAPH_STEPS steps per
// graph) and replayed; no host round trip per sample.
//
// Two step implementations, chosen by batch size:
//  * B < 4: wave-per-output-row GEMV kernels, activations [batch][feature], ONE launch per
//    layer (ar_layer_m_kernel: the residual update is substituted into the next layer's
//    current-tap term, so a layer is a single dependent phase) -- 34 launches per step;
//  * B >= 4: the batch is the N dimension of v_mfma_f32_16x16x4_f32 (activations
//    [feature][padded batch]), so one pass over the 119 MB of weights serves every utterance
//    (two launches per layer, 65 per step).
// Both are chains of dependent, launch-latency-bound kernels.
#include <cstdlib>
#include <cstring>

#include "wn_internal.h"
#include "wn_codec.h"


namespace {

constexpr int AR_BT = 4;            // batch elements per register tile
constexpr int AR_GRAPH_STEPS = 16;  // steps captured per hipGraph
constexpr int AR_HDR = 64;          // header floats (step counter lives in the first 8 bytes)

struct ArDims {
    int B, W, S, G, Cd, OW, Q, mu, loss, M;
};

// State blob (floats): [hdr 64][a_prev B][u ring 4*B][rings...][l B*W][s B*S][g B*G/2][z B*S][out B*OW][ebuf B*OW]
struct ArStateLayout {
    size_t a_prev, uring, rings, l, s, g, z, out, ebuf, encT, slab, l2, dbuf, total;
    int NB;      // 0: GEMV layout [batch][feature]; >0: MFMA layout [feature][NB] (NB = padded batch)
};

// The batched (MFMA) step keeps activations feature-major / batch-minor so that the batch is
// the N dimension of v_mfma_f32_16x16x4_f32: one pass over the weights serves every utterance.
// rows of the GEMV step past the default register tiles (AR_NCA / AR_NCH chunks of 256 floats): the wide instantiation
static bool ar_wide(const wn_config& c) {
    return 3 * c.width + c.deconv_width > 2048 || c.gate_width / 2 > 1024 || c.width > 1024;
}

int ar_padded_batch(int B, bool has_b_pack = true) {
    const char* mode = getenv("WN_AR_MODE");                  // "gemv" | "mfma" override (tests, A/B)
    if (mode && !strcmp(mode, "gemv")) return 0;
    if (B < 4 && !(mode && !strcmp(mode, "mfma"))) return 0;   // measured: the GEMV step wins below 4 utterances
    if (!has_b_pack) return 0;
    return B <= 16 ? 16 : B <= 32 ? 32 : (B + 63) / 64 * 64;
}

ArStateLayout ar_state_layout(const wn_handle* h, int B) {
    const wn_config& c = h->cfg;
    ArStateLayout L;
    L.NB = ar_padded_batch(B, h->ar.wss_b_off != 0);
    const size_t Bp = L.NB ? L.NB : B;
    size_t o = AR_HDR;
    auto carve = [&](size_t n) { size_t r = o; o += align_up(n, 64); return r; };
    L.a_prev = carve(B);
    L.uring = carve(4 * (size_t)B);
    L.rings = carve(h->ar.ring_floats * Bp);
    L.l = carve(Bp * c.width);
    L.s = carve(Bp * c.skip_width);
    L.g = carve(Bp * c.gate_width / 2);
    L.z = carve(Bp * c.skip_width);
    L.out = carve(Bp * (size_t)((c.out_width + 15) / 16 * 16));
    L.ebuf = carve((size_t)B * c.out_width);
    L.encT = carve(Bp * c.deconv_width);
    // K-split partial pre-activations of the gate GEMM: [ceil(K/256)][gate_width][NB]
    L.slab = carve(L.NB ? (size_t)((3 * c.width + c.deconv_width + c.gate_width / 2 + 255) / 256) * c.gate_width * Bp : 0);
    // merged GEMV step: second residual-stream buffer and two pre-activation buffers (ping-pong per layer)
    L.l2 = carve(Bp * c.width);
    L.dbuf = carve(L.NB ? 0 : 2 * (size_t)B * c.gate_width);
    L.total = o;
    return L;
}

// wave-wide sum, partners 32, 16, 8, 4, 2, 1; the four in-row steps are DPP moves instead of LDS permutes (a
// ds_bpermute is ~100 cycles of latency, six in a row were 0.4 us of every layer launch of the one-utterance step)
__device__ inline float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));   // row_ror:8 == xor 8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, true));   // row_ror:4: partner sums equal xor 4's
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1] == xor 2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2] == xor 1
    return v;
}

__device__ inline long long ar_step_of(const float* state) {
    return *reinterpret_cast<const long long*>(state);
}

// Per-lane slice of a GEMV row: K is walked in chunks of AR_KC f4 per lane, and inside a
// chunk EVERY load (weights, then the batch's inputs) is issued before the first use -- a
// plain `for k` loop waits one L2/Infinity-Cache round trip per iteration, which at batch 1
// was most of the step time.  Weights are held in registers across the batch loop.
constexpr int AR_KC = 4;
constexpr int AR_MAXSLAB = 12;      // K-split slabs of the batched gate GEMM (K = 3W + Cd + G/2 <= 3072)                          // f4 per lane per chunk: 8 * 256 lanes-floats = 2048 floats of K

// ---- generic row GEMV: y[b][o] (op)= bias[o] + W[o][:] . x[b][:]  (masked.py:383-405) ----
// MODE 2: out1        z  = relu(.)      x = [relu(s) | enc_t]
// MODE 3: out2        out = .           x = z
template <int MODE>
__global__ __launch_bounds__(256) void ar_rows_kernel(
    float* __restrict__ state, ArStateLayout L, ArDims D, const float* __restrict__ Wm,
    const float* __restrict__ bias, int rows, int K, const float* __restrict__ enc, int Tn, int per_step) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= rows) return;
    const long long t = ar_step_of(state);
    const long long ti = per_step ? 0 : t;
    const float* wrow = Wm + (size_t)o * K;
    const float bo = bias[o];
    auto xptr = [&](int b, int k) -> const float* {
        if (MODE == 2) return k < D.S ? state + L.s + (size_t)b * D.S + k
                                      : enc + ((size_t)b * Tn + ti) * D.Cd + (k - D.S);
        return state + L.z + (size_t)b * D.S + k;
    };
    for (int b0 = 0; b0 < D.B; b0 += AR_BT) {
        float acc[AR_BT];
#pragma unroll
        for (int e = 0; e < AR_BT; ++e) acc[e] = 0.f;
        for (int kc = 0; kc < K; kc += AR_KC * 256) {
            f4 w[AR_KC];
#pragma unroll
            for (int i = 0; i < AR_KC; ++i) {
                const int k = kc + i * 256 + lane * 4;
                w[i] = k < K ? *reinterpret_cast<const f4*>(wrow + k) : (f4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int e = 0; e < AR_BT; ++e) {
                const int b = b0 + e;
                if (b >= D.B) break;
                f4 xv[AR_KC];
#pragma unroll
                for (int i = 0; i < AR_KC; ++i) {
                    const int k = kc + i * 256 + lane * 4;
                    xv[i] = k < K ? *reinterpret_cast<const f4*>(xptr(b, k)) : (f4){0.f, 0.f, 0.f, 0.f};
                    if (MODE == 2 && k < D.S) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) xv[i][c] = fmaxf(xv[i][c], 0.f);       // wavenet.py:494
                    }
                }
#pragma unroll
                for (int i = 0; i < AR_KC; ++i)
                    acc[e] += w[i][0] * xv[i][0] + w[i][1] * xv[i][1] + w[i][2] * xv[i][2] + w[i][3] * xv[i][3];
            }
        }
#pragma unroll
        for (int e = 0; e < AR_BT; ++e) {
            const int b = b0 + e;
            if (b >= D.B) break;
            const float v = wave_sum(acc[e]) + bo;
            if (lane == 0) {
                if (MODE == 2) state[L.z + (size_t)b * D.S + o] = fmaxf(v, 0.f);   // wavenet.py:499
                else state[L.out + (size_t)b * D.OW + o] = v;
            }
        }
    }
}

// =====================  GEMV step (B < 4): one launch per layer  =====================
// A layer needs two dependent phases (pre-activation -> gate -> res/skip), and the step is a chain
// of launch-latency-bound kernels (~4.7 us each), so the count of dependent launches IS the step
// time.  With lin_j = lin_{j-1} + Wres_{j-1} m_{j-1} + bres_{j-1} substituted into the current-tap
// term of layer j,
//   d_j = Wd_j[t-2d] ring_j(t-2d) + Wd_j[t-d] ring_j(t-d) + Wd_j[t] lin_{j-1} + (Wd_j[t] Wres_{j-1}) m_{j-1}
//         + Wc_j enc + (bd_j + bc_j + Wd_j[t] bres_{j-1}),
// every row of {lin_j, s += skip_{j-1}, d_j} depends only on lin_{j-1}, d_{j-1} (m_{j-1} = gate(d_{j-1})
// is recomputed by every workgroup: gate_width/2 sigmoid*tanh) and on ring / enc data of earlier steps:
// ONE kernel per layer, 34 launches per step instead of 65.  Same arithmetic up to fp32
// re-association of the current-tap term; lin_j itself is computed exactly as before.
//   rings: slot = step mod (2d+1), so the slot pushed at step t is not the one holding t-2d.

// One GEMV row with every global load issued before the first use: the weights of the row
// (NC chunks of 256 floats, one f4 per lane each) are loaded up front -- before the prologue of
// the kernel even starts -- and so are the inputs that live in global memory; inputs produced by
// the prologue come from LDS afterwards.  The kernels are one or two memory round trips long, so
// the number of SEQUENTIAL round trips is what matters.
constexpr int AR_NCA = 8;    // chunks of the dilated + cond row: 3W + Cd <= 2048 (every shipped teacher)
constexpr int AR_NCH = 4;    // chunks of an H-long row (gate_width / 2 <= 1024)
constexpr int AR_GEMV_MAXB = 3;   // the GEMV step serves batches below 4 (ar_padded_batch)
// Wide teachers (masked.py takes any width): the same kernels with rows of up to 4096 / 2048 floats in registers and one
// utterance at a time (three utterances' inputs beside a 4096-float row would not fit the register file).  Shapes whose
// batched K = 3W + Cd + G/2 exceeds 256 * AR_MAXSLAB have no batched pack and take this instantiation at every batch size;
// the others (e.g. width 640 with a double gate: K = 2816) switch to the batched MFMA step from four utterances on, like
// the tuned shapes (tests/test_gpu_ar.py::test_wide_teachers_run_on_the_wide_instantiation runs B = 5 and 17 on them)
constexpr int AR_NCA_W = 16, AR_NCH_W = 8, AR_GEMV_MAXB_W = 1;

// sigmoid(a) * tanh(b) with the hardware exp / rcp (abs error ~1e-7, as in the IAF kernels)
__device__ inline float ar_gate(float a, float b) {
#ifdef WN_AR_LIBM_GATE
    return (1.f / (1.f + expf(-a))) * tanhf(b);
#else
    return __builtin_amdgcn_rcpf(1.f + __expf(-a)) * (1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * b) + 1.f));
#endif
}

template <int NC>
struct RowW {
    f4 w[NC];
};
template <int NC>
__device__ inline RowW<NC> row_load(const float* __restrict__ wrow, int K, int lane) {
    RowW<NC> r;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int k = i * 256 + lane * 4;
        r.w[i] = k < K ? *reinterpret_cast<const f4*>(wrow + k) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    return r;
}
template <int NC, class XF>
__device__ inline float row_fma(const RowW<NC>& r, int K, int lane, XF xload) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int k = i * 256 + lane * 4;
        if (k < K) {
            const f4 x = xload(i, k);
            acc += r.w[i][0] * x[0] + r.w[i][1] * x[1] + r.w[i][2] * x[2] + r.w[i][3] * x[3];
        }
    }
    return acc;
}

// inputs of the dilated + cond row that live in global memory: [ring(t-2d) | ring(t-d) | (lin: LDS) | enc]
template <int NCA = AR_NCA>
struct GateX {
    f4 x[NCA];
};
template <int NCA = AR_NCA>
__device__ inline GateX<NCA> gate_x_load(const float* ring2, const float* ring1, const float* enc_t, int W, int Cd, int lane) {
    GateX<NCA> g;
    const int K = 3 * W + Cd;
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
        const int k = i * 256 + lane * 4;
        const float* p = k < W ? ring2 + k : k < 2 * W ? ring1 + (k - W) : k < 3 * W ? nullptr : enc_t + (k - 3 * W);
        g.x[i] = (k < K && p) ? *reinterpret_cast<const f4*>(p) : (f4){0.f, 0.f, 0.f, 0.f};
    }
    return g;
}

// first kernel of a step: lin_0 = conv_start (every workgroup recomputes it: 3 MACs per channel),
// s = skip_start(lin_0), d_0 = dilated_conv_1 + mel_cond_1 pre-activations.  Rows: [0,S) s, [S,S+G) d.
// The batch is walked in chunks of AR_GEMV_MAXB (one chunk for the batches this step is chosen for).
template <int NCA, int NCH, int MB>
__global__ __launch_bounds__(256) void ar_first_m_kernel(
    float* __restrict__ state, ArStateLayout L, ArDims D, const float* __restrict__ wav_in,
    const float* __restrict__ forced, int Tn, const float* __restrict__ wb, const float* __restrict__ Wss,
    const float* __restrict__ bss, const float* __restrict__ Wd, const float* __restrict__ bd,
    const float* __restrict__ enc, int per_step, size_t ring_off, int dil) {
    extern __shared__ __attribute__((aligned(16))) float sh[];          // lin_0 [chunk][W]
    const long long t = ar_step_of(state);
    const long long ti = per_step ? 0 : t;
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool is_s = o < D.S, is_d = !is_s && o < D.S + D.G;
    const int r = o - D.S, K = 3 * D.W + D.Cd;
    const float* ring = state + L.rings + (size_t)D.B * ring_off;
    const size_t slot2 = (size_t)((t + 1) % (2 * dil + 1)) * D.B;          // lin_0[t-2d]
    const size_t slot1 = (size_t)((t + dil + 1) % (2 * dil + 1)) * D.B;    // lin_0[t-d]
    RowW<NCA> wd;
    RowW<NCH> ws;
    float bias = 0.f;
    if (is_d) {
        wd = row_load<NCA>(Wd + (size_t)r * K, K, lane);
        bias = bd[r];
    } else if (is_s) {
        ws = row_load<NCH>(Wss + (size_t)o * D.W, D.W, lane);
        bias = bss[o];
    }
    float* ur = state + L.uring;
    for (int b0 = 0; b0 < D.B; b0 += MB) {
        const int nb = min(MB, D.B - b0);
        GateX<NCA> gx[MB];
        if (is_d) {
#pragma unroll
            for (int e = 0; e < MB; ++e)
                if (e < nb)
                    gx[e] = gate_x_load<NCA>(ring + (slot2 + b0 + e) * D.W, ring + (slot1 + b0 + e) * D.W,
                                        enc + ((size_t)(b0 + e) * Tn + ti) * D.Cd, D.W, D.Cd, lane);
        }
        for (int i = threadIdx.x; i < nb * D.W; i += 256) {
            const int e = i / D.W, c = i - e * D.W, b = b0 + e;
            float a;
            if (wav_in) a = wav_in[b];
            else if (forced) a = t > 0 ? forced[(size_t)b * Tn + (t - 1)] : 0.f;
            else a = t > 0 ? state[L.a_prev + b] : 0.f;                 // fastgen.py:154: audio starts at 0
            const float u = D.mu ? wn_mu_law_scaled(a) : a;
            const float u1 = ur[((t + 3) & 3) * D.B + b], u2 = ur[((t + 2) & 3) * D.B + b];
            const float v = wb[3 * D.W + c] + wb[c] * u2 + wb[D.W + c] * u1 + wb[2 * D.W + c] * u;
            sh[i] = v;
            if (blockIdx.x == 0) {
                state[L.l + (size_t)b * D.W + c] = v;                                        // lin_0, buffer 0
                state[L.rings + (size_t)D.B * ring_off + ((size_t)(t % (2 * dil + 1)) * D.B + b) * D.W + c] = v;
                if (c == 0) ur[(t & 3) * D.B + b] = u;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < MB; ++e) {
            if (e >= nb || !(is_s || is_d)) break;
            const int b = b0 + e;
            const float* lin = sh + (size_t)e * D.W;
            if (is_s) {
                const float v = wave_sum(row_fma<NCH>(ws, D.W, lane, [&](int, int k) {
                    return *reinterpret_cast<const f4*>(lin + k); })) + bias;
                if (lane == 0) state[L.s + (size_t)b * D.S + o] = v;
            } else {
                const float v = wave_sum(row_fma<NCA>(wd, K, lane, [&](int i, int k) {
                    return (k >= 2 * D.W && k < 3 * D.W) ? *reinterpret_cast<const f4*>(lin + (k - 2 * D.W)) : gx[e].x[i]; })) + bias;
                if (lane == 0) state[L.dbuf + (size_t)b * D.G + r] = v;                       // d_0, buffer 0
            }
        }
        __syncthreads();
    }
}

// layer kernel j = 1..N (N = LAST): rows [0,W) lin_j (+ ring push), [W,W+S) s += skip_{j-1},
// [W+S, W+S+G) d_j.  LAST: only the skip rows (lin_N and d_N do not exist).
template <bool LAST, int NCA, int NCH, int MB>
__global__ __launch_bounds__(256) void ar_layer_m_kernel(
    float* __restrict__ state, ArStateLayout L, ArDims D, int cur, const float* __restrict__ Wrs,
    const float* __restrict__ brs, const float* __restrict__ Wd, const float* __restrict__ Wcomp,
    const float* __restrict__ bm, const float* __restrict__ enc, int Tn, int per_step, size_t ring_off, int dil) {
    extern __shared__ __attribute__((aligned(16))) float sh[];          // m [chunk][H] | lin_{j-1} [chunk][W]
    const int H = D.G / 2;
    float* shm = sh;
    float* shl = sh + (size_t)MB * H;
    const long long t = ar_step_of(state);
    const long long ti = per_step ? 0 : t;
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row = LAST ? o + D.W : o;                                 // row of [res | skip | d]
    const bool is_rs = row < D.W + D.S, is_d = !LAST && !is_rs && row < D.W + D.S + D.G;
    const int r = row - D.W - D.S, K = 3 * D.W + D.Cd;
    const float* ring = state + L.rings + (size_t)D.B * ring_off;
    const size_t slot2 = (size_t)((t + 1) % (2 * dil + 1)) * D.B;
    const size_t slot1 = (size_t)((t + dil + 1) % (2 * dil + 1)) * D.B;
    // all global loads of this wave's row are in flight before the gate prologue
    RowW<NCA> wd;
    RowW<NCH> wh;
    float bias = 0.f;
    if (is_d) {
        wd = row_load<NCA>(Wd + (size_t)r * K, K, lane);
        wh = row_load<NCH>(Wcomp + (size_t)r * H, H, lane);
        bias = bm[r];
    } else if (is_rs) {
        wh = row_load<NCH>(Wrs + (size_t)row * H, H, lane);
        bias = brs[row];
    }
    const float* dprev = state + L.dbuf + (size_t)cur * D.B * D.G;
    const float* lprev = state + (cur ? L.l2 : L.l);
    const int nxt = cur ^ 1;
    for (int b0 = 0; b0 < D.B; b0 += MB) {
        const int nb = min(MB, D.B - b0);
        GateX<NCA> gx[MB];
        float sold[MB];      // skip rows accumulate into s: fetched with the other loads
#pragma unroll
        for (int e = 0; e < MB; ++e) {
            sold[e] = 0.f;
            if (e < nb) {
                if (is_d)
                    gx[e] = gate_x_load<NCA>(ring + (slot2 + b0 + e) * D.W, ring + (slot1 + b0 + e) * D.W,
                                        enc + ((size_t)(b0 + e) * Tn + ti) * D.Cd, D.W, D.Cd, lane);
                if (is_rs && row >= D.W) sold[e] = state[L.s + (size_t)(b0 + e) * D.S + (row - D.W)];
            }
        }
        // gate of the previous layer (wavenet.py:479), recomputed by every workgroup
        for (int i = threadIdx.x; i < nb * H; i += 256) {
            const int e = i / H, k = i - e * H;
            const float* dp = dprev + (size_t)(b0 + e) * D.G;
            shm[i] = ar_gate(dp[k], dp[H + k]);
        }
        if (!LAST)
            for (int i = threadIdx.x; i < nb * D.W; i += 256) shl[i] = lprev[(size_t)b0 * D.W + i];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < MB; ++e) {
            if (e >= nb || !(is_rs || is_d)) break;
            const int b = b0 + e;
            const float* m = shm + (size_t)e * H;
            const float* lin = shl + (size_t)e * D.W;
            if (is_rs) {
                const float v = wave_sum(row_fma<NCH>(wh, H, lane, [&](int, int k) {
                    return *reinterpret_cast<const f4*>(m + k); })) + bias;
                if (lane == 0) {
                    if (row < D.W) {
                        const float ln = lin[row] + v;                                        // wavenet.py:481-485
                        state[(nxt ? L.l2 : L.l) + (size_t)b * D.W + row] = ln;
                        // lin_j is the INPUT of layer j: its queue slot of this step (masked.py:357-359)
                        state[L.rings + (size_t)D.B * ring_off + ((size_t)(t % (2 * dil + 1)) * D.B + b) * D.W + row] = ln;
                    } else {
                        state[L.s + (size_t)b * D.S + (row - D.W)] = sold[e] + v;             // wavenet.py:486-490
                    }
                }
            } else {
                float acc = row_fma<NCA>(wd, K, lane, [&](int i, int k) {
                    return (k >= 2 * D.W && k < 3 * D.W) ? *reinterpret_cast<const f4*>(lin + (k - 2 * D.W)) : gx[e].x[i]; });
                acc += row_fma<NCH>(wh, H, lane, [&](int, int k) { return *reinterpret_cast<const f4*>(m + k); });
                const float v = wave_sum(acc) + bias;
                if (lane == 0) state[L.dbuf + (size_t)nxt * D.B * D.G + (size_t)b * D.G + r] = v;
            }
        }
        __syncthreads();
    }
}

// Wcomp[o][k] = sum_c Wd[o][2W + c] * Wres[c][k]   (run once at wn_finalize; fp64 accumulation)
__global__ void ar_compose_kernel(const float* __restrict__ Wd, const float* __restrict__ Wrs,
                                  float* __restrict__ Wcomp, int G, int W, int H, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, o = blockIdx.y;
    if (k >= H || o >= G) return;
    double acc = 0.0;
    for (int c = 0; c < W; ++c) acc += (double)Wd[(size_t)o * K + 2 * W + c] * (double)Wrs[(size_t)c * H + k];
    Wcomp[(size_t)o * H + k] = (float)acc;
}

// ---- sampling heads (loss_func.py:140-206) + feedback de-quantisation (fastgen.py:163-167) ----
// One draw for one batch element by one workgroup (threads >= 256 only take part in the barriers).  `out(i)`: the
// network output, `rnd_at(j)`: the j-th random of this step, `e`: OW floats of scratch (CE head); the quantised
// sample is returned in thread 0.
template <class OutF, class RndF>
__device__ inline int ar_sample_q(const ArDims& D, OutF out, RndF rnd_at, float* e, float* red, float* sel_val, int tid) {
    int q = 0;
    const bool act = tid < 256;
    if (D.loss == WN_LOSS_MOL) {
        // loss_func.py:154-186
        if (tid < D.M) sel_val[tid] = out(tid) - logf(-logf(rnd_at(tid)));
        __syncthreads();
        if (tid == 0) {
            int k = 0;
            for (int i = 1; i < D.M; ++i) if (sel_val[i] > sel_val[k]) k = i;   // first max, like argmax
            const float mean = out(D.M + k);
            const float ls = fminf(fmaxf(out(2 * D.M + k), -7.f), 7.f);
            const float u2 = rnd_at(D.M);
            const float x = mean + expf(ls) * (logf(u2) - logf(1.f - u2));
            q = wn_clip_quantize(x, D.Q);
        }
    } else if (D.loss == WN_LOSS_GAUSS) {
        // loss_func.py:66-75,200-206
        if (tid == 0) {
            const float x = out(0) + expf(fmaxf(out(1), -7.f)) * rnd_at(0);
            q = wn_clip_quantize(x, D.Q);
        }
    } else {
        // categorical draw by inverse CDF from one uniform (sequential fp32 running sum)
        float mx = -INFINITY;
        if (act)
            for (int i = tid; i < D.OW; i += 256) mx = fmaxf(mx, out(i));
        if (act) red[tid] = mx;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
            __syncthreads();
        }
        mx = red[0];
        if (act)
            for (int i = tid; i < D.OW; i += 256) e[i] = expf(out(i) - mx);
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int i = 0; i < D.OW; ++i) tot += e[i];
            const float thr = rnd_at(0) * tot;
            float run = 0.f;
            int k = 0;
            for (int i = 0; i < D.OW; ++i) {
                run += e[i];
                if (run <= thr) k = i + 1;
            }
            q = min(k, D.OW - 1) - D.Q / 2;                      // loss_func.py:149
        }
    }
    return q;
}

// randoms of step t: injected [Tn][B][n_rand] or Philox(seed; step, batch, lane)
__device__ inline float ar_rnd_at(const ArDims& D, const float* rnd, int n_rand, uint64_t seed, long long t, long long ti,
                                  int b, int j) {
    if (rnd) return rnd[((size_t)ti * D.B + b) * n_rand + j];
    uint32_t c[4] = {(uint32_t)t, (uint32_t)(t >> 32), (uint32_t)b, (uint32_t)(j >> 2) + 0x41520000u};
    wn_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    if (D.loss == WN_LOSS_GAUSS) {
        const float u1 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
        return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * wn_u01(c[1]));
    }
    const float u = wn_u01(c[j & 3]);
    return D.loss == WN_LOSS_CE ? u : u * (1.f - 2e-5f) + 1e-5f;
}

// one workgroup per batch element
__global__ __launch_bounds__(256) void ar_sample_kernel(
    float* __restrict__ state, ArStateLayout L, ArDims D, const float* __restrict__ rnd, int n_rand,
    uint64_t seed, int per_step, int Tn, int* __restrict__ idx, float* __restrict__ wav,
    float* __restrict__ out_params) {
    __shared__ float red[256];
    __shared__ float sel_val[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long t = ar_step_of(state);
    const long long ti = per_step ? 0 : t;
    // network output of batch element b: [b][o] (GEMV layout) or [o][NB] (MFMA layout)
    const float* outb = state + L.out + (L.NB ? (size_t)b : (size_t)b * D.OW);
    const int so = L.NB ? L.NB : 1;
    auto out = [&](int i) -> float { return outb[(size_t)i * so]; };
    if (out_params)
        for (int i = tid; i < D.OW; i += 256) out_params[((size_t)b * Tn + ti) * D.OW + i] = out(i);
    auto rnd_at = [&](int j) -> float { return ar_rnd_at(D, rnd, n_rand, seed, t, ti, b, j); };
    const int q = ar_sample_q(D, out, rnd_at, state + L.ebuf + (size_t)b * D.OW, red, sel_val, tid);
    if (tid == 0) {
        const float a = wn_dequant(q, D.Q, D.mu);
        state[L.a_prev + b] = a;
        if (idx) idx[(size_t)b * Tn + ti] = q;
        if (wav) wav[(size_t)b * Tn + ti] = a;
        // the last workgroup to get here advances the step counter (every workgroup read it at its
        // start, before taking a ticket): no separate launch for the increment
        unsigned* ticket = reinterpret_cast<unsigned*>(state) + 4;
        __threadfence();
        if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
            *ticket = 0u;
            *reinterpret_cast<long long*>(state) = t + 1;
        }
    }
}

// =====================  batched step: the batch is the MFMA N dimension  =====================
// Weights are packed in A-fragment order ([row block][k-group of 16][lane][4], one coalesced
// 16-byte load per lane = 4 K-steps of v_mfma_f32_16x16x4_f32); activations are [feature][NB].
// One workgroup = one 16-row block x one chunk of 16*NT batch columns; its four waves split K
// (the step is a chain of latency-bound launches: more, shorter waves) and meet in LDS.

__global__ void ar_start_b_kernel(float* __restrict__ state, ArStateLayout L, ArDims D,
                                  const float* __restrict__ wav_in, const float* __restrict__ forced,
                                  const float* __restrict__ enc, int Tn, int per_step, const float* __restrict__ wb,
                                  size_t ring_off, int dil) {
    const long long t = ar_step_of(state);
    const long long ti = per_step ? 0 : t;
    const int NB = L.NB;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = D.W > D.Cd ? D.W : D.Cd;
    if (i >= rows * NB) return;
    const int c = i / NB, b = i - c * NB;
    float a = 0.f;
    if (b < D.B) {
        if (wav_in) a = wav_in[b];
        else if (forced) a = t > 0 ? forced[(size_t)b * Tn + (t - 1)] : 0.f;
        else a = t > 0 ? state[L.a_prev + b] : 0.f;
    }
    if (c < D.Cd) state[L.encT + (size_t)c * NB + b] = b < D.B ? enc[((size_t)b * Tn + ti) * D.Cd + c] : 0.f;
    if (c < D.W) {
        const float u = D.mu ? wn_mu_law_scaled(a) : a;
        float u1 = 0.f, u2 = 0.f;
        float* ur = state + L.uring;
        if (b < D.B) {
            u1 = ur[((t + 3) & 3) * D.B + b];
            u2 = ur[((t + 2) & 3) * D.B + b];
        }
        const float v = wb[3 * D.W + c] + wb[c] * u2 + wb[D.W + c] * u1 + wb[2 * D.W + c] * u;
        state[L.l + (size_t)c * NB + b] = v;
        // lin_0 is the input of the first causal layer: its queue slot of this step (masked.py:357-359)
        state[L.rings + ring_off * NB + ((size_t)(t % (2 * dil + 1)) * D.W + c) * NB + b] = v;
        if (c == 0 && b < D.B) ur[(t & 3) * D.B + b] = u;
    }
}

// The batched step is a chain of GEMMs [rows][K] x [K][NB]:
//   ar_layer_b_kernel   the res/skip GEMM of layer j-1 and the K-split pre-activation slabs of layer j
//   ar_gate_fin_b_kernel sums the slabs in a fixed order, adds the bias and applies sigmoid * tanh
//   ar_gemm_b_kernel    MODE 2 out1: z = relu(W [relu(s)|enc]);  MODE 3 out2: out = W z
// g[k][b] = sigmoid(bias[k] + sum_z slab[z][k][b]) * tanh(bias[m+k] + sum_z slab[z][m+k][b])  (wavenet.py:479)
__global__ void ar_gate_fin_b_kernel(float* __restrict__ state, ArStateLayout L, ArDims D,
                                     const float* __restrict__ bias, int nslab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = D.G / 2, NB = L.NB;
    if (i >= m * NB) return;
    const int k = i / NB, b = i - k * NB;
    const float* slab = state + L.slab;
    float vs[AR_MAXSLAB], vt[AR_MAXSLAB];
#pragma unroll
    for (int z = 0; z < AR_MAXSLAB; ++z)
        if (z < nslab) {
            vs[z] = slab[((size_t)z * D.G + k) * NB + b];
            vt[z] = slab[((size_t)z * D.G + m + k) * NB + b];
        }
    float hs = bias[k], ht = bias[m + k];
#pragma unroll
    for (int z = 0; z < AR_MAXSLAB; ++z)
        if (z < nslab) { hs += vs[z]; ht += vt[z]; }
    state[L.g + (size_t)k * NB + b] = (1.f / (1.f + expf(-hs))) * tanhf(ht);
}

template <int NT>
struct VecNT { float v[NT]; };
template <int NT>
__device__ inline VecNT<NT> load_nt(const float* p) {
    VecNT<NT> r;
    if (NT == 1) r.v[0] = p[0];
    else if (NT == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        r.v[0] = t.x; r.v[1 % NT] = t.y;
    } else {
        const f4 t = *reinterpret_cast<const f4*>(p);
#pragma unroll
        for (int e = 0; e < NT; ++e) r.v[e] = t[e & 3];
    }
    return r;
}

template <int MODE, int NT>
__global__ __launch_bounds__(1024) void ar_gemm_b_kernel(
    float* __restrict__ state, ArStateLayout L, ArDims D, const float* __restrict__ Ap,
    const float* __restrict__ bias, int rows, int K) {
    constexpr int NRB = 1;
    extern __shared__ __attribute__((aligned(16))) f4 red[];      // [waves][NT][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int NB = L.NB;
    const int mb = blockIdx.x;
    const int col0 = blockIdx.y * 16 * NT + NT * n;
    const int nks4 = K / 16;
    const int per = (nks4 + nw - 1) / nw;
    const int k4a = wave * per, k4b = min(nks4, k4a + per);
    // row pointer of input feature k (a k-group of 16 never straddles two sources)
    auto xrow = [&](int k) -> const float* {
        if (MODE == 2) return k < D.S ? state + L.s + (size_t)k * NB : state + L.encT + (size_t)(k - D.S) * NB;
        return state + L.z + (size_t)k * NB;
    };
    f4 acc[NRB][NT];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int e = 0; e < NT; ++e) acc[rb][e] = (f4){0.f, 0.f, 0.f, 0.f};
    const f4* A0 = reinterpret_cast<const f4*>(Ap) + (size_t)mb * nks4 * 64 + lane;
    for (int k4 = k4a; k4 < k4b; ++k4) {
        f4 a[NRB];
        a[0] = A0[(size_t)k4 * 64];
        float bv[4][NT];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int k = 16 * k4 + 4 * jj + q;
            const VecNT<NT> xv = load_nt<NT>(xrow(k) + col0);
#pragma unroll
            for (int e = 0; e < NT; ++e) bv[jj][e] = xv.v[e];
            if (MODE == 2 && 16 * k4 < D.S) {
#pragma unroll
                for (int e = 0; e < NT; ++e) bv[jj][e] = fmaxf(bv[jj][e], 0.f);     // wavenet.py:494
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int e = 0; e < NT; ++e)
                    acc[rb][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][jj], bv[jj][e], acc[rb][e], 0, 0, 0);
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int e = 0; e < NT; ++e) red[((wave * NRB + rb) * NT + e) * 64 + lane] = acc[rb][e];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        f4 v[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            v[rb] = red[(rb * NT + e) * 64 + lane];
            for (int w = 1; w < nw; ++w) v[rb] += red[((w * NRB + rb) * NT + e) * 64 + lane];   // fixed order
        }
        const int col = col0 + e;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mb + 4 * q + r;
            if (row < rows) {
                const float val = v[0][r] + bias[row];
                if (MODE == 2) state[L.z + (size_t)row * NB + col] = fmaxf(val, 0.f);   // wavenet.py:499
                else state[L.out + (size_t)row * NB + col] = val;
            }
        }
    }
}

// ---- merged batched layer kernel: the res/skip GEMM of layer j-1 and the K-split pre-activation
// slabs of layer j in ONE launch (both depend only on m_{j-1} and lin_{j-1}; see ar_layer_m_kernel
// for the algebra).  blockIdx.x < rs_blocks: 16 rows of [res | skip] (FIRST: skip_start), all of K;
// otherwise (row block, K slab) of [wd_j | wcomp_j] . [ring(t-2d) | ring(t-d) | lin_{j-1} | enc | m_{j-1}].
template <int NT, bool FIRST>
__global__ __launch_bounds__(256) void ar_layer_b_kernel(
    float* __restrict__ state, ArStateLayout L, ArDims D, int cur, const float* __restrict__ Ars,
    const float* __restrict__ brs, int rs_rows, int rs_K, const float* __restrict__ Adc, int Kd, int nslab,
    size_t ring_off, int dil) {
    extern __shared__ __attribute__((aligned(16))) f4 red[];      // [4 waves][NT][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int NB = L.NB;
    const int rs_blocks = (rs_rows + 15) / 16;
    const bool is_rs = (int)blockIdx.x < rs_blocks;
    const int task = is_rs ? blockIdx.x : blockIdx.x - rs_blocks;
    const int mb = is_rs ? task : task / nslab, z = is_rs ? 0 : task - mb * nslab;
    const int col0 = blockIdx.y * 16 * NT + NT * n;
    const long long t = ar_step_of(state);
    const int K = is_rs ? rs_K : Kd, nks4 = K / 16;
    const int g0 = is_rs ? 0 : 16 * z, g1 = is_rs ? nks4 : min(nks4, g0 + 16);
    const int per = (g1 - g0 + 3) / 4;
    const int k4a = g0 + wave * per, k4b = min(g1, k4a + per);
    const float* lcur = state + (cur ? L.l2 : L.l);
    float* lnxt = state + (cur ? L.l : L.l2);
    const float* ring = state + L.rings + ring_off * NB;
    const size_t slot2 = (size_t)((t + 1) % (2 * dil + 1)) * D.W, slot1 = (size_t)((t + dil + 1) % (2 * dil + 1)) * D.W;
    auto xrow = [&](int k) -> const float* {
        if (is_rs) return (FIRST ? lcur : state + L.g) + (size_t)k * NB;
        if (k < D.W) return ring + (slot2 + k) * NB;
        if (k < 2 * D.W) return ring + (slot1 + (k - D.W)) * NB;
        if (k < 3 * D.W) return lcur + (size_t)(k - 2 * D.W) * NB;
        if (k < 3 * D.W + D.Cd) return state + L.encT + (size_t)(k - 3 * D.W) * NB;
        return state + L.g + (size_t)(k - 3 * D.W - D.Cd) * NB;
    };
    f4 acc[NT];
#pragma unroll
    for (int e = 0; e < NT; ++e) acc[e] = (f4){0.f, 0.f, 0.f, 0.f};
    const f4* A0 = reinterpret_cast<const f4*>(is_rs ? Ars : Adc) + (size_t)mb * nks4 * 64 + lane;
    for (int k4 = k4a; k4 < k4b; ++k4) {
        const f4 a = A0[(size_t)k4 * 64];
        float bv[4][NT];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const VecNT<NT> xv = load_nt<NT>(xrow(16 * k4 + 4 * jj + q) + col0);
#pragma unroll
            for (int e = 0; e < NT; ++e) bv[jj][e] = xv.v[e];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int e = 0; e < NT; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], bv[jj][e], acc[e], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < NT; ++e) red[(wave * NT + e) * 64 + lane] = acc[e];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int e = 0; e < NT; ++e) {
        f4 v = red[e * 64 + lane];
        for (int w = 1; w < 4; ++w) v += red[(w * NT + e) * 64 + lane];                  // fixed order
        const int col = col0 + e;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mb + 4 * q + r;
            if (!is_rs) {
                state[L.slab + ((size_t)z * D.G + row) * NB + col] = v[r];
            } else if (row < rs_rows) {
                const float val = v[r] + brs[row];
                if (FIRST) state[L.s + (size_t)row * NB + col] = val;
                else if (row < D.W) {
                    const float ln = lcur[(size_t)row * NB + col] + val;
                    lnxt[(size_t)row * NB + col] = ln;
                    if (Adc)   // lin_j feeds layer j: its queue slot of this step (masked.py:357-359)
                        state[L.rings + ring_off * NB + ((size_t)(t % (2 * dil + 1)) * D.W + row) * NB + col] = ln;
                } else {
                    state[L.s + (size_t)(row - D.W) * NB + col] += val;
                }
            }
        }
    }
}

// A-fragment order of [wd | wcomp] (rows G, K = KA + H), built on the device after the composite
__global__ void ar_frag_dc_kernel(const float* __restrict__ wd, const float* __restrict__ wcomp,
                                  float* __restrict__ dst, int G, int KA, int H) {
    const int nks4 = (KA + H) / 16;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(G / 16) * nks4 * 256;
    if (i >= total) return;
    const int jj = i & 3, lane = (i >> 2) & 63;
    const size_t blk = i >> 8;
    const int k4 = blk % nks4, mb = blk / nks4;
    const int row = 16 * mb + (lane & 15), k = 16 * k4 + 4 * jj + (lane >> 4);
    dst[i] = k < KA ? wd[(size_t)row * KA + k] : wcomp[(size_t)row * H + (k - KA)];
}

ArDims ar_dims(const wn_handle* h, int B) {
    const wn_config& c = h->cfg;
    ArDims D;
    D.B = B; D.W = c.width; D.S = c.skip_width; D.G = c.gate_width; D.Cd = c.deconv_width;
    D.OW = c.out_width; D.Q = c.use_mu_law ? 256 : 65536; D.mu = c.use_mu_law; D.loss = c.loss_type;
    D.M = c.loss_type == WN_LOSS_MOL ? c.mol_mix : 0;
    return D;
}

// enqueue the kernels of ONE step (wavenet.py:408-501 + sampler + queue pushes)
template <int NT>
void ar_enqueue_step_b(wn_handle* h, float* state, const ArStateLayout& L, const ArDims& D, const float* wav_in,
                       const float* forced, const float* enc, int Tn, int per_step, const float* rnd,
                       uint64_t seed, int* idx, float* wav, float* out_params, hipStream_t st) {
    const ArPack& P = h->ar;
    const float* blob = h->d_blob;
    const int NB = L.NB, chunks = NB / (16 * NT);
    const int rows = D.W > D.Cd ? D.W : D.Cd;
    // waves per workgroup = K-split: as many as there are k-groups / 4, capped by 16 waves and
    // by 64 KB of LDS for the partial accumulators
    auto nwaves = [&](int K, int nrb) {
        int w = K / 64;
        const int cap = 65536 / (nrb * NT * 1024);
        w = w > 16 ? 16 : w;
        return w > cap ? cap : (w < 1 ? 1 : w);
    };
    auto lds = [&](int w, int nrb) { return (size_t)w * nrb * NT * 1024; };
    const int w2 = nwaves(D.S + D.Cd, 1), w3 = nwaves(D.S, 1);
    const size_t n = P.layers.size();
    const int KA = 3 * D.W + D.Cd, H = D.G / 2;
    const ArLayerPack& l0 = P.layers[0];
    hipLaunchKernelGGL(ar_start_b_kernel, dim3((rows * NB + 255) / 256), dim3(256), 0, st, state, L, D, wav_in,
                       forced, enc, Tn, per_step, blob + P.start_off, l0.ring_off, l0.dilation);
    // layer kernels: [res/skip of layer j-1 | pre-activation slabs of layer j] in one launch, then the
    // slab sum + gate; lin ping-pongs between the two residual buffers (buffer 0 holds lin_0)
    {
        const int nslab = (KA + 255) / 256;
        hipLaunchKernelGGL((ar_layer_b_kernel<NT, true>), dim3(D.S / 16 + D.G / 16 * nslab, chunks), dim3(256), lds(4, 1), st,
                           state, L, D, 0, blob + P.wss_b_off, blob + P.bss_off, D.S, D.W, blob + l0.wd_b_off, KA, nslab,
                           l0.ring_off, l0.dilation);
        hipLaunchKernelGGL(ar_gate_fin_b_kernel, dim3((H * NB + 255) / 256), dim3(256), 0, st, state, L, D,
                           blob + l0.bd_off, nslab);
    }
    for (size_t j = 1; j <= n; ++j) {
        const ArLayerPack& pv = P.layers[j - 1];
        const int cur = (int)((j - 1) & 1);
        if (j < n) {
            const ArLayerPack& lp = P.layers[j];
            const int Kd = KA + H, nslab = (Kd + 255) / 256;
            hipLaunchKernelGGL((ar_layer_b_kernel<NT, false>), dim3((D.W + D.S) / 16 + D.G / 16 * nslab, chunks), dim3(256),
                               lds(4, 1), st, state, L, D, cur, blob + pv.wrs_b_off, blob + pv.brs_off, D.W + D.S, H,
                               blob + lp.wdc_b_off, Kd, nslab, lp.ring_off, lp.dilation);
            hipLaunchKernelGGL(ar_gate_fin_b_kernel, dim3((H * NB + 255) / 256), dim3(256), 0, st, state, L, D,
                               blob + lp.bm_off, nslab);
        } else {
            // after the last layer only the skip sum is consumed
            hipLaunchKernelGGL((ar_layer_b_kernel<NT, false>), dim3((D.W + D.S) / 16, chunks), dim3(256), lds(4, 1), st,
                               state, L, D, cur, blob + pv.wrs_b_off, blob + pv.brs_off, D.W + D.S, H,
                               (const float*)nullptr, 0, 1, (size_t)0, 1);
        }
    }
    hipLaunchKernelGGL((ar_gemm_b_kernel<2, NT>), dim3(D.S / 16, chunks), dim3(64 * w2), lds(w2, 1), st, state, L, D,
                       blob + P.wo1_b_off, blob + P.bo1_off, D.S, D.S + D.Cd);
    hipLaunchKernelGGL((ar_gemm_b_kernel<3, NT>), dim3((D.OW + 15) / 16, chunks), dim3(64 * w3), lds(w3, 1), st, state, L, D,
                       blob + P.wo2_b_off, blob + P.bo2_off, D.OW, D.S);
    hipLaunchKernelGGL(ar_sample_kernel, dim3(D.B), dim3(256), 0, st, state, L, D, rnd, wn_ar_n_rand(h), seed,
                       per_step, Tn, idx, wav, out_params);
}

void ar_enqueue_step(wn_handle* h, float* state, int B, const float* wav_in, const float* forced,
                     const float* enc, int Tn, int per_step, const float* rnd, uint64_t seed, int* idx,
                     float* wav, float* out_params, hipStream_t st) {
    const ArStateLayout L = ar_state_layout(h, B);
    const ArDims D = ar_dims(h, B);
    if (L.NB == 16) return ar_enqueue_step_b<1>(h, state, L, D, wav_in, forced, enc, Tn, per_step, rnd, seed, idx, wav, out_params, st);
    if (L.NB == 32) return ar_enqueue_step_b<2>(h, state, L, D, wav_in, forced, enc, Tn, per_step, rnd, seed, idx, wav, out_params, st);
    if (L.NB) return ar_enqueue_step_b<4>(h, state, L, D, wav_in, forced, enc, Tn, per_step, rnd, seed, idx, wav, out_params, st);
    const ArPack& P = h->ar;
    const float* blob = h->d_blob;
    {
        // merged step: one launch per layer (see ar_layer_m_kernel); the wide instantiation for rows past the default tiles
        const size_t n = P.layers.size();
        const ArLayerPack& l0 = P.layers[0];
        const bool wide = ar_wide(h->cfg);
        const int mb = wide ? AR_GEMV_MAXB_W : AR_GEMV_MAXB;
        auto first = wide ? ar_first_m_kernel<AR_NCA_W, AR_NCH_W, AR_GEMV_MAXB_W> : ar_first_m_kernel<AR_NCA, AR_NCH, AR_GEMV_MAXB>;
        auto layer = wide ? ar_layer_m_kernel<false, AR_NCA_W, AR_NCH_W, AR_GEMV_MAXB_W> : ar_layer_m_kernel<false, AR_NCA, AR_NCH, AR_GEMV_MAXB>;
        auto lastk = wide ? ar_layer_m_kernel<true, AR_NCA_W, AR_NCH_W, AR_GEMV_MAXB_W> : ar_layer_m_kernel<true, AR_NCA, AR_NCH, AR_GEMV_MAXB>;
        hipLaunchKernelGGL(first, dim3((D.S + D.G + 3) / 4), dim3(256), (size_t)mb * D.W * sizeof(float), st,
                           state, L, D, wav_in, forced, Tn, blob + P.start_off, blob + P.wss_off, blob + P.bss_off,
                           blob + l0.wd_off, blob + l0.bd_off, enc, per_step, l0.ring_off, l0.dilation);
        const size_t shb = (size_t)mb * (D.G / 2 + D.W) * sizeof(float);
        for (size_t j = 1; j < n; ++j) {
            const ArLayerPack& lp = P.layers[j];
            const ArLayerPack& pv = P.layers[j - 1];
            hipLaunchKernelGGL(layer, dim3((D.W + D.S + D.G + 3) / 4), dim3(256), shb, st, state, L, D,
                               (int)((j - 1) & 1), blob + pv.wrs_off, blob + pv.brs_off, blob + lp.wd_off,
                               blob + lp.wcomp_off, blob + lp.bm_off, enc, Tn, per_step, lp.ring_off, lp.dilation);
        }
        const ArLayerPack& pl = P.layers[n - 1];
        hipLaunchKernelGGL(lastk, dim3((D.S + 3) / 4), dim3(256), shb, st, state, L, D,
                           (int)((n - 1) & 1), blob + pl.wrs_off, blob + pl.brs_off, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, enc, Tn, per_step, (size_t)0, 1);
        hipLaunchKernelGGL(ar_rows_kernel<2>, dim3((D.S + 3) / 4), dim3(256), 0, st, state, L, D, blob + P.wo1_off,
                           blob + P.bo1_off, D.S, D.S + D.Cd, enc, Tn, per_step);
        hipLaunchKernelGGL(ar_rows_kernel<3>, dim3((D.OW + 3) / 4), dim3(256), 0, st, state, L, D, blob + P.wo2_off,
                           blob + P.bo2_off, D.OW, D.S, enc, Tn, per_step);
        hipLaunchKernelGGL(ar_sample_kernel, dim3(B), dim3(256), 0, st, state, L, D, rnd, wn_ar_n_rand(h), seed,
                           per_step, Tn, idx, wav, out_params);
        }
}

// The hipGraphs of ONE wn_ar_generate call.  They must outlive the call (it returns before its stream has run them), so
// the handle keeps a list: every call adds its own entry, and retires the entries whose `done` event -- recorded behind
// the last replay -- has passed.  Nothing of a call's graphs is shared with another call: concurrent callers of one
// handle (each with its own state buffer and stream) never touch each other's entries.
struct ArGraphCache {
    hipGraphExec_t exec_multi = nullptr, exec_one = nullptr;
    hipGraph_t g_multi = nullptr, g_one = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

void ar_cache_free(ArGraphCache* c) {
    if (!c) return;
    if (c->done) {
        (void)hipEventSynchronize(c->done);
        (void)hipEventDestroy(c->done);
    } else if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
    }
    if (c->exec_multi) (void)hipGraphExecDestroy(c->exec_multi);
    if (c->exec_one) (void)hipGraphExecDestroy(c->exec_one);
    if (c->g_multi) (void)hipGraphDestroy(c->g_multi);
    if (c->g_one) (void)hipGraphDestroy(c->g_one);
    delete c;
}

}  // namespace

// Process-wide: held while a thread captures AND while finished graphs are retired -- hipEventQuery on an event last recorded
// in a stream that another thread is capturing on is refused by the runtime and invalidates that capture (the `done` events of
// every caller's graphs hang off the shared handle).
static std::mutex g_capture_mu;

void wn_ar_release(wn_handle* h) {
    std::vector<void*> all;
    {
        std::lock_guard<std::mutex> g(h->list_mu);
        all.swap(h->ar_graphs);
    }
    for (void* p : all) ar_cache_free(reinterpret_cast<ArGraphCache*>(p));
    std::lock_guard<std::mutex> cap(g_capture_mu);
    if (h->ar_cap_stream) (void)hipStreamDestroy(reinterpret_cast<hipStream_t>(h->ar_cap_stream));
    h->ar_cap_stream = nullptr;
    (void)hipGetLastError();
}

// entries whose stream has finished replaying them

static void ar_retire_finished(wn_handle* h) {
    std::lock_guard<std::mutex> cap(g_capture_mu);
    std::vector<void*> gone;
    {
        std::lock_guard<std::mutex> g(h->list_mu);
        size_t k = 0;
        for (void* p : h->ar_graphs) {
            ArGraphCache* c = reinterpret_cast<ArGraphCache*>(p);
            if (c->done && hipEventQuery(c->done) == hipSuccess) gone.push_back(p);
            else h->ar_graphs[k++] = p;
        }
        h->ar_graphs.resize(k);
    }
    (void)hipGetLastError();                   // hipEventQuery's hipErrorNotReady is not an error of this call
    for (void* p : gone) ar_cache_free(reinterpret_cast<ArGraphCache*>(p));
}

// ---------------------------------------------------------------------------
static bool K_ok(int k) { return k % 64 == 0; }   // every K of the batched GEMMs is a multiple of 64 (4 waves x 16)

int wn_pack_ar(wn_handle* h, std::vector<float>& blob) {
    const wn_config& c = h->cfg;
    const int W = c.width, S = c.skip_width, G = c.gate_width, Cd = c.deconv_width, OW = c.out_width;
    auto var = [&](const std::string& nme) -> const std::vector<float>& { return h->vars.at(nme).data; };
    auto begin = [&]() { blob.resize(align_up(blob.size(), 64)); return blob.size(); };
    ArPack& P = h->ar;
    {
        P.start_off = begin();
        std::vector<float> Wst = wn_get_kernel(h, "conv_start", "W", false);      // [3][1][W]
        blob.insert(blob.end(), Wst.begin(), Wst.end());
        const auto& b = var("conv_start/biases");
        blob.insert(blob.end(), b.begin(), b.end());
    }
    auto pack_T = [&](const std::vector<float>& Wsrc, int cin, int cout, float* dst, int ld, int col0) {
        // HWIO [1,1,cin,cout] -> dst[o*ld + col0 + ci]
        for (int ci = 0; ci < cin; ++ci)
            for (int o = 0; o < cout; ++o) dst[(size_t)o * ld + col0 + ci] = Wsrc[(size_t)ci * cout + o];
    };
    {
        P.wss_off = begin();
        blob.resize(blob.size() + (size_t)S * W);
        pack_T(wn_get_kernel(h, "skip_start", "W", false), W, S, blob.data() + P.wss_off, W, 0);
        P.bss_off = begin();
        const auto& b = var("skip_start/biases");
        blob.insert(blob.end(), b.begin(), b.end());
    }
    size_t ring = 0;
    for (int i = 0; i < c.num_layers; ++i) {
        const std::string s = std::to_string(i + 1);
        ArLayerPack lp;
        lp.dilation = 1 << (i % c.num_stages);                                     // wavenet.py:453
        lp.ring_off = ring;
        ring += (size_t)(2 * lp.dilation + 1) * W;      // 2d+1 slots: the merged step reads t-2d while t is pushed
        const int K = 3 * W + Cd;
        std::vector<float> Wd = wn_get_kernel(h, "dilated_conv_" + s, "W", false);   // [3][W][G]
        lp.wd_off = begin();
        blob.resize(blob.size() + (size_t)G * K);
        float* dst = blob.data() + lp.wd_off;
        for (int tap = 0; tap < 3; ++tap)      // tap 0 <-> x[t-2d] (masked.py:369-371)
            for (int ci = 0; ci < W; ++ci)
                for (int o = 0; o < G; ++o)
                    dst[(size_t)o * K + tap * W + ci] = Wd[((size_t)tap * W + ci) * G + o];
        pack_T(wn_get_kernel(h, "mel_cond_" + s, "W", false), Cd, G, dst, K, 3 * W);
        lp.bd_off = begin();
        {
            const auto& bd = var("dilated_conv_" + s + "/biases");
            const auto& bc = var("mel_cond_" + s + "/biases");
            for (int o = 0; o < G; ++o) blob.push_back(bd[o] + bc[o]);
            lp.bc_off = begin();
            blob.insert(blob.end(), bc.begin(), bc.end());
        }
        lp.wrs_off = begin();
        blob.resize(blob.size() + (size_t)(W + S) * (G / 2));
        dst = blob.data() + lp.wrs_off;
        pack_T(wn_get_kernel(h, "res_" + s, "W", false), G / 2, W, dst, G / 2, 0);
        pack_T(wn_get_kernel(h, "skip_" + s, "W", false), G / 2, S, dst + (size_t)W * (G / 2), G / 2, 0);
        lp.brs_off = begin();
        {
            const auto& br = var("res_" + s + "/biases");
            const auto& bs = var("skip_" + s + "/biases");
            blob.insert(blob.end(), br.begin(), br.end());
            blob.insert(blob.end(), bs.begin(), bs.end());
        }
        // merged-step additions: space for the composite matrix (filled on the device by
        // wn_ar_post_upload) and the composite bias bm = bd + Wd[tap t] . bres_{j-1}
        if (i > 0) {
            const ArLayerPack& pv = P.layers.back();
            lp.wcomp_off = begin();
            blob.resize(blob.size() + (size_t)G * (G / 2));
            lp.bm_off = begin();
            blob.resize(blob.size() + G);
            for (int o = 0; o < G; ++o) {
                double acc = blob[lp.bd_off + o];
                for (int cc = 0; cc < W; ++cc)
                    acc += (double)blob[lp.wd_off + (size_t)o * K + 2 * W + cc] * (double)blob[pv.brs_off + cc];
                blob[lp.bm_off + o] = (float)acc;
            }
        }
        P.layers.push_back(lp);
    }
    P.ring_floats = ring;
    {
        P.wo1_off = begin();
        blob.resize(blob.size() + (size_t)S * (S + Cd));
        float* dst = blob.data() + P.wo1_off;
        pack_T(wn_get_kernel(h, "out1", "W", false), S, S, dst, S + Cd, 0);
        pack_T(wn_get_kernel(h, "mel_cond_out1", "W", false), Cd, S, dst, S + Cd, S);
        P.bo1_off = begin();
        const auto& b1 = var("out1/biases");
        const auto& b2 = var("mel_cond_out1/biases");
        for (int o = 0; o < S; ++o) blob.push_back(b1[o] + b2[o]);
        P.bco1_off = begin();
        blob.insert(blob.end(), b2.begin(), b2.end());
        P.wo2_off = begin();
        blob.resize(blob.size() + (size_t)OW * S);
        pack_T(wn_get_kernel(h, "out2", "W", false), S, OW, blob.data() + P.wo2_off, S, 0);
        P.bo2_off = begin();
        const auto& b3 = var("out2/biases");
        blob.insert(blob.end(), b3.begin(), b3.end());
        blob.resize(align_up(blob.size(), 64) + 64);      // slack: bias reads of padded rows stay in bounds
    }
    // A-fragment-order copies for the batched (MFMA) step: [row block][k-group of 16][lane][4],
    // lane (i = lane&15, kq = lane>>4), element jj <-> W[16*mb + i][16*k4 + 4*jj + kq]
    auto frag = [&](size_t src_off, int rows, int K) -> size_t {
        const int mbs = (rows + 15) / 16, nks4 = K / 16;
        const size_t off = begin();
        blob.resize(blob.size() + (size_t)mbs * nks4 * 256);
        const float* src = blob.data() + src_off;
        float* dst = blob.data() + off;
        for (int mb = 0; mb < mbs; ++mb)
            for (int k4 = 0; k4 < nks4; ++k4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int jj = 0; jj < 4; ++jj) {
                        const int row = 16 * mb + (lane & 15), k = 16 * k4 + 4 * jj + (lane >> 4);
                        dst[(((size_t)mb * nks4 + k4) * 64 + lane) * 4 + jj] = row < rows ? src[(size_t)row * K + k] : 0.f;
                    }
        return off;
    };
    P.wss_b_off = 0;
    if (K_ok(W) && K_ok(Cd) && K_ok(S) && K_ok(G / 2) && (3 * W + Cd + G / 2 + 255) / 256 <= AR_MAXSLAB) {
        P.wss_b_off = frag(P.wss_off, S, W);
        for (size_t li = 0; li < P.layers.size(); ++li) {
            ArLayerPack& lp = P.layers[li];
            if (li == 0) {
                lp.wd_b_off = frag(lp.wd_off, G, 3 * W + Cd);
            } else {                         // [wd | wcomp] fragments are written on the device (wn_ar_post_upload)
                lp.wdc_b_off = begin();
                blob.resize(blob.size() + (size_t)G * (3 * W + Cd + G / 2));
            }
            lp.wrs_b_off = frag(lp.wrs_off, W + S, G / 2);
            // bias vector of the batched res/skip kernel: [res | skip | gate (dilated + cond)]
            lp.brs_gate_off = begin();
            for (int i = 0; i < W + S; ++i) blob.push_back(blob[lp.brs_off + i]);
            for (int i = 0; i < G; ++i) blob.push_back(blob[lp.bd_off + i]);
        }
        P.wo1_b_off = frag(P.wo1_off, S, S + Cd);
        P.wo2_b_off = frag(P.wo2_off, OW, S);
    }
    return WN_OK;
}

int wn_ar_post_upload(wn_handle* h) {
    const wn_config& c = h->cfg;
    const int W = c.width, G = c.gate_width, H = G / 2, K = 3 * W + c.deconv_width;
    for (size_t j = 1; j < h->ar.layers.size(); ++j) {
        const ArLayerPack& lp = h->ar.layers[j];
        const ArLayerPack& pv = h->ar.layers[j - 1];
        hipLaunchKernelGGL(ar_compose_kernel, dim3((H + 255) / 256, G), dim3(256), 0, 0, h->d_blob + lp.wd_off,
                           h->d_blob + pv.wrs_off, h->d_blob + lp.wcomp_off, G, W, H, K);
        if (lp.wdc_b_off) {
            const size_t total = (size_t)G * (K + H);
            hipLaunchKernelGGL(ar_frag_dc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0,
                               h->d_blob + lp.wd_off, h->d_blob + lp.wcomp_off, h->d_blob + lp.wdc_b_off, G, K, H);
        }
    }
    WN_HIP(h, hipDeviceSynchronize());
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

extern "C" int wn_ar_n_rand(const wn_handle* h) {
    if (!h || h->cfg.kind != WN_KIND_TEACHER) return WN_EINVAL;
    return h->cfg.loss_type == WN_LOSS_MOL ? h->cfg.mol_mix + 1 : 1;
}

extern "C" size_t wn_ar_state_bytes(const wn_handle* h, int B) {
    if (!h || h->cfg.kind != WN_KIND_TEACHER || !h->finalized || B < 1) return 0;
    return align_up(ar_state_layout(h, B).total * sizeof(float), 256);
}

static int ar_check(wn_handle* h, const char* fn, int B) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "%s: null handle", fn);
    if (!h->finalized) return wn_fail(h, WN_ESTATE, "%s: call wn_finalize first", fn);
    if (h->cfg.kind != WN_KIND_TEACHER) return wn_fail(h, WN_EINVAL, "%s: handle is not a teacher Wavenet", fn);
    if (B < 1) return wn_fail(h, WN_EINVAL, "%s: B must be >= 1", fn);
    return WN_OK;
}

extern "C" int wn_ar_reset(wn_handle* h, void* state, int B, void* stream) {
    int rc = ar_check(h, "wn_ar_reset", B);
    if (rc) return rc;
    if (!state) return wn_fail(h, WN_EINVAL, "wn_ar_reset: null state");
    // masked.py:352-355: both queues of every causal layer start as `rate` zeros
    WN_HIP(h, hipMemsetAsync(state, 0, wn_ar_state_bytes(h, B), reinterpret_cast<hipStream_t>(stream)));
    return WN_OK;
}

extern "C" int wn_ar_step(wn_handle* h, void* state, int B, const float* wav_in, const float* enc_t,
                          const float* rnd, uint64_t seed, int32_t* sample, float* out_params, void* stream) {
    int rc = ar_check(h, "wn_ar_step", B);
    if (rc) return rc;
    const WnWork work(h);
    if (!state || !wav_in || !enc_t || !sample) return wn_fail(h, WN_EINVAL, "wn_ar_step: null argument");
    ar_enqueue_step(h, reinterpret_cast<float*>(state), B, wav_in, nullptr, enc_t, 1, 1, rnd, seed, sample,
                    nullptr, out_params, reinterpret_cast<hipStream_t>(stream));
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

extern "C" int wn_ar_generate(wn_handle* h, const float* enc, int B, int Tn, const float* rnd, uint64_t seed,
                              int32_t* idx, float* wav, const float* forced_wav, float* out_params, void* ws,
                              size_t ws_bytes, void* stream) {
    int rc = ar_check(h, "wn_ar_generate", B);
    if (rc) return rc;
    const WnWork work(h);                       // wn_ar_set_graph is refused until this call returns; read once:
    const bool use_graph = h->ar_use_graph;
    if (Tn < 0) return wn_fail(h, WN_EINVAL, "wn_ar_generate: negative length");
    if (Tn == 0) return WN_OK;
    if (!enc || !ws || (!idx && !wav)) return wn_fail(h, WN_EINVAL, "wn_ar_generate: null argument");
    const size_t need = wn_ar_state_bytes(h, B);
    if (ws_bytes < need) return wn_fail(h, WN_ENOMEM, "wn_ar_generate: workspace %zu < %zu bytes", ws_bytes, need);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* state = reinterpret_cast<float*>(ws);
    WN_HIP(h, hipMemsetAsync(state, 0, need, st));

    auto plain = [&](int n) -> int {
        for (int t = 0; t < n; ++t)
            ar_enqueue_step(h, state, B, nullptr, forced_wav, enc, Tn, 0, rnd, seed, idx, wav, out_params, st);
        WN_HIP(h, hipGetLastError());
        return WN_OK;
    };
    // The legacy null stream cannot be captured; WN_AR_GRAPH=0 forces plain launches (A/B runs, tests).
    const char* eg = getenv("WN_AR_GRAPH");
    if (!st || !use_graph || (eg && eg[0] == '0')) return plain(Tn);
    // Every per-step address is derived on the device from the step counter, so a captured step is static:
    // build (AR_GRAPH_STEPS steps) + (1 step) graphs and replay them on the caller's stream.
    // The steps are captured on a PRIVATE non-blocking stream of the handle, never on the caller's: a capture can be
    // invalidated from outside -- on ROCm 7.2 a device-wide synchronise of ANY thread does it, thread-local capture mode or
    // not -- and the runtime then leaves that stream unusable for good (status "invalidated"; hipStreamEndCapture answers
    // "attempt to terminate a thread-local capture sequence from another thread" to the thread that began it, every later launch
    // on the stream fails).  Lost that way, the private stream is dropped and replaced, and the call falls back to plain launches
    // on the caller's stream, which was never in capture mode.  (A non-blocking stream also keeps legacy-stream work of
    // other threads legal during the capture.)  Captures and the retiring of finished graphs are serialised by g_capture_mu.
    ar_retire_finished(h);
    ArGraphCache* gc = new ArGraphCache();
    gc->stream = st;
    static const bool dbg = getenv("WN_AR_DEBUG") != nullptr;
    auto capture_once = [&](int nsteps, hipGraph_t* g, hipGraphExec_t* ex) -> bool {
        std::lock_guard<std::mutex> lk(g_capture_mu);
        if (!h->ar_cap_stream) {
            hipStream_t cs = nullptr;
            if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            h->ar_cap_stream = cs;
        }
        hipStream_t cap = reinterpret_cast<hipStream_t>(h->ar_cap_stream);
        auto drop_stream = [&]() {               // (destroying an invalidated stream may fail too: then it is leaked, once)
            (void)hipStreamDestroy(cap);
            h->ar_cap_stream = nullptr;
            (void)hipGetLastError();
        };
        const hipError_t e0 = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
        if (e0 != hipSuccess) {
            if (dbg) fprintf(stderr, "wn_ar_generate: begin capture failed: %s\n", hipGetErrorString(e0));
            drop_stream();
            return false;
        }
        for (int i = 0; i < nsteps; ++i)
            ar_enqueue_step(h, state, B, nullptr, forced_wav, enc, Tn, 0, rnd, seed, idx, wav, out_params, cap);
        const hipError_t e1 = hipGetLastError();
        const hipError_t e2 = hipStreamEndCapture(cap, g);
        hipError_t e3 = hipSuccess;
        if (e1 != hipSuccess || e2 != hipSuccess || !*g || (e3 = hipGraphInstantiate(ex, *g, nullptr, nullptr, 0)) != hipSuccess) {
            if (dbg) fprintf(stderr, "wn_ar_generate: capture failed: launches %s, end %s, instantiate %s\n",
                             hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
            if (*g) (void)hipGraphDestroy(*g);
            *g = nullptr;
            *ex = nullptr;
            drop_stream();
            return false;
        }
        return true;
    };
    auto capture = [&](int nsteps, hipGraph_t* g, hipGraphExec_t* ex) -> bool {
        return capture_once(nsteps, g, ex) || capture_once(nsteps, g, ex);     // (once more, on a fresh private stream)
    };
    const int multi = Tn / AR_GRAPH_STEPS, rest = Tn % AR_GRAPH_STEPS;
    bool ok = true;
    if (multi) ok = capture(AR_GRAPH_STEPS, &gc->g_multi, &gc->exec_multi);
    if (ok && rest) ok = capture(1, &gc->g_one, &gc->exec_one);
    if (!ok) {
        gc->stream = nullptr;                   // nothing of it was launched
        ar_cache_free(gc);
        return plain(Tn);
    }
    hipError_t e = hipSuccess;
    for (int i = 0; i < multi && e == hipSuccess; ++i) e = hipGraphLaunch(gc->exec_multi, st);
    for (int i = 0; i < rest && e == hipSuccess; ++i) e = hipGraphLaunch(gc->exec_one, st);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&gc->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(gc->done, st);
    {
        std::lock_guard<std::mutex> g(h->list_mu);
        h->ar_graphs.push_back(gc);             // kept until its stream has run it (ar_retire_finished / wn_destroy)
    }
    if (e != hipSuccess) return wn_fail(h, WN_EIO, "wn_ar_generate: graph replay failed: %s", hipGetErrorString(e));
    return WN_OK;
}

// ---------------------------------------------------------------------------
// Fastgen.cond_vars (wavenet/wavenet.py:353-377; fastgen.calculate_cond_vars, fastgen.py:91-115): every layer's 1x1
// conditioning projection mel_cond_i(encoding) -- and mel_cond_out1 -- evaluated in bulk over time, biases included, in
// the reference's [B, T, channels] layout.  The reference builds it "for data visualization"; the sampling loop
// (Fastgen.sample) keeps evaluating the projections per step, and so does the step here: the conditioning columns are 256
// of the 2 048 inputs of a gate row, the step is a chain of launch-latency-bound kernels at every batch size, and a timing
// ablation without them (profiles/r05_ar_step_variants.txt) bounds what hoisting could buy.
// out[col][g] = bias[g] + sum_c W[g][koff + c] * enc[col][c]: a plain LDS-tiled fp32 GEMM, 64 x 64 outputs per workgroup.
namespace {
constexpr int CV_T = 64, CV_K = 16;
__global__ __launch_bounds__(256) void ar_cond_vars_kernel(const float* __restrict__ Wm, int ldw, int koff, const float* __restrict__ bias,
                                                           const float* __restrict__ enc, int Cd, int rows, int64_t cols, float* __restrict__ out) {
    __shared__ float Ws[CV_K][CV_T + 1], Es[CV_K][CV_T + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;            // 4 x 4 outputs per thread: rows 4 tx.., columns 4 ty..
    const int g0 = blockIdx.x * CV_T;
    const int64_t c0 = (int64_t)blockIdx.y * CV_T;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < Cd; k0 += CV_K) {
        for (int i = threadIdx.x; i < CV_T * CV_K; i += 256) {
            const int r = i / CV_K, k = i - r * CV_K;
            Ws[k][r] = (g0 + r < rows) ? Wm[(size_t)(g0 + r) * ldw + koff + k0 + k] : 0.f;
            Es[k][r] = (c0 + r < cols) ? enc[(size_t)(c0 + r) * Cd + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CV_K; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = Ws[k][4 * tx + i]; b[i] = Es[k][4 * ty + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][i] = fmaf(a[i], b[j], acc[j][i]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t col = c0 + 4 * ty + j;
        if (col >= cols) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = g0 + 4 * tx + i;
            if (g < rows) out[(size_t)col * rows + g] = acc[j][i] + bias[g];
        }
    }
}
}  // namespace

extern "C" size_t wn_ar_cond_vars_floats(const wn_handle* h, int B, int Tn) {
    if (!h || h->cfg.kind != WN_KIND_TEACHER || B < 1 || Tn < 1) return 0;
    return (size_t)B * Tn * ((size_t)h->cfg.num_layers * h->cfg.gate_width + h->cfg.skip_width);
}

extern "C" int wn_ar_cond_vars(wn_handle* h, const float* enc, int B, int Tn, float* out, void* stream) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_ar_cond_vars: null handle");
    if (!h->finalized) return wn_fail(h, WN_ESTATE, "wn_ar_cond_vars: call wn_finalize first");
    if (h->cfg.kind != WN_KIND_TEACHER) return wn_fail(h, WN_EINVAL, "wn_ar_cond_vars: handle is not a Wavenet teacher");
    if (B < 1 || Tn < 1 || !enc || !out) return wn_fail(h, WN_EINVAL, "wn_ar_cond_vars: bad argument");
    const wn_config& c = h->cfg;
    const ArPack& P = h->ar;
    const int G = c.gate_width, S = c.skip_width, Cd = c.deconv_width, W = c.width;
    const int64_t cols = (int64_t)B * Tn;
    if ((cols + CV_T - 1) / CV_T > 65535) return wn_fail(h, WN_EINVAL, "wn_ar_cond_vars: B * Tn = %lld exceeds 4 194 240 columns per call", (long long)cols);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 gy((unsigned)1, (unsigned)((cols + CV_T - 1) / CV_T));
    float* dst = out;
    for (const ArLayerPack& lp : P.layers) {                    // 'mel_cond_%d' % (i + 1): [B, Tn, gate_width]
        hipLaunchKernelGGL(ar_cond_vars_kernel, dim3((G + CV_T - 1) / CV_T, gy.y), dim3(256), 0, st, h->d_blob + lp.wd_off,
                           3 * W + Cd, 3 * W, h->d_blob + lp.bc_off, enc, Cd, G, cols, dst);
        dst += (size_t)cols * G;
    }
    hipLaunchKernelGGL(ar_cond_vars_kernel, dim3((S + CV_T - 1) / CV_T, gy.y), dim3(256), 0, st, h->d_blob + P.wo1_off,   // 'mel_cond_out1'
                       S + Cd, S, h->d_blob + P.bco1_off, enc, Cd, S, cols, dst);
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

extern "C" int wn_ar_set_graph(wn_handle* h, int enable) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_ar_set_graph: null handle");
    WN_SWITCH(h, "wn_ar_set_graph");
    h->ar_use_graph = enable != 0;
    return WN_OK;
}

size_t wn_ar_workspace_bytes(const wn_handle* h, int B, int F) {
    const int64_t Tn = wn_ar_length(h, F);
    const size_t dec = WN_WS_HEAD + align_up((size_t)B * h->cfg.deconv_width * Tn * sizeof(float), 256) +
                       wn_deconv_scratch_bytes(h, B, F);
    const size_t ar = wn_ar_state_bytes(h, B);
    return dec > ar ? dec : ar;
}
