// Split-fp16 ("f16x3") variant of the IAF kernels: same math and same structure as
// wn_iaf.hip, but every contraction runs on v_mfma_f32_16x16x32_f16 as
//   a*b ~= ah*bh + ah*bl + al*bh      (hi = f16(x), lo = f16(x - hi), fp32 accumulate)
// which costs 3 fp16 MFMAs (3 x 16 cycles for K = 32) instead of 8 fp32 MFMAs (8 x 32
// cycles): 5.3x less matrix-pipe time at ~22-bit operand precision.  The kernels then stop
// being MFMA-bound and become bound by streaming `enc` (1 KB/sample/layer) -- the HBM
// roofline of BASELINE.md.
//
// Activations are stored as pair planes (wn_codec.h): l [B][2][32][LP+T] words,
// enc [B][2][128][TE] words -- the same bytes as the fp32 layout of wn_iaf.hip.
// K order inside a 32-channel MFMA K-step s: k-slot (kg = lane>>4, e) <-> channel
//   32*s + 16*(e>>2) + 4*kg + (e&3)
// so that the four operand words of a lane are pair rows 16*s + {0,1,8,9} + 2*kg, and so that
// the operand registers of the tap-t K-steps are exactly the accumulator-layout C-in of the
// residual 1x1 and the gated registers are its B operand (see wn_iaf.hip).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_pack_h.h"
#include "wn_mfma_h.h"

namespace {

// ---------------- start conv -> split l  (parallel_wavenet.py:222-225) ----------------
// One thread = one G4 word group (8 channels) at one time step: consecutive threads write
// consecutive 16-byte words of a group row (fully coalesced hi and lo streams).
__global__ __launch_bounds__(256) void iaf_start_h_kernel(const float* __restrict__ x, const float* __restrict__ wb,
                                                          unsigned* __restrict__ l, int64_t T, int XR, int64_t RS,
                                                          unsigned* __restrict__ status) {
    const int b = blockIdx.z, g = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float* xp = x + (size_t)b * XR + IAF_XP + t;
    const float x0 = xp[-3], x1 = xp[-2], x2 = xp[-1];
    const int s = g >> 2, kg = g & 3;
    wn_u4 hw, lw;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 2 * (16 * s + 8 * (i >> 1) + 2 * kg + (i & 1));      // even channel of the pair
        float o[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            o[hh] = wb[3 * IAF_W + c + hh] + wb[c + hh] * x0 + wb[IAF_W + c + hh] * x1 + wb[2 * IAF_W + c + hh] * x2;
        unsigned a, c2;
        wn_split_pair_t(o[0], o[1], a, c2, amax);
        hw[i] = a;
        lw[i] = c2;
    }
    wn_range_flag(amax, status);
    unsigned* base = l + (size_t)b * IAF_W * RS;
    *reinterpret_cast<wn_u4*>(base + ((size_t)g * RS + IAF_LP + t) * 4) = hw;
    *reinterpret_cast<wn_u4*>(base + ((size_t)(8 + g) * RS + IAF_LP + t) * 4) = lw;
}

// ---------------- fused residual layer ----------------
// FIRST (first layer of a flow, HN = 1, d = 1): the l operands are computed from the flow input x
// (start conv fused in), lin is not read.
template <int HN, bool FIRST = false>
__global__ __launch_bounds__(256, 1) void iaf_layer_h_kernel(
    const unsigned* __restrict__ lin, unsigned* __restrict__ lout, const unsigned* __restrict__ enc,
    const unsigned* __restrict__ wpack, int64_t RS, int64_t TE, int c0, int d, int tiles_per_row, int ntiles,
    const float* __restrict__ x, int XR, const float* __restrict__ wstart, unsigned* __restrict__ status) {
    static_assert(!FIRST || HN == 1, "the fused start conv is written for 64-sample tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    float amax = 0.f;                      // range guard of the split words this workgroup produces (wn_codec.h)
    constexpr int TILE = 64 * HN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(ldsw) + lane;          // [((s*4+mb)*2+plane)*64]
    const wn_u4* PRl = Pl + 14 * 4 * 2 * 64;
    const float* ldsf = reinterpret_cast<const float*>(ldsw);
    const float* bg = ldsf + IAF_P_FLOATS + IAF_PR_FLOATS + q * 16;
    const float* br = bg + 64;
    const int RS16 = (int)RS * 16, TE16 = (int)TE * 16;                     // bytes per group row
    const int lane_l = q * RS16 + (wave * 16 * HN + HN * n + IAF_LP) * 16;
    const int lane_e = q * TE16 + (wave * 16 * HN + HN * n + c0) * 16;

    auto tile_src = [&](int tile) -> HSrc {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * TILE;
        HSrc s;
        s.rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lin + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        s.re = __builtin_amdgcn_make_buffer_rsrc((void*)(enc + (size_t)b * IAF_CD * TE), 0, 0x7ffffff0, 0x00020000);
        s.vo[0] = lane_l + (tt - 2 * d) * 16;
        s.vo[1] = lane_l + (tt - d) * 16;
        s.vo[2] = lane_l + tt * 16;
        s.ve = lane_e + tt * 16;
        return s;
    };
    // K-steps 0-5: taps t-2d, t-d, t (two 32-channel steps each); 6-13: the 256 enc channels
    auto loadK = [&](const HSrc& s, int ks) -> KOp<HN> {
        KOp<HN> o;
#pragma unroll
        for (int e = 0; e < HN; ++e) {
            if (FIRST && ks < 6) {
                o.h[e] = o.l[e] = (wn_u4){0u, 0u, 0u, 0u};          // filled by first_layer_operands
            } else if (ks < 6) {
                o.h[e] = buf_ld4(s.rl, s.vo[ks >> 1] + 16 * e, (4 * (ks & 1)) * RS16);
                o.l[e] = buf_ld4(s.rl, s.vo[ks >> 1] + 16 * e, (8 + 4 * (ks & 1)) * RS16);
            } else {
                o.h[e] = buf_ld4<WN_ENC_AUX>(s.re, s.ve + 16 * e, (4 * (ks - 6)) * TE16);
                o.l[e] = buf_ld4<WN_ENC_AUX>(s.re, s.ve + 16 * e, (32 + 4 * (ks - 6)) * TE16);
            }
        }
        return o;
    };

    // One-tile-ahead operand prefetch into the registers that were just consumed (see
    // wn_iaf.hip); weights one K-step ahead from LDS into a register double buffer.
    KOp<HN> bc[14];
    const TileWalk tw = tile_walk(ntiles);
    const int tile0 = tw.first, tstep = tw.step, tend = tw.end;
    // FIRST: x[t-5 .. t-1] of this lane's column, one tile ahead like the other operands
    const f4* wq = reinterpret_cast<const f4*>(ldsw + IAF_LAYER_H_WORDS);
    float xv[5];
    auto load_x = [&](int tile) {
        const int b = tile / tiles_per_row;
        const int t = (tile - b * tiles_per_row) * TILE + wave * 16 + n;
        const float* xp = x + (size_t)b * XR + IAF_XP + t;
#pragma unroll
        for (int j = 0; j < 5; ++j) xv[j] = xp[j - 5];
    };
    if (tile0 < tend) {
        const HSrc s0 = tile_src(tile0);
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) bc[ks] = loadK(s0, ks);
        if (FIRST) load_x(tile0);
    }
    // the weight image is staged AFTER the first tile's operand loads are in flight
    if (FIRST) stage_start_weights(wstart, reinterpret_cast<f4*>(ldsw + IAF_LAYER_H_WORDS));
    stage_words<IAF_LAYER_H_WORDS>(wpack, ldsw);
    if (FIRST && tile0 < tend) {
        const int b0 = tile0 / tiles_per_row;
        KOp<1> f6[6];
        first_layer_operands(xv, (long long)(tile0 - b0 * tiles_per_row) * TILE + wave * 16 + n, q, wq, f6, amax);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) { bc[ks].h[0] = f6[ks].h[0]; bc[ks].l[0] = f6[ks].l[0]; }
    }
    const float inv_m = ldsf[IAF_P_FLOATS + IAF_PR_FLOATS + 128], inv_r = ldsf[IAF_P_FLOATS + IAF_PR_FLOATS + 129];
    for (int tile = tile0; tile < tend; tile += tstep) {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * TILE;
        const int next = tile + tstep;
        const bool has_next = next < tend;
        const HSrc sn = tile_src(has_next ? next : tile);
        if (FIRST && has_next) load_x(next);

        f4 acc[4][HN];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < HN; ++e) acc[mb][e] = (f4){0.f, 0.f, 0.f, 0.f};
        KOp<HN> cur[2];
        wn_u4 a[2][4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            a[0][mb][0] = Pl[((0 * 4 + mb) * 2 + 0) * 64];
            a[0][mb][1] = Pl[((0 * 4 + mb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            if (ks + 1 < 14) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    a[(ks + 1) & 1][mb][0] = Pl[(((ks + 1) * 4 + mb) * 2 + 0) * 64];
                    a[(ks + 1) & 1][mb][1] = Pl[(((ks + 1) * 4 + mb) * 2 + 1) * 64];
                }
            }
            if (ks == 4 || ks == 5) cur[ks - 4] = bc[ks];          // tap t: also the residual C-in
#pragma unroll
            for (int e = 0; e < HN; ++e)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    acc[mb][e] = mfma3(a[ks & 1][mb][0], a[ks & 1][mb][1], bc[ks].h[e], bc[ks].l[e], acc[mb][e]);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12 * HN, 0);
            if (has_next && !(FIRST && ks < 6)) bc[ks] = loadK(sn, ks);
            if (FIRST && ks == 5 && has_next) {
                // the l operands of the next tile, computed while the enc K-steps keep the MFMA pipe busy
                const int bn = next / tiles_per_row;
                KOp<1> f6[6];
                first_layer_operands(xv, (long long)(next - bn * tiles_per_row) * TILE + wave * 16 + n, q, wq, f6, amax);
#pragma unroll
                for (int k2 = 0; k2 < 6; ++k2) { bc[k2].h[0] = f6[k2].h[0]; bc[k2].l[0] = f6[k2].l[0]; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue per column tile: gate, residual 1x1, split, store
        const __amdgpu_buffer_rsrc_t ro =
            __builtin_amdgcn_make_buffer_rsrc((void*)(lout + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        const int vo_out = lane_l + tt * 16;
        wn_u4 ar[4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            ar[mb][0] = PRl[(mb * 2 + 0) * 64];
            ar[mb][1] = PRl[(mb * 2 + 1) * 64];
        }
#pragma unroll
        for (int e = 0; e < HN; ++e) {
            // gate: sigmoid(first half) * tanh(second half)  (parallel_wavenet.py:246-250)
            float g[2][4];
#pragma unroll
            for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    g[mg][r] = gate_scaled(fmaf(acc[mg][e][r], -WN_LOG2E * inv_m, bg[mg * 4 + r]),
                                           fmaf(acc[mg + 2][e][r], 2.f * WN_LOG2E * inv_m, bg[(mg + 2) * 4 + r]));
            wn_u4 gh, gl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned hw, lw;
                wn_split_pair(g[i >> 1][(i & 1) * 2], g[i >> 1][(i & 1) * 2 + 1], hw, lw);
                gh[i] = hw;
                gl[i] = lw;
            }
            wn_u4 oh[2], ol[2];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f4 rc = mfma3(ar[mb][0], ar[mb][1], gh, gl, (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    // l_old of channels 16mb+4q+2rp(+1): K-step 4+(mb>>1), slot 2(mb&1)+rp
                    float l0, l1;
                    wn_join_pair(cur[mb >> 1].h[e][(mb & 1) * 2 + rp], cur[mb >> 1].l[e][(mb & 1) * 2 + rp], l0, l1);
                    const float v0 = l0 + fmaf(rc[2 * rp], inv_r, br[mb * 4 + 2 * rp]);
                    const float v1 = l1 + fmaf(rc[2 * rp + 1], inv_r, br[mb * 4 + 2 * rp + 1]);
                    unsigned hw, lw;
                    wn_split_pair_t(v0, v1, hw, lw, amax);
                    oh[mb >> 1][(mb & 1) * 2 + rp] = hw;
                    ol[mb >> 1][(mb & 1) * 2 + rp] = lw;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                buf_st4<WN_G_ST_AUX>(oh[s2], ro, vo_out + 16 * e, (4 * s2) * RS16);
                buf_st4<WN_G_ST_AUX>(ol[s2], ro, vo_out + 16 * e, (8 + 4 * s2) * RS16);
            }
        }
    }
    wn_range_flag(amax, status);
}

// ---------------- flow head (parallel_wavenet.py:256-277, :319-324) ----------------
template <int HN>
__global__ __launch_bounds__(256, 1) void iaf_head_h_kernel(
    const unsigned* __restrict__ lin, const unsigned* __restrict__ enc, const unsigned* __restrict__ wpack,
    float* __restrict__ x, float* __restrict__ Mt, float* __restrict__ St,
    int64_t RS, int64_t TE, int c0, int XR, int64_t T, int first, int tiles_per_row, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    constexpr int TILE = 64 * HN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(ldsw) + lane;
    const float* ldsf = reinterpret_cast<const float*>(ldsw);
    const float* bo = ldsf + IAF_PH_FLOATS + q * 16;
    const float* wm = bo + 64;
    const float* wsc = wm + 64;
    const int RS16 = (int)RS * 16, TE16 = (int)TE * 16;
    const int lane_l = q * RS16 + (wave * 16 * HN + HN * n + IAF_LP) * 16;
    const int lane_e = q * TE16 + (wave * 16 * HN + HN * n + c0) * 16;

    auto tile_src = [&](int tile) -> HSrc {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * TILE;
        HSrc s;
        s.rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lin + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        s.re = __builtin_amdgcn_make_buffer_rsrc((void*)(enc + (size_t)b * IAF_CD * TE), 0, 0x7ffffff0, 0x00020000);
        s.vo[0] = s.vo[1] = s.vo[2] = lane_l + tt * 16;
        s.ve = lane_e + tt * 16;
        return s;
    };
    // K-steps 0-1: out1 over relu(l); 2-9: mel_cond_out1 over the 256 enc channels
    auto loadK = [&](const HSrc& s, int ks) -> KOp<HN> {
        KOp<HN> o;
#pragma unroll
        for (int e = 0; e < HN; ++e) {
            if (ks < 2) {
                o.h[e] = buf_ld4(s.rl, s.vo[2] + 16 * e, (4 * ks) * RS16);
                o.l[e] = buf_ld4(s.rl, s.vo[2] + 16 * e, (8 + 4 * ks) * RS16);
            } else {
                o.h[e] = buf_ld4<WN_ENC_AUX>(s.re, s.ve + 16 * e, (4 * (ks - 2)) * TE16);
                o.l[e] = buf_ld4<WN_ENC_AUX>(s.re, s.ve + 16 * e, (32 + 4 * (ks - 2)) * TE16);
            }
        }
        return o;
    };
    KOp<HN> bc[10];
    const TileWalk tw = tile_walk(ntiles);
    const int tile0 = tw.first, tstep = tw.step, tend = tw.end;
    if (tile0 < tend) {
        const HSrc s0 = tile_src(tile0);
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) bc[ks] = loadK(s0, ks);
    }
    stage_words<IAF_HEAD_FLOATS>(wpack, ldsw);
    const float bmean = ldsf[IAF_PH_FLOATS + 192], bscale = ldsf[IAF_PH_FLOATS + 193];
    const float inv_m = ldsf[IAF_PH_FLOATS + 194];
    for (int tile = tile0; tile < tend; tile += tstep) {
        const int b = tile / tiles_per_row;
        const int t0 = (tile - b * tiles_per_row) * TILE + wave * 16 * HN;
        const int next = tile + tstep;
        const bool has_next = next < tend;
        const HSrc sn = tile_src(has_next ? next : tile);
        f4 acc[4][HN];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < HN; ++e) acc[mb][e] = (f4){0.f, 0.f, 0.f, 0.f};
        wn_u4 a[2][4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            a[0][mb][0] = Pl[((0 * 4 + mb) * 2 + 0) * 64];
            a[0][mb][1] = Pl[((0 * 4 + mb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            if (ks + 1 < 10) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    a[(ks + 1) & 1][mb][0] = Pl[(((ks + 1) * 4 + mb) * 2 + 0) * 64];
                    a[(ks + 1) & 1][mb][1] = Pl[(((ks + 1) * 4 + mb) * 2 + 1) * 64];
                }
            }
#pragma unroll
            for (int e = 0; e < HN; ++e) {
                wn_u4 bh = bc[ks].h[e], bl = bc[ks].l[e];
                if (ks < 2) {                         // relu(l) (:256) on the reconstructed value
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v0, v1;
                        wn_join_pair(bh[i], bl[i], v0, v1);
                        unsigned hw, lw;
                        wn_split_pair(fmaxf(v0, 0.f), fmaxf(v1, 0.f), hw, lw);
                        bh[i] = hw;
                        bl[i] = lw;
                    }
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    acc[mb][e] = mfma3(a[ks & 1][mb][0], a[ks & 1][mb][1], bh, bl, acc[mb][e]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            if (has_next) bc[ks] = loadK(sn, ks);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < HN; ++e) {
            float pm = 0.f, ps = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = fmaxf(fmaf(acc[mb][e][r], inv_m, bo[mb * 4 + r]), 0.f);
                    pm = fmaf(wm[mb * 4 + r], o, pm);
                    ps = fmaf(wsc[mb * 4 + r], o, ps);
                }
            pm += __shfl_xor(pm, 16);
            ps += __shfl_xor(ps, 16);
            pm += __shfl_xor(pm, 32);
            ps += __shfl_xor(ps, 32);
            if (q == 0) {
                const int64_t t = t0 + HN * n + e;
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
}

}  // namespace

// ---------------------------------------------------------------------------
int wn_pack_iaf_h(wn_handle* h, std::vector<float>& blob) {
    const wn_config& c = h->cfg;
    auto var = [&](const std::string& nme) -> const std::vector<float>& { return h->vars.at(nme).data; };
    unsigned layer_id = 0;
    for (int k = 0; k < c.n_flows; ++k) {
        const std::string p = "iaf_" + std::to_string(k + 1);
        IafFlowPack& fp = h->flows[k];
        for (int i = 0; i < c.iaf_layers[k]; ++i) {
            const std::string s = std::to_string(i + 1);
            std::vector<float> Wd = wn_get_kernel(h, p + "/dilated_conv_" + s, "W", false);   // [3][64][64]
            std::vector<float> Wc = wn_get_kernel(h, p + "/mel_cond_" + s, "W", false);       // [256][64]
            std::vector<float> Wr = wn_get_kernel(h, p + "/res_" + s, "W", false);            // [32][64]
            const auto& bd = var(p + "/dilated_conv_" + s + "/biases");
            const auto& bc = var(p + "/mel_cond_" + s + "/biases");
            const auto& br = var(p + "/res_" + s + "/biases");
            blob.resize(align_up(blob.size(), 64));
            fp.layers[i].off_h = blob.size();
            blob.resize(blob.size() + IAF_LAYER_H_WORDS);
            unsigned* P = reinterpret_cast<unsigned*>(blob.data() + fp.layers[i].off_h);
            float* tb = blob.data() + fp.layers[i].off_h + IAF_P_FLOATS + IAF_PR_FLOATS;
            const float sm = std::min(pick_scale(Wd.data(), Wd.size()), pick_scale(Wc.data(), Wc.size()));
            const float sr = pick_scale(Wr.data(), Wr.size());
            for (int ks = 0; ks < 14; ++ks)
                for (int mb = 0; mb < 4; ++mb)
                    pack_afrag(P + ((size_t)(ks * 4 + mb) * 2) * 256, [&](int e, int kg, int i16) {
                        const int o = 16 * mb + i16;
                        const int ch = 16 * (e >> 2) + 4 * kg + (e & 3);
                        if (ks < 6) return sm * Wd[((size_t)(ks >> 1) * 64 + 32 * (ks & 1) + ch) * 64 + o];
                        return sm * Wc[((size_t)32 * (ks - 6) + ch) * 64 + o];
                    });
            for (int mb = 0; mb < 4; ++mb)
                pack_afrag(P + IAF_P_FLOATS + (size_t)(mb * 2) * 256, [&](int e, int kg, int i16) {
                    const int ch = 16 * (e >> 2) + 4 * kg + (e & 3);
                    return sr * Wr[(size_t)ch * 64 + 16 * mb + i16];
                });
            for (int q = 0; q < 4; ++q)
                for (int mb = 0; mb < 4; ++mb)
                    for (int r = 0; r < 4; ++r) {
                        const int o = 16 * mb + 4 * q + r;
                        // gate bias, pre-scaled for gate_scaled(): sigmoid rows (o < 32) by -log2(e), tanh rows by 2 log2(e)
                        tb[q * 16 + mb * 4 + r] = (bd[o] + bc[o]) * (mb < 2 ? -1.4426950408889634f : 2.8853900817779268f);
                        tb[64 + q * 16 + mb * 4 + r] = br[o];
                    }
            tb[128] = 1.0f / sm;
            tb[129] = 1.0f / sr;
            tb[130] = tb[131] = 0.f;
            fp.layers[i].id = ++layer_id;
            for (int m = 0; m < 4; ++m) memcpy(&tb[132 + m], &fp.layers[i].id, 4);
        }
        {
            std::vector<float> Wo = wn_get_kernel(h, p + "/out1", "W", false);            // [64][64]
            std::vector<float> Wco = wn_get_kernel(h, p + "/mel_cond_out1", "W", false);  // [256][64]
            std::vector<float> Wm = wn_get_kernel(h, p + "/out2_mean", "W", false);
            std::vector<float> Ws = wn_get_kernel(h, p + "/out2_scale", "W", false);
            const auto& bo = var(p + "/out1/biases");
            const auto& bco = var(p + "/mel_cond_out1/biases");
            blob.resize(align_up(blob.size(), 64));
            fp.head_off_h = blob.size();
            blob.resize(blob.size() + IAF_HEAD_FLOATS);
            unsigned* P = reinterpret_cast<unsigned*>(blob.data() + fp.head_off_h);
            float* tb = blob.data() + fp.head_off_h + IAF_PH_FLOATS;
            const float sm = std::min(pick_scale(Wo.data(), Wo.size()), pick_scale(Wco.data(), Wco.size()));
            for (int ks = 0; ks < 10; ++ks)
                for (int mb = 0; mb < 4; ++mb)
                    pack_afrag(P + ((size_t)(ks * 4 + mb) * 2) * 256, [&](int e, int kg, int i16) {
                        const int o = 16 * mb + i16;
                        const int ch = 16 * (e >> 2) + 4 * kg + (e & 3);
                        if (ks < 2) return sm * Wo[((size_t)32 * ks + ch) * 64 + o];
                        return sm * Wco[((size_t)32 * (ks - 2) + ch) * 64 + o];
                    });
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
            tb[194] = 1.0f / sm;
            tb[195] = 0.f;
        }
    }
    // row-block table of the hoisted conditioning GEMM: the cond K-steps of each pack
    // (layer: K-steps 6-13, head: 2-9) are contiguous 8 x 4 x 2 x 256 words
    {
        std::vector<unsigned> tab;
        for (int k = 0; k < c.n_flows; ++k) {
            IafFlowPack& fp = h->flows[k];
            fp.rb_base = (int)tab.size();
            for (const IafLayerPack& lp : fp.layers) tab.push_back((unsigned)(lp.off_h + 6 * 2048));
            tab.push_back((unsigned)(fp.head_off_h + 2 * 2048));
        }
        if (blob.size() >= 0xffff0000u) return wn_fail(h, WN_EINVAL, "weight blob too large");
        blob.resize(align_up(blob.size(), 64));
        h->cond_tab_off = blob.size();
        h->cond_rows = (int)tab.size();
        blob.resize(blob.size() + align_up(tab.size(), 4));
        memcpy(blob.data() + h->cond_tab_off, tab.data(), tab.size() * sizeof(unsigned));
    }
    // launch plan of the layer-group kernel (wn_iaf_g.hip) and the row-block orders the conditioning GEMM needs
    // for it: natural row blocks first, then the ones a decimated group (or its flow head) consumes
    {
        h->groups_ok = true;
        for (IafFlowPack& fp : h->flows) {
            std::vector<int> dil;
            for (const IafLayerPack& lp : fp.layers) dil.push_back(lp.dilation);
            if (!wn_iaf_g_plan(dil, fp.groups)) { fp.groups.clear(); h->groups_ok = false; }
        }
        if (!h->groups_ok) for (IafFlowPack& fp : h->flows) fp.groups.clear();
        const int R = h->cond_rows;
        std::vector<unsigned> id(R), all, flw(R);
        std::vector<unsigned> all_dec;
        h->n_nat_flow.assign(c.n_flows, 0);
        for (int i = 0; i < R; ++i) id[i] = (unsigned)i;
        for (int k = 0; k < c.n_flows; ++k) {
            const IafFlowPack& fp = h->flows[k];
            const int nl = (int)fp.layers.size();
            std::vector<int> kind(nl + 1, 0);
            for (const WnGroup& g : fp.groups)
                for (int i = g.begin; i < g.end; ++i) kind[i] = g.kind;
            kind[nl] = fp.groups.empty() ? 0 : fp.groups.back().kind;      // the head runs inside the flow's last group
            std::vector<unsigned> loc_nat, loc_dec;
            for (int i = 0; i <= nl; ++i) (kind[i] ? loc_dec : loc_nat).push_back((unsigned)i);
            h->n_nat_flow[k] = (int)loc_nat.size();
            for (unsigned v : loc_nat) all.push_back(fp.rb_base + v);
            for (unsigned v : loc_dec) all_dec.push_back(fp.rb_base + v);
            size_t o = fp.rb_base;
            for (unsigned v : loc_nat) flw[o++] = v;
            for (unsigned v : loc_dec) flw[o++] = v;
        }
        h->n_nat_all = (int)all.size();
        all.insert(all.end(), all_dec.begin(), all_dec.end());
        auto put = [&](const std::vector<unsigned>& t) {
            blob.resize(align_up(blob.size(), 64));
            const size_t off = blob.size();
            blob.resize(blob.size() + align_up(t.size(), 4));
            memcpy(blob.data() + off, t.data(), t.size() * sizeof(unsigned));
            return off;
        };
        h->order_id_off = put(id);
        h->order_all_off = put(all);
        h->order_flow_off = put(flw);
    }
    return WN_OK;
}

int wn_iaf_h_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_h_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_LAYER_H_WORDS * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_h_kernel<1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (IAF_LAYER_H_WORDS + IAF_START_LDS_WORDS) * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_h_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_LAYER_H_WORDS * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_head_h_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_HEAD_FLOATS * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_head_h_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, IAF_HEAD_FLOATS * 4));
    return WN_OK;
}

void wn_iaf_h_start(const float* x, const float* wb, float* l, int64_t T, int XR, int64_t RS, int B, hipStream_t st,
                    unsigned* status) {
    dim3 g((unsigned)((T + 255) / 256), 8, B);
    hipLaunchKernelGGL(iaf_start_h_kernel, g, dim3(256), 0, st, x, wb, reinterpret_cast<unsigned*>(l), T, XR, RS, status);
}

// 64- or 128-sample workgroup tiles: whichever leaves the last round of the persistent grid
// fuller (B*T/64 tiles over num_cu workgroups); 128 halves the LDS weight traffic per sample.
static int pick_hn(int B, int64_t T, int num_cu) {
    const int64_t n1 = B * (T / 64), n2 = B * (T / 128);
    const int64_t c1 = (n1 + num_cu - 1) / num_cu, c2 = 2 * ((n2 + num_cu - 1) / num_cu);
    const char* force = getenv("WN_HN");
    if (force) return atoi(force) == 1 ? 1 : 2;
    return c1 < c2 ? 1 : 2;
}

// x != nullptr: first layer of a flow -- the start conv is evaluated inside the kernel from the flow
// input x (row stride XR) with the start weights wstart, lin is not read
void wn_iaf_h_layer(const float* lin, float* lout, const float* enc, const float* wpack, int64_t RS, int64_t TE,
                    int c0, int d, int B, int64_t T, int num_cu, hipStream_t st, unsigned* status, const float* x, int XR,
                    const float* wstart) {
    const int hn = x ? 1 : pick_hn(B, T, num_cu);
    const int tiles_per_row = (int)(T / (64 * hn)), ntiles = B * tiles_per_row;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    const unsigned* li = reinterpret_cast<const unsigned*>(lin);
    unsigned* lo = reinterpret_cast<unsigned*>(lout);
    const unsigned* en = reinterpret_cast<const unsigned*>(enc);
    const unsigned* wp = reinterpret_cast<const unsigned*>(wpack);
    if (x)
        hipLaunchKernelGGL((iaf_layer_h_kernel<1, true>), dim3(grid), dim3(256),
                           (IAF_LAYER_H_WORDS + IAF_START_LDS_WORDS) * 4, st, li, lo, en, wp, RS, TE, c0, d, tiles_per_row,
                           ntiles, x, XR, wstart, status);
    else if (hn == 1)
        hipLaunchKernelGGL((iaf_layer_h_kernel<1, false>), dim3(grid), dim3(256), IAF_LAYER_H_WORDS * 4, st, li, lo, en, wp,
                           RS, TE, c0, d, tiles_per_row, ntiles, x, XR, wstart, status);
    else
        hipLaunchKernelGGL((iaf_layer_h_kernel<2, false>), dim3(grid), dim3(256), IAF_LAYER_H_WORDS * 4, st, li, lo, en, wp,
                           RS, TE, c0, d, tiles_per_row, ntiles, x, XR, wstart, status);
}

void wn_iaf_h_head(const float* lin, const float* enc, const float* wpack, float* x, float* Mt, float* St,
                   int64_t RS, int64_t TE, int c0, int XR, int64_t T, int first, int B, int num_cu, hipStream_t st) {
    const int hn = pick_hn(B, T, num_cu);
    const int tiles_per_row = (int)(T / (64 * hn)), ntiles = B * tiles_per_row;
    const int grid = ntiles < num_cu ? ntiles : num_cu;
    auto kern = hn == 1 ? iaf_head_h_kernel<1> : iaf_head_h_kernel<2>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), IAF_HEAD_FLOATS * 4, st,
                       reinterpret_cast<const unsigned*>(lin), reinterpret_cast<const unsigned*>(enc),
                       reinterpret_cast<const unsigned*>(wpack), x, Mt, St, RS, TE, c0, XR, T, first, tiles_per_row,
                       ntiles);
}
