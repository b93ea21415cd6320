// "Flow pipeline": every residual layer and every flow head of the student as ONE persistent launch.
//
//   wavenet/parallel_wavenet.py:227-254 (residual layers), :222-225 (start conv), :256-287 (head),
//   :289-345 (feed_forward: the four flows chained on new_x).
//
// One launch per layer (wn_iaf_h.hip / wn_iaf_c.hip) makes a single 4.8 s utterance a chain of ~60
// dependent launches, each with only ~4.7 tiles per CU: launch floors and tile-round quantisation,
// not bytes, set the time (0.40 of the HBM roofline at one utterance).  But the network is CAUSAL in
// time: column t of layer i+1 needs columns <= t of layer i, column t of flow k+1 needs columns < t of
// flow k.  So the whole student is a systolic pipeline over 64-sample time tiles:
//
//   * stage = one residual layer (start conv folded into a flow's first layer) or one flow head;
//     the 256 workgroups of the launch (one per CU) are split evenly over the stages
//     (parallel_wavenet.json: 60 layers + 4 heads = 64 stages x 4 workgroups);
//   * the workgroups of a stage walk the time tiles in order (workgroup j takes tiles j, j+n, ...);
//     each of its 4 waves owns 16 columns of the tile, exactly as in iaf_layer_h_kernel, and is an
//     independent worker: no workgroup barrier after the weights are staged in LDS;
//   * every stage writes its OWN full-length output buffer (write-once: an address is never
//     rewritten inside a launch), with write-through `sc1` stores; a wave publishes "my first c tiles
//     are complete" in a per-wave progress word after its stores were acknowledged
//     (s_waitcnt vmcnt -> sc1 flag store: the R1 hand-off of the CDNA4 guide, G16);
//   * a consumer wave polls the 4 x n progress words of the producer stage with ONE sc1 load (lane i
//     reads word i), decides with a ballot, and reads the tile with sc1 loads (served by L2, never by
//     the CU's own L1).  The taps t-d, t-2d are L2 hits: the workgroups of a stage -- and eight
//     consecutive stages -- sit on the same XCD;
//   * the operands of the NEXT tile are prefetched during the K loop of the current one when the
//     poll of the previous iteration already showed them complete (steady state: the producer is
//     several tiles ahead); otherwise the wave finishes its tile, publishes, and spins.
//     Every spin is bounded; a timeout raises the error word and the output is poisoned with NaN.
//
// The conditioning 1x1 runs inside every layer (fused form: 14 K-steps): `enc` is read from HBM once
// per XCD instead of once per layer, there is no 1.26 GB projected-term workspace, and the whole
// launch is bound by the fp16 matrix pipe (3 x 61 440 FLOP per sample and layer), not by HBM.
#include <algorithm>
#include <cstdlib>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_mfma_h.h"

namespace {

constexpr int SC1 = 16;                       // aux bit of buffer loads/stores: agent scope, write-through
constexpr int OOB = (int)0x80000000;          // voffset bit that pushes an access past every descriptor
constexpr unsigned SPIN_LIMIT = 1u << 22;     // polls of ~0.3 us each before a wait gives up
constexpr int PIPE_MAX_WG = 16;
#ifndef WN_PIPE_TAP_AUX
#define WN_PIPE_TAP_AUX 0
#endif
constexpr int TAP_AUX = WN_PIPE_TAP_AUX;      // cache policy of the tap loads t-d, t-2d (0 = plain, 16 = sc1)               // workgroups per stage (4 progress words each, one per lane)

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ inline rsrc_t mk_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ inline unsigned ld1_sc1(rsrc_t r, int voff) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, SC1);
}
__device__ inline float ldf_sc1(rsrc_t r, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, SC1));
}
__device__ inline void st1_sc1(unsigned v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, 0, SC1);
}
__device__ inline void stf_sc1(float v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, 0, SC1);
}
__device__ inline void st4_sc1(wn_u4 v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, SC1);
}

struct PipeArgs {
    const float* blob;
    const int* stages;             // 8 ints per stage: type, d, w_off, ws_off, lin, lout, flow, enc
    unsigned* lbuf;                // one activation buffer per layer, G4 words
    long long lbuf_words;          //   words per buffer (batch x 64 x RS)
    const unsigned* enc;           // one upsampled-mel image per deconv stack, G4 words
    long long enc_words;
    float* X;                      // flow inputs/outputs X[0 .. n_flows]: [batch][XR]
    long long x_floats;
    float* M;                      // running mean_tot / scale_tot after flow k: [batch][T]
    float* S;
    long long ms_floats;
    unsigned* cnt;                 // progress words [stage][64], then the error word
    int cnt_words;
    long long RS, TE, T;
    int c0, XR, tiles_per_row, ntiles, s0, nst;
};

// what a wave knows about the stage it consumes from
struct PipeDep {
    rsrc_t rc;          // the producer's 64 progress words (an empty descriptor when there is none)
    int nwg;            // producer workgroups
    int jw;             // the producer workgroup lane i watches (i >> 2)
    bool valid;         // lane < 4 * nwg
    rsrc_t rerr;
};

// c = tiles completed by the producer wave this lane watches; its tiles are jw, jw + nwg, ...
__device__ inline bool dep_ready(unsigned c, const PipeDep& d, int tile) {
    const int next_open = d.valid ? d.jw + (int)c * d.nwg : 0x7fffffff;
    return __builtin_amdgcn_ballot_w64(next_open <= tile) == 0;
}
__device__ inline unsigned dep_poll(const PipeDep& d, int lane) { return ld1_sc1(d.rc, lane * 4); }

// blocking wait until every tile <= `tile` of the producer stage is complete; false = gave up
__device__ inline bool dep_wait(const PipeDep& d, int tile, int lane, int code) {
    for (unsigned spins = 0;; ++spins) {
        const unsigned c = dep_poll(d, lane);
        if (dep_ready(c, d, tile)) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((spins & 255u) == 255u) {
            if (__builtin_amdgcn_readfirstlane(ld1_sc1(d.rerr, 0)) != 0) return false;
            if (spins > SPIN_LIMIT) {
                st1_sc1((unsigned)code, d.rerr, lane == 0 ? 0 : OOB);
                return false;
            }
        }
    }
}

// publish "this wave has completed `done` tiles": lane 0 only (the others store out of range)
__device__ inline void publish(unsigned done, rsrc_t rown, int own_off) { st1_sc1(done, rown, own_off); }

// K-step at which a wave publishes the PREVIOUS tile, and a LOWER bound of the VMEM operations it has
// issued since that tile's last store by then (s_waitcnt vmcnt counts loads and stores in issue order, so
// "at most N outstanding" with N <= that number means the stores are acknowledged).  Only the 16-byte
// operand loads are counted -- two per K-step, never merged by the compiler and pinned in their K-step by
// the memory-clobbering asm; the progress poll and the 4-byte x / mean / scale loads (which the compiler
// does merge) come on top.  First layer of a flow: no operand loads before K-step 6.
constexpr int KPUB = 10;
constexpr int NPUB_LAYER = 2 * (KPUB + 1);
constexpr int NPUB_FIRST = 2 * (KPUB + 1 - 6);
constexpr int KPUB_HEAD = 7;
constexpr int NPUB_HEAD = 2 * (KPUB_HEAD + 1);

#define WN_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

struct PipeCtx {
    int lane, wave, j, nwg, own_off;
    rsrc_t rown;
    PipeDep dep;
};

// ---------------- residual layer stage ----------------
template <bool FIRST>
__device__ __forceinline__ void pipe_layer(const PipeArgs& A, const int* __restrict__ stg, unsigned* ldsw,
                                           const PipeCtx& cx) {
    const int lane = cx.lane, wave = cx.wave;
    const int n = lane & 15, q = lane >> 4;
    const int d = stg[1];
    const unsigned* wpack = reinterpret_cast<const unsigned*>(A.blob + (unsigned)stg[2]);
    const unsigned* lin = FIRST ? nullptr : A.lbuf + (size_t)stg[4] * A.lbuf_words;
    unsigned* lout = A.lbuf + (size_t)stg[5] * A.lbuf_words;
    const float* xin = A.X + (size_t)stg[6] * A.x_floats;
    const unsigned* enc = A.enc + (size_t)stg[7] * A.enc_words;
    const int64_t RS = A.RS, TE = A.TE;
    const int tpr = A.tiles_per_row, ntiles = A.ntiles, XR = A.XR;

    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(ldsw) + lane;          // [((s*4+mb)*2+plane)*64]
    const wn_u4* PRl = Pl + 14 * 4 * 2 * 64;
    const float* ldsf = reinterpret_cast<const float*>(ldsw);
    const float* bg = ldsf + IAF_P_FLOATS + IAF_PR_FLOATS + q * 16;
    const float* br = bg + 64;
    const f4* wq = reinterpret_cast<const f4*>(ldsw + IAF_LAYER_H_WORDS);
    const int RS16 = (int)RS * 16, TE16 = (int)TE * 16;                     // bytes per group row
    const int lane_l = q * RS16 + (wave * 16 + n + IAF_LP) * 16;
    const int lane_e = q * TE16 + (wave * 16 + n + A.c0) * 16;

    if (FIRST) stage_start_weights(A.blob + (unsigned)stg[3], reinterpret_cast<f4*>(ldsw + IAF_LAYER_H_WORDS));
    stage_words<IAF_LAYER_H_WORDS>(wpack, ldsw);
    const float inv_m = ldsf[IAF_P_FLOATS + IAF_PR_FLOATS + 128], inv_r = ldsf[IAF_P_FLOATS + IAF_PR_FLOATS + 129];

    struct Src {
        rsrc_t rl, re, rx;
        int vo[3], ve, vx;
    };
    auto tile_src = [&](int tile, int pred) -> Src {
        const int b = tile / tpr;
        const int tt = (tile - b * tpr) * 64;
        Src s;
        if (!FIRST) s.rl = mk_rsrc(lin + (size_t)b * IAF_W * RS, IAF_W * (int)RS * 4);
        s.re = mk_rsrc(enc + (size_t)b * IAF_CD * TE, IAF_CD * (int)TE * 4);
        if (FIRST) s.rx = mk_rsrc(xin + (size_t)b * XR, XR * 4);
        s.vo[0] = (lane_l + (tt - 2 * d) * 16) | pred;
        s.vo[1] = (lane_l + (tt - d) * 16) | pred;
        s.vo[2] = (lane_l + tt * 16) | pred;
        s.ve = (lane_e + tt * 16) | pred;
        s.vx = ((IAF_XP + tt + wave * 16 + n - 5) * 4) | pred;
        return s;
    };
    // K-steps 0-5: taps t-2d, t-d, t (two 32-channel steps each); 6-13: the 256 enc channels
    auto loadK = [&](const Src& s, int ks) -> KOp<1> {
        KOp<1> o;
        if (FIRST && ks < 6) {
            o.h[0] = o.l[0] = (wn_u4){0u, 0u, 0u, 0u};                      // filled by first_layer_operands
        } else if (ks < 4) {
            // taps t-2d, t-d: lines a workgroup of this stage (same XCD) fetched before as tap t -> L2 hits.
            // Plain loads are safe here: the buffers are write-once, a 128-byte line never straddles two
            // tiles and no wave touches a line before its tile was published, so no cache holds an old copy.
            o.h[0] = buf_ld4<TAP_AUX>(s.rl, s.vo[ks >> 1], (4 * (ks & 1)) * RS16);
            o.l[0] = buf_ld4<TAP_AUX>(s.rl, s.vo[ks >> 1], (8 + 4 * (ks & 1)) * RS16);
        } else if (ks < 6) {
            o.h[0] = buf_ld4<SC1>(s.rl, s.vo[ks >> 1], (4 * (ks & 1)) * RS16);
            o.l[0] = buf_ld4<SC1>(s.rl, s.vo[ks >> 1], (8 + 4 * (ks & 1)) * RS16);
        } else {
            o.h[0] = buf_ld4(s.re, s.ve, (4 * (ks - 6)) * TE16);
            o.l[0] = buf_ld4(s.re, s.ve, (32 + 4 * (ks - 6)) * TE16);
        }
        return o;
    };
    float xv[5];
    auto load_x = [&](const Src& s) {
#pragma unroll
        for (int jx = 0; jx < 5; ++jx) xv[jx] = ldf_sc1(s.rx, s.vx + 4 * jx);
    };
    auto tile_t = [&](int tile) -> long long {
        const int b = tile / tpr;
        return (long long)(tile - b * tpr) * 64 + wave * 16 + n;
    };
    KOp<1> bc[14];
    auto load_all = [&](int tile) {
        const Src s0 = tile_src(tile, 0);
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) bc[ks] = loadK(s0, ks);
        if (FIRST) {
            load_x(s0);
            KOp<1> f6[6];
            first_layer_operands(xv, tile_t(tile), q, wq, f6);
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) { bc[ks].h[0] = f6[ks].h[0]; bc[ks].l[0] = f6[ks].l[0]; }
        }
    };

    int tile = cx.j;
    if (!dep_wait(cx.dep, tile, lane, 0x100 + A.s0)) return;
    load_all(tile);
    unsigned c_async = dep_poll(cx.dep, lane);
    unsigned done = 0;
    for (;;) {
        const int b = tile / tpr;
        const int tt = (tile - b * tpr) * 64;
        const int next = tile + cx.nwg;
        const bool has_next = next < ntiles;
        // the poll issued one tile ago decides whether the next tile's operands can be prefetched now
        const bool rdy = has_next && dep_ready(c_async, cx.dep, next);
        const Src sn = tile_src(has_next ? next : tile, rdy ? 0 : OOB);
        c_async = dep_poll(cx.dep, lane);
        if (FIRST) load_x(sn);

        f4 acc[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = (f4){0.f, 0.f, 0.f, 0.f};
        KOp<1> cur[2];
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
            for (int mb = 0; mb < 4; ++mb)
                acc[mb] = mfma3(a[ks & 1][mb][0], a[ks & 1][mb][1], bc[ks].h[0], bc[ks].l[0], acc[mb]);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            if (!(FIRST && ks < 6)) bc[ks] = loadK(sn, ks);
            if (FIRST && ks == 5 && rdy) {
                KOp<1> f6[6];
                first_layer_operands(xv, tile_t(next), q, wq, f6);
#pragma unroll
                for (int k2 = 0; k2 < 6; ++k2) { bc[k2].h[0] = f6[k2].h[0]; bc[k2].l[0] = f6[k2].l[0]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks == KPUB) {
                // the stores of the previous tile are older than the last NPUB operations: acknowledged
                WN_WAIT_VM(FIRST ? NPUB_FIRST : NPUB_LAYER);
                publish(done, cx.rown, cx.own_off);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: gate, residual 1x1, split, store (parallel_wavenet.py:246-254)
        const rsrc_t ro = mk_rsrc(lout + (size_t)b * IAF_W * RS, IAF_W * (int)RS * 4);
        const int vo_out = lane_l + tt * 16;
        wn_u4 ar[4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            ar[mb][0] = PRl[(mb * 2 + 0) * 64];
            ar[mb][1] = PRl[(mb * 2 + 1) * 64];
        }
        {
            float g[2][4];
#pragma unroll
            for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    g[mg][r] = sigmoidf_(fmaf(acc[mg][r], inv_m, bg[mg * 4 + r])) *
                               tanhf_(fmaf(acc[mg + 2][r], inv_m, bg[(mg + 2) * 4 + r]));
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
                    float l0, l1;
                    wn_join_pair(cur[mb >> 1].h[0][(mb & 1) * 2 + rp], cur[mb >> 1].l[0][(mb & 1) * 2 + rp], l0, l1);
                    const float v0 = l0 + fmaf(rc[2 * rp], inv_r, br[mb * 4 + 2 * rp]);
                    const float v1 = l1 + fmaf(rc[2 * rp + 1], inv_r, br[mb * 4 + 2 * rp + 1]);
                    unsigned hw, lw;
                    wn_split_pair(v0, v1, hw, lw);
                    oh[mb >> 1][(mb & 1) * 2 + rp] = hw;
                    ol[mb >> 1][(mb & 1) * 2 + rp] = lw;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                st4_sc1(oh[s2], ro, vo_out, (4 * s2) * RS16);
                st4_sc1(ol[s2], ro, vo_out, (8 + 4 * s2) * RS16);
            }
        }
        ++done;
        if (!rdy) {
            // nothing of the next tile is in flight: publish at once, then wait for the producer
            WN_WAIT_VM(0);
            publish(done, cx.rown, cx.own_off);
            if (!has_next) return;
            if (!dep_wait(cx.dep, next, lane, 0x200 + A.s0)) return;
            load_all(next);
            c_async = dep_poll(cx.dep, lane);
        }
        tile = next;
    }
}

// ---------------- flow head stage (parallel_wavenet.py:256-277, :319-324) ----------------
__device__ __forceinline__ void pipe_head(const PipeArgs& A, const int* __restrict__ stg, unsigned* ldsw,
                                          const PipeCtx& cx) {
    const int lane = cx.lane, wave = cx.wave;
    const int n = lane & 15, q = lane >> 4;
    const unsigned* wpack = reinterpret_cast<const unsigned*>(A.blob + (unsigned)stg[2]);
    const unsigned* lin = A.lbuf + (size_t)stg[4] * A.lbuf_words;
    const int flow = stg[6];
    const float* xin = A.X + (size_t)flow * A.x_floats;
    float* xout = A.X + (size_t)(flow + 1) * A.x_floats;
    const float* Min = A.M + (size_t)(flow > 0 ? flow - 1 : 0) * A.ms_floats;
    const float* Sin = A.S + (size_t)(flow > 0 ? flow - 1 : 0) * A.ms_floats;
    float* Mout = A.M + (size_t)flow * A.ms_floats;
    float* Sout = A.S + (size_t)flow * A.ms_floats;
    const unsigned* enc = A.enc + (size_t)stg[7] * A.enc_words;
    const int64_t RS = A.RS, TE = A.TE, T = A.T;
    const int tpr = A.tiles_per_row, ntiles = A.ntiles, XR = A.XR;
    const bool first = flow == 0;

    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(ldsw) + lane;
    const float* ldsf = reinterpret_cast<const float*>(ldsw);
    const float* bo = ldsf + IAF_PH_FLOATS + q * 16;
    const float* wm = bo + 64;
    const float* wsc = wm + 64;
    const int RS16 = (int)RS * 16, TE16 = (int)TE * 16;
    const int lane_l = q * RS16 + (wave * 16 + n + IAF_LP) * 16;
    const int lane_e = q * TE16 + (wave * 16 + n + A.c0) * 16;

    stage_words<IAF_HEAD_FLOATS>(wpack, ldsw);
    const float bmean = ldsf[IAF_PH_FLOATS + 192], bscale = ldsf[IAF_PH_FLOATS + 193];
    const float inv_m = ldsf[IAF_PH_FLOATS + 194];

    struct Src {
        rsrc_t rl, re, rx, rm, rs;
        int vl, ve, vx, vt;
    };
    auto tile_src = [&](int tile, int pred) -> Src {
        const int b = tile / tpr;
        const int tt = (tile - b * tpr) * 64;
        Src s;
        s.rl = mk_rsrc(lin + (size_t)b * IAF_W * RS, IAF_W * (int)RS * 4);
        s.re = mk_rsrc(enc + (size_t)b * IAF_CD * TE, IAF_CD * (int)TE * 4);
        s.rx = mk_rsrc(xin + (size_t)b * XR, XR * 4);
        s.rm = mk_rsrc(Min + (size_t)b * T, (int)T * 4);
        s.rs = mk_rsrc(Sin + (size_t)b * T, (int)T * 4);
        s.vl = (lane_l + tt * 16) | pred;
        s.ve = (lane_e + tt * 16) | pred;
        s.vx = ((IAF_XP + tt + wave * 16 + n) * 4) | pred;
        s.vt = ((tt + wave * 16 + n) * 4) | pred | (first ? OOB : 0);
        return s;
    };
    // K-steps 0-1: out1 over relu(l); 2-9: mel_cond_out1 over the 256 enc channels
    auto loadK = [&](const Src& s, int ks) -> KOp<1> {
        KOp<1> o;
        if (ks < 2) {
            o.h[0] = buf_ld4<SC1>(s.rl, s.vl, (4 * ks) * RS16);
            o.l[0] = buf_ld4<SC1>(s.rl, s.vl, (8 + 4 * ks) * RS16);
        } else {
            o.h[0] = buf_ld4(s.re, s.ve, (4 * (ks - 2)) * TE16);
            o.l[0] = buf_ld4(s.re, s.ve, (32 + 4 * (ks - 2)) * TE16);
        }
        return o;
    };
    KOp<1> bc[10];
    float xc, mc, sc;                 // x, mean_tot, scale_tot of this lane's column (flow input side)
    auto load_xms = [&](const Src& s, float& x_, float& m_, float& s_) {
        x_ = ldf_sc1(s.rx, s.vx);
        m_ = ldf_sc1(s.rm, s.vt);
        s_ = ldf_sc1(s.rs, s.vt);
    };
    auto load_all = [&](int tile) {
        const Src s0 = tile_src(tile, 0);
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) bc[ks] = loadK(s0, ks);
        load_xms(s0, xc, mc, sc);
    };

    int tile = cx.j;
    if (!dep_wait(cx.dep, tile, lane, 0x300 + A.s0)) return;
    load_all(tile);
    unsigned c_async = dep_poll(cx.dep, lane);
    unsigned done = 0;
    for (;;) {
        const int b = tile / tpr;
        const int t0 = (tile - b * tpr) * 64 + wave * 16;
        const int next = tile + cx.nwg;
        const bool has_next = next < ntiles;
        const bool rdy = has_next && dep_ready(c_async, cx.dep, next);
        const Src sn = tile_src(has_next ? next : tile, rdy ? 0 : OOB);
        c_async = dep_poll(cx.dep, lane);
        float xn, mn, sn_;
        load_xms(sn, xn, mn, sn_);

        f4 acc[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = (f4){0.f, 0.f, 0.f, 0.f};
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
            wn_u4 bh = bc[ks].h[0], bl = bc[ks].l[0];
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
                acc[mb] = mfma3(a[ks & 1][mb][0], a[ks & 1][mb][1], bh, bl, acc[mb]);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            bc[ks] = loadK(sn, ks);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == KPUB_HEAD) {
                WN_WAIT_VM(NPUB_HEAD);
                publish(done, cx.rown, cx.own_off);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            float pm = 0.f, ps = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = fmaxf(fmaf(acc[mb][r], inv_m, bo[mb * 4 + r]), 0.f);
                    pm = fmaf(wm[mb * 4 + r], o, pm);
                    ps = fmaf(wsc[mb * 4 + r], o, ps);
                }
            pm += __shfl_xor(pm, 16);
            ps += __shfl_xor(ps, 16);
            pm += __shfl_xor(pm, 32);
            ps += __shfl_xor(ps, 32);
            const float mean = pm + bmean;
            const float s = fminf(fmaxf(softplus_tf(ps + bscale), EXP_M9), EXP_7);   // :105-114
            const float xo = xc * s + mean;                                          // :277
            const float mo = first ? mean : mean + mc * s;                           // :322-323
            const float so = first ? s : sc * s;
            // only the q == 0 lanes store (the others go out of range): no divergent branch around VMEM
            const int kill = q == 0 ? 0 : OOB;
            const rsrc_t rxo = mk_rsrc(xout + (size_t)b * XR, XR * 4);
            const rsrc_t rmo = mk_rsrc(Mout + (size_t)b * T, (int)T * 4);
            const rsrc_t rso = mk_rsrc(Sout + (size_t)b * T, (int)T * 4);
            stf_sc1(xo, rxo, ((IAF_XP + t0 + n) * 4) | kill);
            stf_sc1(mo, rmo, ((t0 + n) * 4) | kill);
            stf_sc1(so, rso, ((t0 + n) * 4) | kill);
        }
        ++done;
        xc = xn; mc = mn; sc = sn_;
        if (!rdy) {
            WN_WAIT_VM(0);
            publish(done, cx.rown, cx.own_off);
            if (!has_next) return;
            if (!dep_wait(cx.dep, next, lane, 0x400 + A.s0)) return;
            load_all(next);
            c_async = dep_poll(cx.dep, lane);
        }
        tile = next;
    }
}

__global__ __launch_bounds__(256, 1) void iaf_pipe_kernel(const PipeArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    // logical workgroup index: consecutive logical indices share an XCD (block b runs on XCD b % 8;
    // observed placement, used for speed only)
    const int G = gridDim.x;
    int L = blockIdx.x;
    if ((G & 7) == 0) L = (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3);
    // even split of the workgroups over the stages of this launch; the first `rem` stages get one more
    const int nst = A.nst, base = G / nst, rem = G - base * nst;
    int i, j;
    if (L < rem * (base + 1)) {
        i = L / (base + 1);
        j = L - i * (base + 1);
    } else {
        const int r = L - rem * (base + 1);
        i = rem + r / base;
        j = r - (i - rem) * base;
    }
    auto wgs = [&](int st) { return min(PIPE_MAX_WG, st < rem ? base + 1 : base); };
    PipeCtx cx;
    cx.nwg = wgs(i);
    cx.j = j;
    if (j >= cx.nwg || j >= A.ntiles) return;            // uniform per workgroup
    cx.lane = threadIdx.x & 63;
    cx.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sg = A.s0 + i;                              // global stage index
    const rsrc_t rcnt = mk_rsrc(A.cnt, A.cnt_words * 4);
    cx.rown = rcnt;
    cx.own_off = cx.lane == 0 ? (sg * 64 + j * 4 + cx.wave) * 4 : OOB;
    cx.dep.rerr = mk_rsrc(A.cnt + (A.cnt_words - 1), 4);
    // stage i consumes from stage i - 1 of the same launch; the first stage of a launch has no
    // in-launch producer (earlier launches have completed)
    cx.dep.nwg = i > 0 ? wgs(i - 1) : 1;
    cx.dep.rc = mk_rsrc(A.cnt + (size_t)(sg > 0 ? sg - 1 : 0) * 64, i > 0 ? 256 : 0);
    cx.dep.jw = cx.lane >> 2;
    cx.dep.valid = i > 0 && cx.lane < 4 * cx.dep.nwg;

    const int* stg = A.stages + (size_t)sg * 8;
    const int type = stg[0];
    if (type == 2) pipe_head(A, stg, ldsw, cx);
    else if (type == 1) pipe_layer<true>(A, stg, ldsw, cx);
    else pipe_layer<false>(A, stg, ldsw, cx);
}

__global__ void pipe_zero_pads_kernel(unsigned* __restrict__ lbuf, int64_t row_words, int pad_words, int rows) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= pad_words) return;
    for (int row = blockIdx.y; row < rows; row += gridDim.y) lbuf[(size_t)row * row_words + c] = 0u;
}

// NaN-poison the output when the pipeline raised its error word (a wait timed out)
__global__ void pipe_poison_kernel(const unsigned* __restrict__ err, float* __restrict__ wav, int64_t n) {
    if (*err == 0) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wav[i] = __builtin_nanf("");
}

}  // namespace

// ---------------------------------------------------------------------------
// stage table: appended to the weight blob by wn_pack_iaf_h
void wn_iaf_p_stage_table(const wn_handle* h, std::vector<int>& tab) {
    const wn_config& c = h->cfg;
    int g = 0;                                   // global layer index = activation buffer index
    for (int k = 0; k < c.n_flows; ++k) {
        const IafFlowPack& fp = h->flows[k];
        for (size_t i = 0; i < fp.layers.size(); ++i, ++g) {
            const int row[8] = {i == 0 ? 1 : 0, fp.layers[i].dilation, (int)fp.layers[i].off_h, (int)fp.start_off,
                                i == 0 ? 0 : g - 1, g, k, fp.deconv_stack};
            tab.insert(tab.end(), row, row + 8);
        }
        const int row[8] = {2, 0, (int)fp.head_off_h, 0, g - 1, 0, k, fp.deconv_stack};
        tab.insert(tab.end(), row, row + 8);
    }
}

bool wn_iaf_p_supported(const wn_handle* h) {
    const wn_config& c = h->cfg;
    if (c.precision != WN_PREC_F16X3) return false;
    for (int k = 0; k < c.n_flows; ++k)
        if (h->flows[k].layers.empty() || h->flows[k].layers[0].dilation != 1) return false;
    return true;
}

int wn_iaf_p_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_pipe_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (IAF_LAYER_H_WORDS + IAF_START_LDS_WORDS) * 4));
    return WN_OK;
}

// utterances per pipeline launch: the write-once activation buffers cost n_layers x 256 B per sample
int wn_iaf_p_chunk(const wn_handle* h, int B, int64_t T) {
    int nl = 0;
    for (const IafFlowPack& fp : h->flows) nl += (int)fp.layers.size();
    const double per_utt = (double)nl * IAF_W * (double)(IAF_LP + T) * 4.0;
    double budget = 24e9;
    if (const char* e = getenv("WN_PIPE_GB")) budget = atof(e) * 1e9;
    int bc = (int)(budget / per_utt);
    // 32-bit tile indices and byte offsets inside the kernel
    bc = std::max(1, std::min(bc, B));
    return bc;
}

int wn_iaf_p_run(wn_handle* h, const WnPipeBufs& P, int B0, int Bc, hipStream_t st) {
    const int nstages = h->pipe_stages;
    PipeArgs A;
    A.blob = h->d_blob;
    A.stages = reinterpret_cast<const int*>(h->d_blob + h->pipe_tab_off);
    A.lbuf = P.lbuf;
    A.lbuf_words = (long long)Bc * IAF_W * P.RS;
    A.enc = P.enc + (size_t)B0 * IAF_CD * P.TE;
    A.enc_words = P.enc_words;
    A.X = P.X + (size_t)B0 * P.XR;
    A.x_floats = P.x_floats;
    A.M = P.M + (size_t)B0 * P.T;
    A.S = P.S + (size_t)B0 * P.T;
    A.ms_floats = P.ms_floats;
    A.cnt = P.cnt;
    A.cnt_words = nstages * 64 + 1;
    A.RS = P.RS;
    A.TE = P.TE;
    A.T = P.T;
    A.c0 = P.c0;
    A.XR = P.XR;
    A.tiles_per_row = (int)(P.T / 64);
    A.ntiles = Bc * A.tiles_per_row;
    WN_HIP(h, hipMemsetAsync(P.cnt, 0, (size_t)A.cnt_words * 4, st));
    int grid = h->num_cu;
    if (const char* e = getenv("WN_PIPE_GRID")) grid = std::max(1, atoi(e));
    for (int s0 = 0; s0 < nstages; s0 += grid) {
        A.s0 = s0;
        A.nst = std::min(grid, nstages - s0);
        hipLaunchKernelGGL(iaf_pipe_kernel, dim3(grid), dim3(256), (IAF_LAYER_H_WORDS + IAF_START_LDS_WORDS) * 4, st, A);
    }
    return WN_OK;
}

void wn_iaf_p_zero_pads(unsigned* lbuf, int64_t RS, int rows, hipStream_t st) {
    const int pad = 4 * IAF_LP;
    dim3 g((pad + 255) / 256, std::min(rows, 32768));
    hipLaunchKernelGGL(pipe_zero_pads_kernel, g, dim3(256), 0, st, lbuf, 4 * RS, pad, rows);
}

void wn_iaf_p_poison(const unsigned* err, float* wav, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(pipe_poison_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, err, wav, n);
}
