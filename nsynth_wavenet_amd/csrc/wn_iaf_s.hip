// "Segment-resident" form of the IAF residual stack (EXPERIMENTAL, precision 'f16x3-resident'): ONE launch runs
// every layer of a flow over a pass of time, with the activations AND the upsampled mel of a time segment living on
// the CU that owns it.
//
//   wavenet/parallel_wavenet.py:227-254 (residual layers of one flow), masked.py:160-232 (causal dilated conv).
//
// Why: with one launch per layer (wn_iaf_c.hip / wn_iaf_h.hip) a 4.8 s utterance moves, per layer and sample,
// 256 B of `l` in, 256 B out and either 1 KB of `enc` (fused) or 2 x 256 B of the hoisted projection through
// the fabric -- ~5 GB per utterance at the ~6 TB/s an XCD fabric port delivers -- and pays a launch floor and a
// tile round-off 60 times.  Here nothing of that crosses the fabric twice:
//
//   * workgroup c (one per CU, 4 waves) owns a contiguous segment of the pass (NBW blocks of 16 columns per wave,
//     template parameter 2 or 3); it runs ALL layers of the flow on them, layer after layer;
//   * the upsampled mel of the segment (`enc`, 1 KB per column) is loaded ONCE per launch into registers -- it is
//     the MFMA B operand of the conditioning 1x1 of every layer (64 registers per block and lane; the compiler
//     keeps them in AGPRs);
//   * the residual stream `l` of the segment lives in LDS (two buffers, read / write);
//   * the dilated-conv fragments (48 KB) and the residual fragments (8 KB) of a layer arrive in LDS by LDS-DMA
//     (global_load_lds_dwordx4, no registers) under the previous phase; the conditioning weights (64 KB per
//     layer) stream from L2 as MFMA A operands;
//   * a layer's output is ALSO written to a global buffer of its own (write-once, write-through `sc1` stores):
//     the causal taps t-d, t-2d that fall LEFT of the segment are read from there -- the left neighbours'
//     columns -- after those neighbours published "layer j done" in a per-wave progress word (same R1 hand-off
//     as wn_iaf_p.hip; every spin is bounded).  A wave only ever waits for LOWER-numbered workgroups.
//   * software pipeline over layers: the conditioning K-steps of layer j+1 (MFMA, depend on nothing layer j
//     produces) are issued in the same region as the epilogue of layer j (VALU).
//
// Measured on MI355X (config 2, one utterance): correct (golden vectors), but NOT faster than the default form --
// 1.61 ms per call in its first, un-pipelined version (three blocks per wave), 2.2 ms in this one.  Where the
// cycles go (WN_SRF_DEBUG=<workgroup> prints s_memtime stamps per layer): the conditioning A fragments are read by
// all four waves of a CU through its one 64 B/clk L1 path (256 KB per layer: ~4k cycles against ~3k cycles of
// MFMA work for two blocks per wave), the gate / split epilogue costs ~2.6k cycles per block and does not overlap
// with MFMAs inside one 512-register wave (the compiler keeps the two streams apart; with three blocks per wave
// the fused region spills and collapses), and every layer pays two workgroup barriers plus a neighbour hand-off.
// Kept as a tested form and as the record of that measurement; see DESIGN.md section 3.7.
//
// A launch covers one flow, one utterance and one pass; longer utterances take several passes, left to right
// (a pass reads the previous passes' columns like any left neighbour).
#include <algorithm>
#include <cstdlib>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_mfma_h.h"

namespace {

constexpr int SC1 = 16;
constexpr int OOB = (int)0x80000000;
constexpr unsigned SPIN_LIMIT = 1u << 22;

constexpr int NBW_MAX = 3;                     // blocks of 16 columns per wave (kernel template parameter: 2 or 3)
constexpr int IMG_A_WORDS = 6 * 2048;          // dilated-conv fragments, K-steps 0-5
constexpr int IMG_TAIL_WORDS = IAF_PR_FLOATS + 128 + 4;   // residual 1x1 fragments | biases | 1/scales
constexpr int IMG_WORDS = IMG_A_WORDS + IMG_TAIL_WORDS;   // 14 468 words = 57 872 B
static_assert(IMG_WORDS % 4 == 0, "");
// LDS: two `l` buffers | dilated fragments | residual fragments | two bias tails
constexpr int srf_lds_bytes(int nbw) { return 2 * 16 * (16 * 4 * nbw) * 16 + IMG_A_WORDS * 4 + IAF_PR_FLOATS * 4 + 2 * 132 * 4; }
constexpr int SRF_LDS_BYTES = srf_lds_bytes(NBW_MAX);   // 154 480 + ... = 254 ... see static_assert
static_assert(SRF_LDS_BYTES <= 160 * 1024, "LDS budget");

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ inline rsrc_t mk_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ inline unsigned ld1_sc1(rsrc_t r, int voff) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, SC1);
}
__device__ inline void st1_sc1(unsigned v, rsrc_t r, int voff) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, 0, SC1);
}
__device__ inline void st4_sc1(wn_u4 v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, SC1);
}
#define WN_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

struct SrfLayer {
    unsigned w_off;       // float offset of the split-fp16 layer pack (IAF_LAYER_H_WORDS) in the blob
    int d;                // dilation
};
constexpr int SRF_MAX_LAYERS = 64;

struct SrfArgs {
    const float* blob;
    const unsigned* enc;          // this utterance's upsampled mel, G4 words [2][32][TE][4]
    unsigned* lbuf;               // activation buffers [nbuf][16 rows][RS][4] of this utterance; layer j reads
    long long lbuf_words;         //   buffer j and writes buffer j + 1
    unsigned* flags;              // [grid][4] progress words, then the error word
    int flag_words;
    unsigned epoch;               // progress base of this launch (flags are monotone over a generate call)
    long long RS, TE;
    int c0;                       // centre-crop offset of enc
    int col0;                     // first column of the pass (multiple of 16)
    int nblk;                     // 16-column blocks in the pass
    int nlayers;
    unsigned long long* dbg;      // dev aid (WN_SRF_DEBUG): per-layer s_memtime stamps of one wave, or null
    int dbg_c;
    SrfLayer layers[SRF_MAX_LAYERS];
};

// wait until the left neighbours this layer reads from have published `need`; false = gave up
__device__ inline bool wait_left(rsrc_t rf, rsrc_t rerr, int c, int nn, unsigned need, int lane) {
    for (int g0 = 0; g0 < nn; g0 += 16) {
        const int cn = c - 1 - g0 - (lane >> 2);
        const bool valid = (g0 + (lane >> 2)) < nn;
        const int off = valid ? (cn * 4 + (lane & 3)) * 4 : OOB;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = ld1_sc1(rf, off);
            // progress words only grow inside a generate call; compare as a signed distance
            const bool ok = !valid || (int)(v - need) >= 0;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 255u) == 255u) {
                if (__builtin_amdgcn_readfirstlane(ld1_sc1(rerr, 0)) != 0) return false;
                if (spins > SPIN_LIMIT) {
                    st1_sc1(0x500u + (unsigned)c, rerr, lane == 0 ? 0 : OOB);
                    return false;
                }
            }
        }
    }
    return true;
}

template <int NBW>
__global__ __launch_bounds__(256, 1) void iaf_srf_kernel(const SrfArgs A) {
    constexpr int SEG = 16 * 4 * NBW;              // columns per workgroup
    constexpr int LROW = SEG * 16;                 // bytes of one group row of an LDS `l` buffer
    constexpr int LBUF_BYTES = 16 * LROW;          // 2 planes x 8 groups
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int c = blockIdx.x, G = gridDim.x;

    // ---- my share of the pass: contiguous blocks per workgroup, contiguous blocks per wave ----
    const int base = A.nblk / G, rem = A.nblk - base * G;
    const int my_nblk = base + (c < rem ? 1 : 0);
    const int my_b0 = c * base + min(c, rem);
    const int wb = my_nblk >> 2, wr = my_nblk & 3;
    const int cnt = wb + (wave < wr ? 1 : 0);                   // blocks of this wave (<= NBW)
    const int blk0 = wave * wb + min(wave, wr);                 // first block of this wave inside the segment
    const int seg_col = A.col0 + 16 * my_b0;                    // first global column of the segment
    const rsrc_t rf = mk_rsrc(A.flags, A.flag_words * 4);
    const rsrc_t rerr = mk_rsrc(A.flags + (A.flag_words - 1), 4);
    const int own_flag = lane == 0 ? (c * 4 + wave) * 4 : OOB;
    const unsigned L = (unsigned)A.nlayers;
    if (my_nblk == 0 || cnt == 0) st1_sc1(A.epoch + L, rf, own_flag);   // nobody has to wait for an idle wave
    if (my_nblk == 0) return;

    unsigned char* lds8 = reinterpret_cast<unsigned char*>(lds);
    unsigned* frag = lds + 2 * LBUF_BYTES / 4;                                     // dilated-conv fragments of the current layer
    unsigned* prw = frag + IMG_A_WORDS;                                            // residual 1x1 fragments of the current layer
    float* tails = reinterpret_cast<float*>(prw + IAF_PR_FLOATS);                  // [2][132]: biases | 1/scales, per layer parity
    const wn_u4* PRl = reinterpret_cast<const wn_u4*>(prw) + lane;                 // [(mb*2+plane)*64]
    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(frag) + lane;                 // [((ks*4+mb)*2+plane)*64]

    const int RS16 = (int)A.RS * 16, TE16 = (int)A.TE * 16;
    const rsrc_t rblob = mk_rsrc(A.blob, 0x7ffffff0);
    const rsrc_t renc = mk_rsrc(A.enc, IAF_CD * (int)A.TE * 4);

    // ---- enc of my columns: the B operands of the conditioning 1x1 of every layer, loaded once ----
    KOp<1> enc[NBW][8];
#pragma unroll
    for (int k = 0; k < NBW; ++k) {
        const int col = seg_col + 16 * (blk0 + k) + n + A.c0;
        const int vo = (q * TE16 + col * 16) | (k < cnt ? 0 : OOB);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            enc[k][ks].h[0] = buf_ld4(renc, vo, (4 * ks) * TE16);
            enc[k][ks].l[0] = buf_ld4(renc, vo, (32 + 4 * ks) * TE16);
        }
    }
    // ---- the launch's input (buffer 0) of my segment -> LDS buffer 0 ----
    {
        const rsrc_t rin0 = mk_rsrc(A.lbuf, IAF_W * (int)A.RS * 4);
#pragma unroll
        for (int k = 0; k < NBW; ++k) {
            if (k < cnt) {
                const int lc = 16 * (blk0 + k) + n;
                const int vo = q * RS16 + (IAF_LP + seg_col + lc) * 16;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {               // rows 4*r4 + q: planes hi (0-7) and lo (8-15)
                    const wn_u4 v = buf_ld4(rin0, vo, (4 * r4) * RS16);
                    *reinterpret_cast<wn_u4*>(lds8 + (4 * r4 + q) * LROW + lc * 16) = v;
                }
            }
        }
    }
    // dilated-conv fragments (48 KB) of layer jj: global -> LDS by LDS-DMA (global_load_lds_dwordx4: each lane's 16
    // bytes land at M0 + 16 * lane, no registers), 12 instructions per wave; the 132-word bias tail goes through
    // one register of the first 33 threads
    constexpr int FR_PER_THREAD = IMG_A_WORDS / 4 / 256;        // 12
    auto frag_dma = [&](int jj) {
        const unsigned* src = reinterpret_cast<const unsigned*>(A.blob) + A.layers[jj].w_off;
#pragma unroll
        for (int i = 0; i < FR_PER_THREAD; ++i) {
            const unsigned* g = src + (size_t)(i * 256 + wave * 64 + lane) * 4;
            unsigned* l = frag + (i * 256 + wave * 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    // residual 1x1 fragments (8 KB) of layer jj -> LDS, 2 instructions per wave; issued at the top of layer jj (the
    // epilogue of layer jj-1 has been left by every wave), needed by the epilogue of layer jj
    auto pr_dma = [&](int jj) {
        const unsigned* src = reinterpret_cast<const unsigned*>(A.blob) + A.layers[jj].w_off + IAF_P_FLOATS;
#pragma unroll
        for (int i = 0; i < IAF_PR_FLOATS / 4 / 256; ++i) {
            const unsigned* g = src + (size_t)(i * 256 + wave * 64 + lane) * 4;
            unsigned* l = prw + (i * 256 + wave * 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    auto tail_load = [&](int jj) -> wn_u4 {
        return buf_ld4(rblob, threadIdx.x < 33 ? (IAF_P_FLOATS + IAF_PR_FLOATS) * 4 + (int)threadIdx.x * 16 : OOB,
                       (int)A.layers[jj].w_off * 4);
    };
    auto tail_store = [&](int jj, const wn_u4& tl) {
        if (threadIdx.x < 33) reinterpret_cast<wn_u4*>(tails + (jj & 1) * 132)[threadIdx.x] = tl;
    };
    // A fragments of the conditioning weights of layer jj, K-step ks: straight from L2 into registers
    auto load_ac = [&](int jj, int ks, wn_u4 (&dst)[4][2]) {
        const int wo = (int)A.layers[jj].w_off * 4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            dst[mb][0] = buf_ld4(rblob, (((6 + ks) * 4 + mb) * 2 + 0) * 1024 + lane * 16, wo);
            dst[mb][1] = buf_ld4(rblob, (((6 + ks) * 4 + mb) * 2 + 1) * 1024 + lane * 16, wo);
        }
    };

    // ---- prologue: weights of layer 0, conditioning 1x1 of layer 0 (parallel_wavenet.py:238-244) ----
    f4 accn[NBW][4];
    {
        frag_dma(0);
        pr_dma(0);
        const wn_u4 tl = tail_load(0);
#pragma unroll
        for (int k = 0; k < NBW; ++k)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) accn[k][mb] = (f4){0.f, 0.f, 0.f, 0.f};
        wn_u4 a[3][4][2];
        load_ac(0, 0, a[0]);
        load_ac(0, 1, a[1]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 2 < 8) load_ac(0, ks + 2, a[(ks + 2) % 3]);
#pragma unroll
            for (int k = 0; k < NBW; ++k)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    accn[k][mb] = mfma3(a[ks % 3][mb][0], a[ks % 3][mb][1], enc[k][ks].h[0], enc[k][ks].l[0], accn[k][mb]);
            __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12 * NBW, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        tail_store(0, tl);
        WN_WAIT_VM0();
    }
    __syncthreads();

    const bool dbg_on = A.dbg && c == A.dbg_c && wave == 0 && lane == 0;
#define STAMP(i) do { if (dbg_on) A.dbg[j * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    for (int j = 0; j < A.nlayers; ++j) {
        STAMP(0);
        const int d = A.layers[j].d;
        const unsigned* lin_g = A.lbuf + (size_t)j * A.lbuf_words;
        unsigned* lout_g = A.lbuf + (size_t)(j + 1) * A.lbuf_words;
        const rsrc_t rin = mk_rsrc(lin_g, IAF_W * (int)A.RS * 4);
        const rsrc_t rout = mk_rsrc(lout_g, IAF_W * (int)A.RS * 4);
        const unsigned char* lrd = lds8 + (j & 1) * LBUF_BYTES;
        unsigned char* lwr = lds8 + ((j + 1) & 1) * LBUF_BYTES;
        const float* tail = tails + (j & 1) * 132;
        const float* bg = tail + q * 16;
        const float* br = bg + 64;
        const bool has_next = j + 1 < A.nlayers;

        f4 acc[NBW][4];
#pragma unroll
        for (int k = 0; k < NBW; ++k)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[k][mb] = accn[k][mb];

        if (j > 0) pr_dma(j);       // lands during the dilated K-steps; waited for (vmcnt) before the barrier below
        if (cnt > 0) {
            // left neighbours whose layer j-1 output this wave reads (taps t-d, t-2d left of the segment)
            int nn = 0;
            if (j > 0 && c > 0) {
                const int reach = 2 * d - 16 * blk0;                // columns left of the SEGMENT this wave reads
                const int per = 16 * max(base, 1);                  // every left neighbour of the pass has >= base blocks
                if (reach > 0) nn = min(c, (reach + per - 1) / per);
            }
            if (nn > 0 && !wait_left(rf, rerr, c, nn, A.epoch + (unsigned)j, lane)) return;
            STAMP(1);
            // dilated conv, K-step-outer over my blocks.  Tap of column lc = 16 (blk0 + k) + n with shift sh:
            // source column lc - sh; inside the segment -> LDS, left of it -> the global buffer of layer j-1
            // (which also holds the zeros left of the utterance).  Order of the K-steps: tap t first (always
            // local: covers the latency of the remote loads), then t-d, then t-2d; the remote operands go through
            // two register slots per block that are refilled as soon as they are consumed.
            struct RemS { wn_u4 h[NBW], l[NBW]; };
            auto load_rem = [&](int tp, int sx) -> RemS {
                RemS r;
                const int sh = (2 - tp) * d;
#pragma unroll
                for (int k = 0; k < NBW; ++k) {
                    const int lc = 16 * (blk0 + k) + n;
                    const bool remote = lc - sh < 0 && k < cnt;
                    const int vo = (q * RS16 + (IAF_LP + seg_col + lc - sh) * 16) | (remote ? 0 : OOB);
                    r.h[k] = buf_ld4<SC1>(rin, vo, (4 * sx) * RS16);
                    r.l[k] = buf_ld4<SC1>(rin, vo, (8 + 4 * sx) * RS16);
                }
                return r;
            };
            RemS rs0 = load_rem(1, 0), rs1 = load_rem(1, 1);          // tap t-d, both channel halves
            auto tap_ops = [&](int ks, const RemS& rs, KOp<1> (&dst)[NBW]) {
                const int tp = ks >> 1, sx = ks & 1;
                const int sh = (2 - tp) * d;
#pragma unroll
                for (int k = 0; k < NBW; ++k) {
                    const int src = 16 * (blk0 + k) + n - sh;
                    const int lsrc = src < 0 ? 0 : (src > SEG - 1 ? SEG - 1 : src);   // (blocks beyond `cnt` may point past the segment)
                    wn_u4 bh = *reinterpret_cast<const wn_u4*>(lrd + (4 * sx + q) * LROW + lsrc * 16);
                    wn_u4 bl = *reinterpret_cast<const wn_u4*>(lrd + (8 + 4 * sx + q) * LROW + lsrc * 16);
                    if (tp < 2) {
                        const bool remote = src < 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bh[e] = remote ? rs.h[k][e] : bh[e];
                            bl[e] = remote ? rs.l[k][e] : bl[e];
                        }
                    }
                    dst[k].h[0] = bh;
                    dst[k].l[0] = bl;
                }
            };
            constexpr int ORD[6] = {4, 5, 2, 3, 0, 1};
            // A fragments: ONE register set, refilled row block by row block right after the row block's last
            // MFMA of the K-step was issued (the other three row blocks' MFMAs cover the LDS latency)
            wn_u4 a[4][2];
            KOp<1> bq[2][NBW];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                a[mb][0] = Pl[((ORD[0] * 4 + mb) * 2 + 0) * 64];
                a[mb][1] = Pl[((ORD[0] * 4 + mb) * 2 + 1) * 64];
            }
            tap_ops(ORD[0], rs0, bq[0]);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (i + 1 < 6) {
                    const int kn = ORD[i + 1];
                    tap_ops(kn, (kn & 1) ? rs1 : rs0, bq[(i + 1) & 1]);
                    // the slot just consumed held tap t-d; refill it with tap t-2d of the same channel half
                    if (kn == 2) rs0 = load_rem(0, 0);
                    if (kn == 3) rs1 = load_rem(0, 1);
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
                    for (int k = 0; k < NBW; ++k)
                        acc[k][mb] = mfma3(a[mb][0], a[mb][1], bq[i & 1][k].h[0], bq[i & 1][k].l[0], acc[k][mb]);
                    if (i + 1 < 6) {
                        a[mb][0] = Pl[((ORD[i + 1] * 4 + mb) * 2 + 0) * 64];
                        a[mb][1] = Pl[((ORD[i + 1] * 4 + mb) * 2 + 1) * 64];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP(2);
        // everybody is done with the fragments of layer j (they are replaced under the epilogue) and the residual
        // fragments of layer j have landed
        WN_WAIT_VM0();
        __syncthreads();
        STAMP(3);
        wn_u4 ftl = (wn_u4){0u, 0u, 0u, 0u};
        if (has_next) {
            frag_dma(j + 1);
            ftl = tail_load(j + 1);
        }
        if (cnt > 0) {
            const float inv_m = tail[128], inv_r = tail[129];
            // ---- fused region: epilogue of layer j (VALU) interleaved with the conditioning 1x1 of layer j+1
            //      (MFMA), which depends on nothing this layer produces ----
#pragma unroll
            for (int k = 0; k < NBW; ++k)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) accn[k][mb] = (f4){0.f, 0.f, 0.f, 0.f};
            if (has_next) {
                wn_u4 a[2][4][2];
                load_ac(j + 1, 0, a[0]);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 1 < 8) load_ac(j + 1, ks + 1, a[(ks + 1) & 1]);
#pragma unroll
                    for (int k = 0; k < NBW; ++k)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb)
                            accn[k][mb] = mfma3(a[ks & 1][mb][0], a[ks & 1][mb][1], enc[k][ks].h[0], enc[k][ks].l[0], accn[k][mb]);
                }
            }
            // epilogue per block (parallel_wavenet.py:246-254): gate, residual 1x1, add, split, store
#pragma unroll
            for (int k = 0; k < NBW; ++k) {
                const int lc = 16 * (blk0 + k) + n;
                float g[2][4];
#pragma unroll
                for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // sigmoid(u) * tanh(v) = (e^{2v} - 1) / ((1 + e^{-u}) (e^{2v} + 1)): two exp, ONE rcp
                        const float u = fmaf(acc[k][mg][r], inv_m, bg[mg * 4 + r]);
                        const float v = fminf(fmaxf(fmaf(acc[k][mg + 2][r], inv_m, bg[(mg + 2) * 4 + r]), -15.f), 15.f);
                        const float eu = __expf(-u), ev = __expf(2.f * v);
                        g[mg][r] = (ev - 1.f) * __builtin_amdgcn_rcpf((1.f + eu) * (ev + 1.f));
                    }
                wn_u4 gh, gl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned hw, lw;
                    wn_split_pair(g[i >> 1][(i & 1) * 2], g[i >> 1][(i & 1) * 2 + 1], hw, lw);
                    gh[i] = hw;
                    gl[i] = lw;
                }
                // the residual's skip input: tap t of this column, still in the read buffer
                wn_u4 ch[2], cl[2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int lcc = lc > SEG - 1 ? SEG - 1 : lc;
                    ch[s2] = *reinterpret_cast<const wn_u4*>(lrd + (4 * s2 + q) * LROW + lcc * 16);
                    cl[s2] = *reinterpret_cast<const wn_u4*>(lrd + (8 + 4 * s2 + q) * LROW + lcc * 16);
                }
                wn_u4 oh[2], ol[2];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const f4 rc = mfma3(PRl[(mb * 2 + 0) * 64], PRl[(mb * 2 + 1) * 64], gh, gl, (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        float l0, l1;
                        wn_join_pair(ch[mb >> 1][(mb & 1) * 2 + rp], cl[mb >> 1][(mb & 1) * 2 + rp], l0, l1);
                        const float v0 = l0 + fmaf(rc[2 * rp], inv_r, br[mb * 4 + 2 * rp]);
                        const float v1 = l1 + fmaf(rc[2 * rp + 1], inv_r, br[mb * 4 + 2 * rp + 1]);
                        unsigned hw, lw;
                        wn_split_pair(v0, v1, hw, lw);
                        oh[mb >> 1][(mb & 1) * 2 + rp] = hw;
                        ol[mb >> 1][(mb & 1) * 2 + rp] = lw;
                    }
                }
                // a block beyond `cnt` is not mine: its global stores go out of range, its LDS stores are skipped
                const int vo_out = (q * RS16 + (IAF_LP + seg_col + lc) * 16) | (k < cnt ? 0 : OOB);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    st4_sc1(oh[s2], rout, vo_out, (4 * s2) * RS16);
                    st4_sc1(ol[s2], rout, vo_out, (8 + 4 * s2) * RS16);
                }
                if (k < cnt) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        *reinterpret_cast<wn_u4*>(lwr + (4 * s2 + q) * LROW + lc * 16) = oh[s2];
                        *reinterpret_cast<wn_u4*>(lwr + (8 + 4 * s2 + q) * LROW + lc * 16) = ol[s2];
                    }
                }
            }
        }
        STAMP(4);
        if (has_next) tail_store(j + 1, ftl);
        // every global access of this layer is older than the fragment loads that just came back: the stores
        // of layer j have been acknowledged (vmcnt retires in issue order) -> publish "j + 1 layers done"
        WN_WAIT_VM0();
        if (cnt > 0 && has_next) st1_sc1(A.epoch + (unsigned)j + 1u, rf, own_flag);
        STAMP(5);
        // layer j's output in LDS and the fragments / tail of layer j + 1 are visible
        __syncthreads();
        STAMP(6);
        STAMP(7);
    }
    // the launch ends here: the kernel boundary publishes the last layer
}

}  // namespace

// ---------------------------------------------------------------------------
int wn_iaf_s_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_srf_kernel<3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, srf_lds_bytes(3)));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_srf_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, srf_lds_bytes(2)));
    return WN_OK;
}

int wn_iaf_s_max_cols(const wn_handle* h) {
    int nbw = 2;                                   // blocks per wave of a full pass
    if (const char* e = getenv("WN_SRF_NBW")) nbw = atoi(e) == 3 ? 3 : 2;
    return h->num_cu * 16 * 4 * nbw;
}
int wn_iaf_s_max_layers() { return SRF_MAX_LAYERS; }

// Runs layers [0, n) of flow `fp` of ONE utterance over the pass [col0, col0 + 16 nblk).  lbuf holds n + 1
// activation buffers of RS x 64 words each (buffer 0 = the flow's start-conv output, complete).
int wn_iaf_s_flow(wn_handle* h, const IafFlowPack& fp, const unsigned* enc_utt, unsigned* lbuf, int64_t RS, int64_t TE,
                  int c0, int col0, int nblk, unsigned* flags, unsigned epoch, hipStream_t st) {
    if ((int)fp.layers.size() > SRF_MAX_LAYERS) return wn_fail(h, WN_EINVAL, "flow has more than %d layers", SRF_MAX_LAYERS);
    if (nblk > h->num_cu * 4 * NBW_MAX) return wn_fail(h, WN_EINVAL, "pass of %d blocks exceeds the resident capacity", nblk);
    SrfArgs A;
    A.blob = h->d_blob;
    A.enc = enc_utt;
    A.lbuf = lbuf;
    A.lbuf_words = (long long)IAF_W * RS;
    A.flags = flags;
    A.flag_words = h->num_cu * 4 + 1;
    A.epoch = epoch;
    A.RS = RS;
    A.TE = TE;
    A.c0 = c0;
    A.col0 = col0;
    A.nblk = nblk;
    A.nlayers = (int)fp.layers.size();
    A.dbg = nullptr;
    A.dbg_c = 0;
    static unsigned long long* dbg_dev = nullptr;
    const char* de = getenv("WN_SRF_DEBUG");
    if (de) {
        if (!dbg_dev) (void)hipMalloc(&dbg_dev, SRF_MAX_LAYERS * 8 * sizeof(unsigned long long));
        A.dbg = dbg_dev;
        A.dbg_c = atoi(de);
    }
    for (size_t i = 0; i < fp.layers.size(); ++i) {
        A.layers[i].w_off = (unsigned)fp.layers[i].off_h;
        A.layers[i].d = fp.layers[i].dilation;
    }
    // a pass that fits two blocks per wave runs the two-block kernel (every wave computes all its block slots)
    if (nblk <= h->num_cu * 4 * 2)
        hipLaunchKernelGGL(iaf_srf_kernel<2>, dim3(h->num_cu), dim3(256), srf_lds_bytes(2), st, A);
    else
        hipLaunchKernelGGL(iaf_srf_kernel<3>, dim3(h->num_cu), dim3(256), srf_lds_bytes(3), st, A);
    if (de) {      // dev aid: print the stamps of workgroup WN_SRF_DEBUG (100 MHz ticks) for this launch
        unsigned long long hb[SRF_MAX_LAYERS * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb, dbg_dev, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "srf col0 %d nblk %d layers %d:", col0, nblk, A.nlayers);
        for (int j = 0; j < A.nlayers; ++j) {
            fprintf(stderr, "\n  L%02d d%3d", j, A.layers[j].d);
            for (int i = 1; i < 8; ++i) fprintf(stderr, " %6.2f", (double)(hb[j * 8 + i] - hb[j * 8 + i - 1]) * 0.01);
            if (j + 1 < A.nlayers) fprintf(stderr, " | next %6.2f", (double)(hb[(j + 1) * 8] - hb[j * 8 + 7]) * 0.01);
        }
        fprintf(stderr, "\n");
    }
    return WN_OK;
}
