// Residual layers of a flow on HOISTED conditioning with the activations resident per CU (EXPERIMENTAL, WITHHELD;
// precision 'f16x3-hoisted-resident'): one launch runs every layer of the flow over a pass of <= 256 x 192 columns.
//
//   wavenet/parallel_wavenet.py:227-254 (residual layers), masked.py:160-232 (causal dilated conv); the
//   conditioning 1x1 of every layer comes from the GEMM of wn_iaf_c.hip (`C`, Fastgen.cond_vars' precedent).
//
// Same arithmetic as iaf_layer_c_kernel (accumulators start from the C tile, six K-steps of taps, gate, residual
// 1x1), different data movement.  With one launch per layer a 4.8 s utterance gives every wave 2-3 blocks of 16
// columns per launch: the launch is a latency chain (load taps from the fabric, compute, store, launch boundary)
// and reaches 0.40 of the HBM roofline.  Here
//   * workgroup c (one per CU, TWELVE waves = three per SIMD, one block of 16 columns each) owns 192 columns of the
//     pass and keeps their residual stream `l` in LDS (two buffers, read / write) through all layers;
//   * the taps t-d, t-2d that fall inside the segment are LDS reads; the ones left of it are read from the global
//     buffer of the previous layer, which every workgroup also writes (write-once, write-through `sc1` stores)
//     and publishes through a per-wave progress word once its stores are acknowledged -- the R1 hand-off of the
//     CDNA4 guide; every spin is bounded and a wave only ever waits for LOWER-numbered workgroups;
//   * the dilated-conv fragments (48 KB) and the residual fragments (8 KB) of the next layer arrive in LDS by
//     LDS-DMA while the current layer computes;
//   * with three waves per SIMD the hardware overlaps one wave's gate / split arithmetic (VALU) with the other
//     wave's MFMAs -- the overlap a single 512-register wave per SIMD cannot get from the compiler.
// Fabric traffic per sample and layer: the C tile (256 B, read once, non-temporal), `l` written once (256 B) and
// the halo columns read back; no tap re-reads, no launch floor per layer.
//
// STATUS: WITHHELD -- not parity-clean.  wn_create refuses cond_mode 5 unless WN_UNVERIFIED_FORMS=1 is set, and no test
// or bench line uses it.  The golden vectors pass, but the cross-form fuzz (tests/tools/fuzz_gpu.py) finds about one
// wrong 16-column block per few thousand layer-blocks, run-to-run different.  What two days of bisection established
// (scripts/dev_res_race.py compares the per-layer buffers of repeated identical calls):
//   * signature: ONE residual-output channel of ONE block, all 16 columns, = skip input + bias: the residual-1x1
//     accumulator of that channel read as ~0.  It is always lanes 48-63 of the FIRST accumulator register the epilogue
//     reads after the residual MFMAs, whichever chain / register the allocator put there;
//   * the victim is always the lowest-numbered ACTIVE wave of a SIMD that holds exactly two active waves and one
//     idle one (workgroups with 5 or 10 of 12 blocks); fully occupied workgroups were clean in 48 runs x 30 layers;
//   * ruled out by instrumented builds: stale LDS-DMA fragments (the A operands as used compare equal to the blob),
//     loads in flight during the MFMAs (vmcnt(0) before them: still fails), the LDS bias reads between MFMA and
//     VALU (biases preloaded: still fails), accumulating out of place (in-place chains: still fails);
//   * without the 64 wait states below HALF of all blocks are wrong in some register allocations and none in
//     others; with them the rate drops to the rare event above.  A micro-benchmark of the bare pattern (twelve waves
//     leave a barrier, read fragments from LDS, run the four chains of three MFMAs from zero, read the results with
//     VALU compares; idle waves issuing LDS-DMA; scripts/ubench/mfma_hazard.hip) shows NO wrong value in 200 000
//     iterations at any occupancy with the compiler's own spacing -- so it is not simply "too few wait states after
//     v_mfma_f32_16x16x32_f16"; something else this kernel has (the inline-asm mixed-precision VALU ops around the
//     MFMAs, the sc1 stores and polls in flight, the LDS-DMA into the fragment buffer) is part of it.  Unexplained.
// The default form (one launch per layer, wn_iaf_c.hip) never shows this: its waves are not barrier-aligned into
// simultaneous MFMA bursts, and its determinism / cross-form checks have run clean over >10^4 fuzz cases.
//
// Measured on MI355X (config 2, one utterance): the eight launches of a call take 954 us
// against 1 014 us for the 52 layer launches of the default form, the whole call 1.61 ms against 1.57 ms (the
// separate start-conv and head launches eat the difference).  A layer-pass costs ~15k cycles (WN_RES_DEBUG prints
// the stamps): ~4.6k in the K loop (three waves per SIMD share the matrix pipe: 3.5k of MFMA issue), ~3.6k in the
// epilogue (all waves in their VALU phase at once: the barrier between K loop and epilogue -- needed because the
// fragment buffer is re-filled under the epilogue -- puts the waves in lockstep, so VALU and MFMA phases do not
// overlap across waves either), 2-4k waiting for the left neighbour's progress word (a flag hop is ~1.5-2 us
// under load) and ~2.5k in acknowledgement waits and barriers.  See DESIGN.md section 3.7.
#include <algorithm>
#include <cstdlib>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_mfma_h.h"

namespace {

constexpr int SC1 = 16;
constexpr int NT = 2;
constexpr int OOB = (int)0x80000000;
constexpr unsigned SPIN_LIMIT = 1u << 22;

#ifndef WN_RES_WAVES
#define WN_RES_WAVES 12
#endif
constexpr int RW = WN_RES_WAVES;               // waves per workgroup = blocks of 16 columns per workgroup (three per SIMD)
constexpr int SEG = 16 * RW;                   // 192 columns
constexpr int LROW = SEG * 16;                 // bytes of one group row of an LDS `l` buffer
constexpr int LBUF_BYTES = 16 * LROW;          // 2 planes x 8 groups = 32 KB
constexpr int FRAG_WORDS = 6 * 2048;           // dilated-conv fragments, K-steps 0-5 (48 KB)
constexpr int TAIL_WORDS = 132;                // biases (128) | 1/scale_main, 1/scale_res, pad
constexpr int RES_LDS_BYTES = 2 * LBUF_BYTES + FRAG_WORDS * 4 + IAF_PR_FLOATS * 4 + 2 * TAIL_WORDS * 4;
static_assert(RES_LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int RES_MAX_LAYERS = 64;

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
__device__ inline f4 ldf4_nt(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, NT));
}
#define WN_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

struct ResLayer {
    unsigned w_off;       // float offset of the split-fp16 layer pack in the blob
    int d;
};
struct ResArgs {
    const float* blob;
    const float* C;               // hoisted conditioning of this utterance, row block of the flow's first layer
    long long rb_floats;          //   floats per row block ((T / 16) * 1024)
    unsigned* lbuf;               // activation buffers [n + 1][16 rows][RS][4] of this utterance
    long long lbuf_words;
    unsigned* flags;              // [grid][RW] progress words, then the error word
    int flag_words;
    unsigned epoch;
    long long RS;
    int col0, nblk, nlayers;
    unsigned long long* dbg;      // dev aid (WN_RES_DEBUG): per-layer s_memtime stamps of one wave, or null
    int dbg_c;
    ResLayer layers[RES_MAX_LAYERS];
};

// wait until the `nn` left neighbours have published `need` (their RW progress words each); false = gave up
__device__ inline bool wait_left(rsrc_t rf, rsrc_t rerr, int c, int nn, unsigned need, int lane) {
    constexpr int NPER = 64 / RW;                  // neighbours polled by one load (RW progress words each)
    for (int g0 = 0; g0 < nn; g0 += NPER) {
        const int cn = c - 1 - g0 - lane / RW;
        const bool valid = lane < NPER * RW && (g0 + lane / RW) < nn;
        const int off = valid ? (cn * RW + (lane % RW)) * 4 : OOB;
        for (unsigned spins = 0;; ++spins) {
            const unsigned v = ld1_sc1(rf, off);
            const bool ok = !valid || (int)(v - need) >= 0;          // progress words only grow inside a call
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
            if (spins > 64u) __builtin_amdgcn_s_sleep(1);          // the first polls go back to back: a hop is ~1 us
            if ((spins & 255u) == 255u) {
                if (__builtin_amdgcn_readfirstlane(ld1_sc1(rerr, 0)) != 0) return false;
                if (spins > SPIN_LIMIT) {
                    st1_sc1(0x600u + (unsigned)c, rerr, lane == 0 ? 0 : OOB);
                    return false;
                }
            }
        }
    }
    return true;
}

__global__ __launch_bounds__(64 * RW, 1) void iaf_res_kernel(const ResArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int c = blockIdx.x, G = gridDim.x;

    // my share of the pass: contiguous blocks per workgroup, wave w takes block w of the segment
    const int base = A.nblk / G, rem = A.nblk - base * G;
    const int my_nblk = base + (c < rem ? 1 : 0);
    const int my_b0 = c * base + min(c, rem);
    const bool mine = wave < my_nblk;
    const int seg_col = A.col0 + 16 * my_b0;
    const int lc = 16 * wave + n;                                   // my column inside the segment
    const rsrc_t rf = mk_rsrc(A.flags, A.flag_words * 4);
    const rsrc_t rerr = mk_rsrc(A.flags + (A.flag_words - 1), 4);
    const int own_flag = lane == 0 ? (c * RW + wave) * 4 : OOB;      // (RW <= 16 words per workgroup are reserved)
    const unsigned L = (unsigned)A.nlayers;
    if (!mine) st1_sc1(A.epoch + L, rf, own_flag);                   // nobody has to wait for an idle wave
    if (my_nblk == 0) return;

    unsigned char* lds8 = reinterpret_cast<unsigned char*>(lds);
    unsigned* frag = lds + 2 * LBUF_BYTES / 4;
    unsigned* prw = frag + FRAG_WORDS;
    float* tails = reinterpret_cast<float*>(prw + IAF_PR_FLOATS);
    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(frag) + lane;       // [((ks*4+mb)*2+plane)*64]
    const wn_u4* PRl = reinterpret_cast<const wn_u4*>(prw) + lane;       // [(mb*2+plane)*64]

    const int RS16 = (int)A.RS * 16;
    const rsrc_t rblob = mk_rsrc(A.blob, 0x7ffffff0);
    const int kb = (seg_col >> 4) + wave;                               // my global 16-column block
    const int vc = mine ? kb * 4096 + lane * 16 : OOB;

    // LDS-DMA of `words` words of layer jj's pack (from word offset `src_off`) to `dst`: every lane's 16 bytes land
    // at M0 + 16 * lane; the eight waves take 1 KB slices in turn
    auto dma = [&](int jj, int src_off, unsigned* dst, int words) {
        const unsigned* src = reinterpret_cast<const unsigned*>(A.blob) + A.layers[jj].w_off + src_off;
        for (int i = wave; i < words / 256; i += RW) {
            const unsigned* g = src + (size_t)(i * 64 + lane) * 4;
            unsigned* l = dst + i * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    auto tail_load = [&](int jj) -> wn_u4 {
        const int ti = (int)threadIdx.x;
        return buf_ld4(rblob, ti < 33 ? (IAF_P_FLOATS + IAF_PR_FLOATS) * 4 + ti * 16 : OOB,
                       (int)A.layers[jj].w_off * 4);
    };
    auto c_tile = [&](int jj, f4 (&dst)[4]) {
        const rsrc_t rc = mk_rsrc(A.C + (size_t)jj * A.rb_floats, (int)(A.rb_floats * 4));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) dst[mb] = ldf4_nt(rc, vc, mb * 1024);
    };

    // ---- prologue: weights of layer 0, my columns of the launch's input (buffer 0) -> LDS buffer 0, C tile 0 ----
    dma(0, 0, frag, FRAG_WORDS);
    dma(0, IAF_P_FLOATS, prw, IAF_PR_FLOATS);
    {
        const wn_u4 tl = tail_load(0);
        if (mine) {
            const rsrc_t rin0 = mk_rsrc(A.lbuf, IAF_W * (int)A.RS * 4);
            const int vo = q * RS16 + (IAF_LP + seg_col + lc) * 16;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<wn_u4*>(lds8 + (4 * r4 + q) * LROW + lc * 16) = buf_ld4(rin0, vo, (4 * r4) * RS16);
        }
        if (threadIdx.x < 33) reinterpret_cast<wn_u4*>(tails)[threadIdx.x] = tl;
    }
    f4 cn[4];
    c_tile(0, cn);
    WN_WAIT_VM0();
    __syncthreads();

    const bool dbg_on = A.dbg && c == A.dbg_c && wave == 0 && lane == 0;
#define STAMP(i) do { if (dbg_on) A.dbg[j * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    for (int j = 0; j < A.nlayers; ++j) {
        STAMP(0);
        const int d = A.layers[j].d;
        const rsrc_t rin = mk_rsrc(A.lbuf + (size_t)j * A.lbuf_words, IAF_W * (int)A.RS * 4);
        const rsrc_t rout = mk_rsrc(A.lbuf + (size_t)(j + 1) * A.lbuf_words, IAF_W * (int)A.RS * 4);
        const unsigned char* lrd = lds8 + (j & 1) * LBUF_BYTES;
        unsigned char* lwr = lds8 + ((j + 1) & 1) * LBUF_BYTES;
        const float* tail = tails + (j & 1) * TAIL_WORDS;
        const float* bg = tail + q * 16;
        const float* br = bg + 64;
        const bool has_next = j + 1 < A.nlayers;

        f4 acc[4];
        wn_u4 ch[2], cl[2];                                         // tap t of my column: also the residual's skip input
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) ch[s2] = cl[s2] = (wn_u4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = cn[mb];
        if (mine) {
            // left neighbours whose layer j-1 output I read: columns [seg_col + 16 wave - 2d, seg_col) of buffer j
            if (j > 0 && c > 0) {
                const int reach = 2 * d - 16 * wave;
                if (reach > 0) {
                    const int per = 16 * max(base, 1);              // every left neighbour of the pass has >= base blocks
                    const int nn = min(c, (reach + per - 1) / per);
                    if (!wait_left(rf, rerr, c, nn, A.epoch + (unsigned)j, lane)) return;
                }
            }
            STAMP(1);
            // taps: source column lc - sh; inside the segment -> LDS, left of it -> the global buffer of layer j-1
            // (which also holds the zeros left of the utterance).  Remote words of both taps are requested up
            // front; tap t (always local) runs first and covers their latency.
            wn_u4 rh[2][2], rl[2][2];                               // [tap t-2d, t-d][channel half]
#pragma unroll
            for (int tp = 0; tp < 2; ++tp) {
                const int sh = (2 - tp) * d;
                const int vo = (q * RS16 + (IAF_LP + seg_col + lc - sh) * 16) | (lc - sh < 0 ? 0 : OOB);
#pragma unroll
                for (int sx = 0; sx < 2; ++sx) {
                    rh[tp][sx] = buf_ld4<SC1>(rin, vo, (4 * sx) * RS16);
                    rl[tp][sx] = buf_ld4<SC1>(rin, vo, (8 + 4 * sx) * RS16);
                }
            }
            // only now the bulk prefetches: a progress poll queues behind whatever this CU has in flight
            if (j > 0) dma(j, IAF_P_FLOATS, prw, IAF_PR_FLOATS);   // residual fragments: land during the K loop
            if (has_next) c_tile(j + 1, cn);                        // next layer's C tile: a whole layer ahead
            constexpr int ORD[6] = {4, 5, 2, 3, 0, 1};
            wn_u4 a[2][4][2];
            wn_u4 bh[2], bl[2];                                     // B operand of the current / next K-step
            auto tap_b = [&](int ks, wn_u4& h_, wn_u4& l_) {
                const int tp = ks >> 1, sx = ks & 1;
                const int src = lc - (2 - tp) * d;
                const int lsrc = src < 0 ? 0 : src;
                h_ = *reinterpret_cast<const wn_u4*>(lrd + (4 * sx + q) * LROW + lsrc * 16);
                l_ = *reinterpret_cast<const wn_u4*>(lrd + (8 + 4 * sx + q) * LROW + lsrc * 16);
            };
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                a[0][mb][0] = Pl[((ORD[0] * 4 + mb) * 2 + 0) * 64];
                a[0][mb][1] = Pl[((ORD[0] * 4 + mb) * 2 + 1) * 64];
            }
            tap_b(ORD[0], bh[0], bl[0]);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int ks = ORD[i], tp = ks >> 1, sx = ks & 1;
                if (i + 1 < 6) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        a[(i + 1) & 1][mb][0] = Pl[((ORD[i + 1] * 4 + mb) * 2 + 0) * 64];
                        a[(i + 1) & 1][mb][1] = Pl[((ORD[i + 1] * 4 + mb) * 2 + 1) * 64];
                    }
                    tap_b(ORD[i + 1], bh[(i + 1) & 1], bl[(i + 1) & 1]);
                }
                wn_u4 ch_ = bh[i & 1], cl_ = bl[i & 1];
                if (tp == 2) {
                    ch[sx] = ch_;
                    cl[sx] = cl_;
                }
                if (tp < 2) {
                    const bool remote = lc - (2 - tp) * d < 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ch_[e] = remote ? rh[tp][sx][e] : ch_[e];
                        cl_[e] = remote ? rl[tp][sx][e] : cl_[e];
                    }
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma3(a[i & 1][mb][0], a[i & 1][mb][1], ch_, cl_, acc[mb]);
            }
        }
        if (!mine && j > 0) dma(j, IAF_P_FLOATS, prw, IAF_PR_FLOATS);
        STAMP(2);
        // everybody is done with the dilated fragments of layer j (replaced under the epilogue); the residual
        // fragments of layer j have landed
        WN_WAIT_VM0();
        STAMP(3);
        __syncthreads();
        STAMP(4);
        wn_u4 ftl = (wn_u4){0u, 0u, 0u, 0u};
        if (has_next) {
            dma(j + 1, 0, frag, FRAG_WORDS);
            ftl = tail_load(j + 1);
        }
        if (mine) {
            // epilogue (parallel_wavenet.py:246-254): gate, residual 1x1, add, split, store
            const float inv_m = tail[128], inv_r = tail[129];
            float g[2][4];
#pragma unroll
            for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // sigmoid(u) * tanh(v) = (e^{2v} - 1) / ((1 + e^{-u}) (e^{2v} + 1)): two exp, ONE rcp
                    const float u = fmaf(acc[mg][r], inv_m, bg[mg * 4 + r]);
                    const float v = fminf(fmaxf(fmaf(acc[mg + 2][r], inv_m, bg[(mg + 2) * 4 + r]), -15.f), 15.f);
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
            f4 brv[4];                       // residual biases: in registers before the residual MFMAs start
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                brv[mb] = *reinterpret_cast<const f4*>(br + mb * 4);
                asm volatile("" : "+v"(brv[mb]));
            }
            wn_u4 oh[2], ol[2];
            f4 rcs[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                rcs[mb] = mfma3(PRl[(mb * 2 + 0) * 64], PRl[(mb * 2 + 1) * 64], gh, gl, (f4){0.f, 0.f, 0.f, 0.f});
            // OPEN DEFECT (why this form is withheld, see the file header): rarely the FIRST accumulator register of one
            // of these chains reads back as ~0 in lanes 48-63 of one wave (the block's output for that channel is then
            // skip + bias, all 16 columns).  The 64 wait states below removed it for fully occupied workgroups
            // (12 active waves) but not for partly filled ones.
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f4 rc = rcs[mb];
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    float l0, l1;
                    wn_join_pair(ch[mb >> 1][(mb & 1) * 2 + rp], cl[mb >> 1][(mb & 1) * 2 + rp], l0, l1);
                    const float v0 = l0 + fmaf(rc[2 * rp], inv_r, brv[mb][2 * rp]);
                    const float v1 = l1 + fmaf(rc[2 * rp + 1], inv_r, brv[mb][2 * rp + 1]);
                    unsigned hw, lw;
                    wn_split_pair(v0, v1, hw, lw);
                    oh[mb >> 1][(mb & 1) * 2 + rp] = hw;
                    ol[mb >> 1][(mb & 1) * 2 + rp] = lw;
                }
            }
            const int vo_out = q * RS16 + (IAF_LP + seg_col + lc) * 16;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                st4_sc1(oh[s2], rout, vo_out, (4 * s2) * RS16);
                st4_sc1(ol[s2], rout, vo_out, (8 + 4 * s2) * RS16);
                *reinterpret_cast<wn_u4*>(lwr + (4 * s2 + q) * LROW + lc * 16) = oh[s2];
                *reinterpret_cast<wn_u4*>(lwr + (8 + 4 * s2 + q) * LROW + lc * 16) = ol[s2];
            }
        }
        if (has_next && threadIdx.x < 33) reinterpret_cast<wn_u4*>(tails + ((j + 1) & 1) * TAIL_WORDS)[threadIdx.x] = ftl;
        STAMP(5);
        // my stores of layer j are acknowledged and the next fragments have landed: publish "j + 1 layers done"
        WN_WAIT_VM0();
        STAMP(6);
        if (mine && has_next) st1_sc1(A.epoch + (unsigned)j + 1u, rf, own_flag);
        // layer j's output in LDS and the fragments / tail of layer j + 1 are visible
        __syncthreads();
        STAMP(7);
    }
    // the launch ends here: the kernel boundary publishes the last layer
}

}  // namespace

// ---------------------------------------------------------------------------
int wn_iaf_r_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_res_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, RES_LDS_BYTES));
    return WN_OK;
}

int wn_iaf_r_max_cols(const wn_handle* h) { return h->num_cu * SEG; }
int wn_iaf_r_max_layers() { return RES_MAX_LAYERS; }

// Layers [0, n) of flow `fp` of ONE utterance over the pass [col0, col0 + 16 nblk).  lbuf: n + 1 activation buffers
// (buffer 0 = the flow's start-conv output, complete); C: the utterance's hoisted conditioning at the flow's
// first row block (one row block per layer, rb_floats apart).
int wn_iaf_r_flow(wn_handle* h, const IafFlowPack& fp, const float* C, int64_t rb_floats, unsigned* lbuf, int64_t RS,
                  int col0, int nblk, unsigned* flags, unsigned epoch, hipStream_t st) {
    if ((int)fp.layers.size() > RES_MAX_LAYERS) return wn_fail(h, WN_EINVAL, "flow has more than %d layers", RES_MAX_LAYERS);
    if (nblk > h->num_cu * RW) return wn_fail(h, WN_EINVAL, "pass of %d blocks exceeds the resident capacity", nblk);
    ResArgs A;
    A.blob = h->d_blob;
    A.C = C;
    A.rb_floats = rb_floats;
    A.lbuf = lbuf;
    A.lbuf_words = (long long)IAF_W * RS;
    A.flags = flags;
    A.flag_words = h->num_cu * 16 + 1;      // the workspace holds 16 words per workgroup; the error word sits behind them
    A.epoch = epoch;
    A.RS = RS;
    A.col0 = col0;
    A.nblk = nblk;
    A.nlayers = (int)fp.layers.size();
    A.dbg = nullptr;
    A.dbg_c = 0;
    static unsigned long long* dbg_dev = nullptr;
    const char* de = getenv("WN_RES_DEBUG");
    if (de) {
        if (!dbg_dev) (void)hipMalloc(&dbg_dev, RES_MAX_LAYERS * 8 * sizeof(unsigned long long));
        A.dbg = dbg_dev;
        A.dbg_c = atoi(de);
    }
    for (size_t i = 0; i < fp.layers.size(); ++i) {
        A.layers[i].w_off = (unsigned)fp.layers[i].off_h;
        A.layers[i].d = fp.layers[i].dilation;
    }
    hipLaunchKernelGGL(iaf_res_kernel, dim3(h->num_cu), dim3(64 * RW), RES_LDS_BYTES, st, A);
    if (de) {      // dev aid: stamps of workgroup WN_RES_DEBUG, wave 0 (shader cycles / 100), one line per layer:
                   // wait-left | K loop | residual-fragment wait | barrier | epilogue | ack + DMA wait | barrier
        unsigned long long hb[RES_MAX_LAYERS * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hb, dbg_dev, sizeof(hb), hipMemcpyDeviceToHost);
        fprintf(stderr, "res col0 %d nblk %d layers %d:", col0, nblk, A.nlayers);
        for (int j = 0; j < A.nlayers; ++j) {
            fprintf(stderr, "\n  L%02d d%3d", j, A.layers[j].d);
            for (int i = 1; i < 8; ++i) fprintf(stderr, " %6.2f", (double)(hb[j * 8 + i] - hb[j * 8 + i - 1]) * 0.01);
        }
        fprintf(stderr, "\n");
    }
    return WN_OK;
}
