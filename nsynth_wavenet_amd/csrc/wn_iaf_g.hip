// Layer groups resident in LDS: up to five residual layers of the IAF student per launch, on hoisted conditioning.
//
// The single-layer launches of wn_iaf_c.hip are priced by what every launch pays again -- ~5 us of start-up,
// dispatch and drain for ~3 us of matrix work per CU -- and every layer boundary is a round trip
// of the residual stream l through the fabric.  Here a workgroup owns a SEGMENT of time and runs a whole group of
// layers on it with l in LDS:
//
//   * natural group ("nat"): layers with dilations d1..dk, 2 * sum(d) <= 64 -- (1, 2, 4, 8, 16) of every cycle.  The
//     causal halo of the group (62 samples) is RECOMPUTED: a segment of 24 sixteen-sample blocks loads 4 halo blocks
//     + 20 output blocks, layer j computes only the blocks later layers still need, nothing is exchanged between
//     workgroups (no flags, no polls, no acknowledgements: the hand-offs that sank the round-2 single-launch forms).
//   * decimated group ("dec"): layers whose dilations are all multiples of 32 -- (32, ..., 512).  On the residue class
//     t = r (mod 32) these layers ARE a dilation-(1, 2, 4, 8, 16) group of the 32x shorter sequence, so the SAME
//     kernel runs them on a decimated view: the natural group writes its output in "DL" layout ([residue][time / 32],
//     16-byte words scattered once), the decimated group reads contiguous rows, and writes natural order back.
//
// A ten-layer dilation cycle is two launches instead of eight, the residual stream crosses the fabric twice per
// cycle instead of eight times, and the only traffic that scales with the layer count is the hoisted term C
// (256 B per sample and layer, read once).  Hoisted rows of decimated layers are written by the conditioning GEMM
// in decimated block order (iaf_cond_h_kernel, wn_iaf_c.hip), so a C block is one contiguous 4 KB read here too.
//
// Reference semantics: parallel_wavenet.py:227-254 (layer), :222-225 (start conv, FIRST), :256-277 / :319-324
// (flow head, LAST); masked.py:160-232 (causal dilated conv with zero history).
//
// LDS map (161 296 of 163 840 B): l as [plane hi | lo][block -1 .. 23][group 8][column 16] x 16 B (block -1 is a
// zero block: taps left of the segment), one layer's dilated-conv fragments (48 KB, reused for the head image), one
// layer tail (residual fragments, biases, scales), the start-conv weights.
#include <algorithm>
#include <cstdlib>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_mfma_h.h"
#include "wn_iaf_c.h"


namespace {

// Twelve waves (three per SIMD) of two blocks each: measured 2-4 % faster than eight waves of three blocks at every
// batch size (A/B on one box: 64.4 vs 66.4 us per launch at one utterance) -- the K loop hides its LDS latency better.
constexpr int GK_WAVES = 12, GK_HN = 2, GK_NBLK = GK_WAVES * GK_HN, GK_THREADS = GK_WAVES * 64;
static_assert(GK_NBLK == 24, "segment of 24 blocks");
constexpr int GK_BLK_BYTES = 2048;                             // one plane of one block: [group 8][column 16] x 16 B
constexpr int GK_PLANE = (GK_NBLK + 1) * GK_BLK_BYTES;         // 51 200: blocks -1 .. 23
constexpr int GK_A_OFF = 2 * GK_PLANE;                         // dilated-conv fragments of the current layer
constexpr int GK_T_OFF = GK_A_OFF + LC_A_WORDS * 4;            // tail of the current layer
constexpr int GK_S_OFF = GK_T_OFF + LC_TAIL_WORDS * 4;         // start-conv weights (FIRST)
constexpr int GK_LDS_BYTES = GK_S_OFF + IAF_START_LDS_WORDS * 4;
static_assert(GK_LDS_BYTES <= 160 * 1024, "LDS budget of one CU");
static_assert(GK_S_OFF % 16 == 0 && GK_T_OFF % 16 == 0, "");
static_assert(HC_LDS_WORDS <= LC_A_WORDS, "the head image takes over the fragment buffer");
constexpr int GK_MAXL = 5;
constexpr int GK_ST = LC_A_WORDS / 4 / GK_THREADS;           // 16-byte pieces of the next image a thread stages
static_assert(GK_ST * GK_THREADS * 4 == LC_A_WORDS && (LC_TAIL_WORDS + 3) / 4 <= GK_THREADS && HC_TAIL_WORDS <= LC_TAIL_WORDS && HC_A_WORDS <= LC_A_WORDS, "staging shares");
constexpr int GK_DEC = 32;                                     // decimation of a "dec" group
constexpr int GK_PADJ = 64;                                    // zero columns in front of every residue row of the DL layout

struct GLayer {
    const unsigned* w;      // layer image (fragments at 0, tail at IAF_P_FLOATS words)
    const float* C;         // hoisted row block of this layer (batch row 0)
    int d;                  // dilation in the group's own time base (1 .. 16)
    int first;              // first LDS block later layers (or the output) still need from this layer
};

struct GArgs {
    const unsigned* lin;
    unsigned* lout;
    int64_t RS;             // 16-byte words per (plane, group) row of l
    int64_t c_bstride;      // floats of C per batch row
    GLayer L[GK_MAXL];
    int nl;
    int in_dec, out_dec;    // layout of the input / output residual stream: 0 natural, 1 DL
    int hb, nb_out;         // halo blocks, output blocks per segment (hb + nb_out = 24)
    int segs;               // segments per unit (unit = utterance, or one residue row of an utterance)
    int nbu;                // 16-column blocks per unit
    int nb_tot;             // 16-column blocks per utterance (T / 16): extent of one row block of C
    int ntasks;
    int RJ;                 // DL: 16-byte words per residue row (GK_PADJ + T / 32)
    // FIRST: the group opens a flow, its input is start_conv(shift_right(x))
    const float* x;
    int XR;
    const float* wstart;
    // LAST: the group closes a flow, the flow head runs on its output
    const float* Ch;
    const unsigned* whead;
    const float* xin;       // flow input ...
    float* xout;            // ... and output x <- x * s + mean: the same array, except when the flow is ONE group (FIRST
                            // and LAST in one launch): neighbouring segments still read x[t-3 .. t-1] for their start conv
    float* Mt;
    float* St;
    int64_t T;
    int first_flow;
    unsigned* status;
};

// LDS-DMA the compiler does not see (16 B per lane: lane i's bytes land at lds_byte_addr + 16 i).  Through the builtin
// every later LDS access of the wave is preceded by s_waitcnt vmcnt(0) -- the compiler cannot tell which LDS bytes the
// DMA writes -- which turns the asynchronous load of the next layer's fragments into a 1-2 us stall at the start of the
// epilogue.  Completion is the caller's business: an explicit s_waitcnt vmcnt(0) before the barrier that publishes it.
__device__ inline void g_dma16(const unsigned* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}
// `words` (multiple of 4) from src to LDS, 1 KB per instruction, spread over the waves
__device__ inline void g_dma_image(const unsigned* src, unsigned dst, int words, int wave, int lane, int nwaves = GK_WAVES) {
    const int full = words >> 8, rem = (words & 255) >> 2;
    for (int i = wave; i < full; i += nwaves) g_dma16(src + (size_t)(i * 64 + lane) * 4, dst + i * 1024);
    if (rem && wave == full % nwaves && lane < rem) g_dma16(src + (size_t)(full * 64 + lane) * 4, dst + full * 1024);
}
__device__ inline void g_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... all but the N youngest requests of the wave
template <int N>
__device__ inline void g_dma_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(GK_THREADS, 1) void iaf_group_kernel(const GArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    char* lds = reinterpret_cast<char*>(ldsw);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int RS16 = (int)A.RS * 16;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    float amax = 0.f;

    // zero block -1 of both planes (never written afterwards)
    if (threadIdx.x < 256)
        *reinterpret_cast<wn_u4*>(lds + (threadIdx.x >> 7) * GK_PLANE + (threadIdx.x & 127) * 16) = (wn_u4){0u, 0u, 0u, 0u};
    if (FIRST) stage_start_weights(A.wstart, reinterpret_cast<f4*>(lds + GK_S_OFF));

    // task walk: every XCD takes one contiguous eighth of the tasks (neighbouring segments share their halo in its L2;
    // a decimated group's 32 residues of one time range share the 64-byte lines of the scattered writes)
    int first, end, step;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, per = (A.ntasks + 7) >> 3;
        first = xcd * per + (int)(blockIdx.x >> 3);
        end = min(A.ntasks, (xcd + 1) * per);
        step = (int)(gridDim.x >> 3);
    } else {
        first = blockIdx.x;
        end = A.ntasks;
        step = gridDim.x;
    }

    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(lds + GK_A_OFF) + lane;     // [((ks*4+mb)*2+plane)*64]
    const float* tailf = reinterpret_cast<const float*>(lds + GK_T_OFF);
    PairLayer W;
    W.Pl = Pl;
    W.PRl = reinterpret_cast<const wn_u4*>(lds + GK_T_OFF) + lane;
    W.bg = tailf + IAF_PR_FLOATS + q * 16;
    W.br = W.bg + 64;
    const int own = q * 256 + n * 16;                     // byte offset of a lane's word inside a block plane (group q)

    for (int task = first; task < end; task += step) {
        int b, seg, r = 0;
        if (A.in_dec) {
            r = task & (GK_DEC - 1);
            const int u = task >> 5;
            b = u / A.segs;
            seg = u - b * A.segs;
        } else {
            b = task / A.segs;
            seg = task - b * A.segs;
        }
        const int blk0 = seg * A.nb_out - A.hb;                                   // unit block index of LDS block 0
        const int cblk0 = (A.in_dec ? r * A.nbu : 0) + blk0;                      // its block index inside a row block of C
        const int qcol0 = (A.in_dec ? r * A.RJ + GK_PADJ : IAF_LP) + 16 * blk0;   // its 16-byte column in a row of lin
        // LDS blocks of this wave: wave + 12 e.  The blocks a late layer skips (its `first`: the halo, blocks 0 .. 3 for
        // the fifth layer) then belong to four different waves on four different SIMDs, so every SIMD loses one of its six
        // blocks there; with adjacent blocks per wave two SIMDs would skip two and the barrier would wait for the others.
        // (Round 4, measured and not kept: those four blocks given to waves 8 .. 11, the youngest wave of every SIMD, whose
        // epilogue nothing hides: 58.1 against 57.7 us per launch; uneven shares -- three blocks for the oldest wave of a
        // SIMD, two, one for the youngest: 92 us, the three-block waves spill at 168 registers.)
        auto blk_of = [&](int e) { return wave + GK_WAVES * e; };
        auto active = [&](int i) { return (unsigned)(blk0 + i) < (unsigned)A.nbu; };
        // time of column n of LDS block i
        auto time_of = [&](int i) -> int { return A.in_dec ? r + GK_DEC * (16 * (blk0 + i) + n) : 16 * (blk0 + i) + n; };

        // The accumulators of a layer START from its hoisted tile: requested as soon as the previous layer's epilogue
        // arithmetic has released the registers (layer 0: here), one 16-byte load per row block.
        f4 acc[GK_HN][4];
        auto load_c = [&](const float* Cbase) {
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(Cbase + (size_t)b * A.c_bstride), 0, A.nb_tot * 4096, 0x00020000);
#pragma unroll
            for (int e = 0; e < GK_HN; ++e)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)      // blocks outside the row block fall outside the descriptor: zeros
                    acc[e][mb] = buf_ldf4(rc, (cblk0 + blk_of(e)) * 4096 + lane * 16, mb * 1024);
        };
        // ---- prologue: l segment, first layer's image, first C tiles ----
        if (FIRST) {
            // l0 = start_conv(shift_right(x)) (parallel_wavenet.py:222-225), zero left of the utterance
            const f4* wq = reinterpret_cast<const f4*>(lds + GK_S_OFF);
            __syncthreads();                               // start weights staged (and the previous task's readers are done)
            const float* xb = A.x + (size_t)b * A.XR + IAF_XP;
#pragma unroll
            for (int e = 0; e < GK_HN; ++e) {
                const int i = blk_of(e), t = time_of(i);
                const bool on = active(i);
                float x0 = 0.f, x1 = 0.f, x2 = 0.f;
                if (on) { x0 = xb[t - 3]; x1 = xb[t - 2]; x2 = xb[t - 1]; }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    wn_u4 hw, lw;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 2 * (16 * s + 8 * (k >> 1) + 2 * q + (k & 1));
                        const f4 wa = wq[c], wb = wq[c + 1];
                        float v0 = wa[3] + wa[0] * x0 + wa[1] * x1 + wa[2] * x2;
                        float v1 = wb[3] + wb[0] * x0 + wb[1] * x1 + wb[2] * x2;
                        if (!on) v0 = v1 = 0.f;
                        unsigned a, c2;
                        wn_split_pair_t(v0, v1, a, c2, amax);
                        hw[k] = a;
                        lw[k] = c2;
                    }
                    *reinterpret_cast<wn_u4*>(lds + (i + 1) * GK_BLK_BYTES + s * 1024 + own) = hw;
                    *reinterpret_cast<wn_u4*>(lds + GK_PLANE + (i + 1) * GK_BLK_BYTES + s * 1024 + own) = lw;
                }
            }
        } else {
            __syncthreads();                               // the previous task's readers of l are done
            const unsigned* src = A.lin + (size_t)b * IAF_W * A.RS;
            for (int c = wave; c < 4 * GK_NBLK; c += GK_WAVES) {
                const int pl = c / (2 * GK_NBLK), rem = c - pl * 2 * GK_NBLK, blk = rem >> 1, g0 = (rem & 1) * 4;
                // right of the unit: the zero pad at the start of the row (left of it the pads are zero by themselves)
                const int col = blk0 + blk < A.nbu ? qcol0 + 16 * blk + n : 0;
                g_dma16(src + ((size_t)(pl * 8 + g0 + q) * A.RS + col) * 4, lds_base + pl * GK_PLANE + (blk + 1) * GK_BLK_BYTES + g0 * 256);
            }
        }
        g_dma_image(A.L[0].w, lds_base + GK_A_OFF, LC_A_WORDS, wave, lane);
        g_dma_image(A.L[0].w + IAF_P_FLOATS, lds_base + GK_T_OFF, LC_TAIL_WORDS, wave, lane);
        load_c(A.L[0].C);
        // the segment and the first image have landed (LDS-DMA the compiler does not track: an explicit wait); the eight tile
        // loads are the youngest requests of the wave and vmcnt retires in order, so they stay in flight across the barrier
        g_dma_wait_but<4 * GK_HN>();
        __syncthreads();

        for (int j = 0; j < A.nl; ++j) {
            const bool fin = j + 1 == A.nl;
            const int d = A.L[j].d;
            const bool run = blk_of(GK_HN - 1) >= A.L[j].first;   // any block of this wave still needed from this layer
            // ---- the NEXT image (the next layer's fragments and tail, or the head's) is requested in front of the K loop,
            // where the vector-memory port is idle, and staged through registers: it cannot go to LDS before the last K loop
            // of this layer has read the current one, and an LDS-DMA issued at that point would put its 2-3 k cycles of
            // latency between the two barriers below
            const unsigned* s_img = nullptr;
            const unsigned* s_tail = nullptr;
            int n_img = 0, n_tail = 0;
            unsigned d_tail = GK_T_OFF;
            if (!fin) {
                s_img = A.L[j + 1].w;
                s_tail = s_img + IAF_P_FLOATS;
                n_img = LC_A_WORDS / 4;
                n_tail = (LC_TAIL_WORDS + 3) / 4;
            } else if (LAST) {
                s_img = A.whead;
                s_tail = s_img + IAF_PH_FLOATS;
                n_img = HC_A_WORDS / 4;
                n_tail = (HC_TAIL_WORDS + 3) / 4;
                d_tail = GK_A_OFF + HC_A_WORDS * 4;
            }
            wn_u4 st[GK_ST], stt = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < GK_ST; ++i) {
                const int idx = i * GK_THREADS + (int)threadIdx.x;
                st[i] = (wn_u4){0u, 0u, 0u, 0u};
                if (idx < n_img) st[i] = *reinterpret_cast<const wn_u4*>(s_img + (size_t)idx * 4);
            }
            if ((int)threadIdx.x < n_tail) stt = *reinterpret_cast<const wn_u4*>(s_tail + (size_t)threadIdx.x * 4);

            // ---- K loop: dilated conv of this wave's blocks on top of the hoisted tile ----
            if (run) {
                // B operands: column 16 i + n - shift of the layer input, one 16-byte LDS word per (tap, half, plane)
                int ba[GK_HN][3];
#pragma unroll
                for (int e = 0; e < GK_HN; ++e)
#pragma unroll
                    for (int tap = 0; tap < 3; ++tap) {
                        const int c = 16 * blk_of(e) + n - (2 - tap) * d;
                        ba[e][tap] = (max(c >> 4, -1) + 1) * GK_BLK_BYTES + q * 256 + (c & 15) * 16;
                    }
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {
                    wn_u4 a[4][2];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        a[mb][0] = Pl[((ks * 4 + mb) * 2 + 0) * 64];
                        a[mb][1] = Pl[((ks * 4 + mb) * 2 + 1) * 64];
                    }
#pragma unroll
                    for (int e = 0; e < GK_HN; ++e) {
                        const wn_u4 bh = *reinterpret_cast<const wn_u4*>(lds + ba[e][ks >> 1] + (ks & 1) * 1024);
                        const wn_u4 bl = *reinterpret_cast<const wn_u4*>(lds + GK_PLANE + ba[e][ks >> 1] + (ks & 1) * 1024);
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) acc[e][mb] = mfma3(a[mb][0], a[mb][1], bh, bl, acc[e][mb]);
                    }
                }
            }
            bool on[GK_HN];
#pragma unroll
            for (int e = 0; e < GK_HN; ++e) on[e] = blk_of(e) >= A.L[j].first && active(blk_of(e));
            wn_u4 oh[GK_HN][2], ol[GK_HN][2];
            // Epilogue arithmetic of the wave's blocks.  Nothing is written in place here -- other waves' K loops may still
            // read this layer's input -- the outputs wait in registers.  The accumulators are free afterwards: the next
            // layer's (or the head's) hoisted tile is requested into them.
            auto epi_math = [&]() {
                W.inv_m = tailf[IAF_PR_FLOATS + 128];
                W.inv_r = tailf[IAF_PR_FLOATS + 129];
#pragma unroll
                for (int e = 0; e < GK_HN; ++e) {
                    if (!on[e]) continue;                      // not needed / outside the utterance (stays zero)
                    const char* blk = lds + (blk_of(e) + 1) * GK_BLK_BYTES + own;
                    wn_u4 lh[2], ll[2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        lh[s2] = *reinterpret_cast<const wn_u4*>(blk + s2 * 1024);
                        ll[s2] = *reinterpret_cast<const wn_u4*>(blk + GK_PLANE + s2 * 1024);
                    }
                    pair_epilogue(W, acc[e], lh, ll, oh[e], ol[e], amax);
                }
                if (!fin) load_c(A.L[j + 1].C);
                else if (LAST) load_c(A.Ch);
            };
            // The waves of a SIMD get the matrix pipe in the order of their age, so a wave that runs its epilogue arithmetic
            // STRAIGHT behind its own K loop -- no barrier in between -- does it in the shadow of the younger waves' MFMAs
            // (profiles/r04_issue_overlap_ubench.txt: VALU work issues beside a running MFMA on gfx950, within a wave and
            // between waves; r04_group_kernel_stamps.txt: the oldest wave of a SIMD is through its K loop after 4.3 k cycles,
            // the youngest after 8.5 k).  Only the youngest wave's arithmetic is left exposed.  (Putting the barrier behind
            // the K loops instead, so that the older waves write their outputs while the youngest still computes, was
            // measured much slower -- 70 against 58 us per launch: the write burst starves the youngest waves' LDS reads.)
            epi_math();
            if (fin && !LAST) {
                // the group's output leaves from the registers: natural order, or scattered once into the DL layout of the
                // next (decimated) group.  Nothing of this layer goes back to LDS.
                const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(A.lout + (size_t)b * IAF_W * A.RS), 0, IAF_W * (int)A.RS * 4, 0x00020000);
#pragma unroll
                for (int e = 0; e < GK_HN; ++e) {
                    if (!on[e]) continue;
                    const int t = time_of(blk_of(e));
                    const int col = A.out_dec ? (t & (GK_DEC - 1)) * A.RJ + GK_PADJ + (t >> 5) : IAF_LP + t;
                    const int vo = q * RS16 + col * 16;
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        buf_st4<WN_G_ST_AUX>(oh[e][s2], ro, vo, (4 * s2) * RS16);
                        buf_st4<WN_G_ST_AUX>(ol[e][s2], ro, vo, (8 + 4 * s2) * RS16);
                    }
                }
                continue;
            }
            __syncthreads();                               // every K loop has read this layer's input and fragments, every
                                                           // epilogue this layer's tail
            // in-place update of the wave's blocks (LAST: the head below reads its input from here) ...
#pragma unroll
            for (int e = 0; e < GK_HN; ++e) {
                if (!on[e]) continue;
                char* blk = lds + (blk_of(e) + 1) * GK_BLK_BYTES + own;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    *reinterpret_cast<wn_u4*>(blk + s2 * 1024) = oh[e][s2];
                    *reinterpret_cast<wn_u4*>(blk + GK_PLANE + s2 * 1024) = ol[e][s2];
                }
            }
            // ... the staged fragments (only K loops read that buffer: all done) ...
#pragma unroll
            for (int i = 0; i < GK_ST; ++i) {
                const int idx = i * GK_THREADS + (int)threadIdx.x;
                if (idx < n_img) *reinterpret_cast<wn_u4*>(lds + GK_A_OFF + idx * 16) = st[i];
            }
            // ... and the staged tail
            if ((int)threadIdx.x < n_tail) *reinterpret_cast<wn_u4*>(lds + d_tail + threadIdx.x * 16) = stt;
            __syncthreads();                               // layer output and next image in LDS
        }

        if (LAST) {
            // ---- flow head on the group's output (same arithmetic as iaf_layer_c_kernel<1, true>) ----
            const wn_u4* PHl = Pl;                          // the head image sits in the fragment buffer now
            const float* hf = reinterpret_cast<const float*>(lds + GK_A_OFF) + HC_A_WORDS;
            const float* bo = hf + q * 16;
            const float* wm = bo + 64;
            const float* wsc = wm + 64;
            const float bmean = hf[192], bscale = hf[193], inv_h = hf[194];
#pragma unroll
            for (int e = 0; e < GK_HN; ++e) {
                const int i = blk_of(e);
                if (i < A.hb || !active(i)) continue;
                f4 hacc[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) hacc[mb] = acc[e][mb];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    // the group's output block, written in place by this lane in the last layer's epilogue
                    const char* blk = lds + (i + 1) * GK_BLK_BYTES + own + ks * 1024;
                    wn_u4 bh = *reinterpret_cast<const wn_u4*>(blk), bl = *reinterpret_cast<const wn_u4*>(blk + GK_PLANE);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {             // relu(l) (:256) on the reconstructed value
                        float v0, v1;
                        wn_join_pair(bh[k], bl[k], v0, v1);
                        unsigned hw, lw;
                        wn_split_pair(fmaxf(v0, 0.f), fmaxf(v1, 0.f), hw, lw);
                        bh[k] = hw;
                        bl[k] = lw;
                    }
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        hacc[mb] = mfma3(PHl[((ks * 4 + mb) * 2 + 0) * 64], PHl[((ks * 4 + mb) * 2 + 1) * 64], bh, bl, hacc[mb]);
                }
                float pm = 0.f, ps = 0.f;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float o = fmaxf(fmaf(hacc[mb][rr], inv_h, bo[mb * 4 + rr]), 0.f);
                        pm = fmaf(wm[mb * 4 + rr], o, pm);
                        ps = fmaf(wsc[mb * 4 + rr], o, ps);
                    }
                pm += __shfl_xor(pm, 16);
                ps += __shfl_xor(ps, 16);
                pm += __shfl_xor(pm, 32);
                ps += __shfl_xor(ps, 32);
                if (q == 0) {
                    const int64_t t = time_of(i);
                    const float mean = pm + bmean;
                    const float sc = fminf(fmaxf(softplus_tf(ps + bscale), EXP_M9), EXP_7);   // :105-114
                    const size_t xo = (size_t)b * A.XR + IAF_XP + t;
                    A.xout[xo] = A.xin[xo] * sc + mean;                                       // :277
                    float* mp = A.Mt + (size_t)b * A.T + t;
                    float* sp = A.St + (size_t)b * A.T + t;
                    if (A.first_flow) { *mp = mean; *sp = sc; }
                    else { *mp = mean + *mp * sc; *sp = *sp * sc; }                           // :322-323
                }
            }
        }
    }
    wn_range_flag(amax, A.status);
}


}  // namespace


int wn_iaf_g_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_group_kernel<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS_BYTES));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_group_kernel<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS_BYTES));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_group_kernel<false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS_BYTES));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_group_kernel<true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS_BYTES));
    return WN_OK;
}

// Groups of one flow: consecutive layers that one launch can run.  kind 0: natural (2 * sum(d) <= 64), kind 1:
// decimated (every d a multiple of 32 and 2 * sum(d / 32) <= 64).  Returns false when the flow cannot be covered
// by strictly alternating natural / decimated groups starting with a natural one (then lA keeps the natural and lB
// the DL layout for the whole call, see wn_iaf_generate) -- the caller falls back to the per-layer launches.
bool wn_iaf_g_plan(const std::vector<int>& dil, std::vector<WnGroup>& out) {
    out.clear();
    size_t i = 0;
    int kind = 0;
    while (i < dil.size()) {
        WnGroup g;
        g.kind = kind;
        g.begin = (int)i;
        int sum = 0;
        while (i < dil.size() && g.n() < GK_MAXL) {
            const int d = dil[i];
            if (kind == 1 && d % GK_DEC) break;
            const int dl = kind ? d / GK_DEC : d;
            if (dl < 1 || 2 * (sum + dl) > 64) break;
            sum += dl;
            ++i;
            g.end = (int)i;
        }
        if (g.n() == 0) return false;
        g.halo_cols = 2 * sum;
        out.push_back(g);
        kind ^= 1;
    }
    return !out.empty();
}

// One group launch.  layers: the group's packs; Cg: hoisted row block of its first layer (consecutive row blocks
// follow at rb_floats); the head's row block (last == true) follows the last layer's.
void wn_iaf_g_run(const wn_handle* h, const WnGroup& g, const IafLayerPack* layers, const float* Cg, size_t rb_floats,
                  int64_t c_bstride, const float* lin, float* lout, int64_t RS, int out_dec, int B, int64_t T,
                  const float* x, int XR, const float* wstart, bool last, const float* whead, const float* xin,
                  float* xout, float* Mt, float* St, int first_flow, unsigned* status, hipStream_t st) {
    GArgs A{};
    A.lin = reinterpret_cast<const unsigned*>(lin);
    A.lout = reinterpret_cast<unsigned*>(lout);
    A.RS = RS;
    A.c_bstride = c_bstride;
    A.nl = g.n();
    A.in_dec = g.kind;
    A.out_dec = out_dec;
    A.hb = std::max(2, ((g.halo_cols + 15) / 16 + 1) & ~1);          // even: a natural segment starts on a multiple of 32
    A.nb_out = GK_NBLK - A.hb;
    A.nb_tot = (int)(T / 16);
    A.nbu = g.kind ? (int)(T / (16 * GK_DEC)) : A.nb_tot;
    A.segs = (A.nbu + A.nb_out - 1) / A.nb_out;
    A.ntasks = B * A.segs * (g.kind ? GK_DEC : 1);
    A.RJ = GK_PADJ + (int)(T / GK_DEC);
    // the first LDS block layer j must produce: the output starts at column 16 hb and every later layer reaches
    // 2 d columns to the left
    int need = 16 * A.hb;
    for (int j = A.nl - 1; j >= 0; --j) {
        const int d = layers[g.begin + j].dilation / (g.kind ? GK_DEC : 1);
        A.L[j].w = reinterpret_cast<const unsigned*>(h->d_blob + layers[g.begin + j].off_h);
        A.L[j].C = Cg + (size_t)j * rb_floats;
        A.L[j].d = d;
        A.L[j].first = std::max(0, need) / 16;
        need -= 2 * d;
    }
    A.x = x;
    A.XR = XR;
    A.wstart = wstart;
    A.Ch = Cg + (size_t)A.nl * rb_floats;
    A.whead = reinterpret_cast<const unsigned*>(whead);
    A.xin = xin;
    A.xout = xout;
    A.Mt = Mt;
    A.St = St;
    A.T = T;
    A.first_flow = first_flow;
    A.status = status;
    int grid = std::min(A.ntasks, h->num_cu);
    if (grid >= 8) grid = std::min((grid + 7) / 8 * 8, std::max(8, h->num_cu / 8 * 8));   // XCD-aware walk: a multiple of 8
    auto kern = x ? (last ? iaf_group_kernel<true, true> : iaf_group_kernel<true, false>)
                  : (last ? iaf_group_kernel<false, true> : iaf_group_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GK_THREADS), GK_LDS_BYTES, st, A);
}
