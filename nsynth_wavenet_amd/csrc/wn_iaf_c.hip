// Split-fp16 IAF path with the conditioning GEMMs hoisted out of the residual layers.
//
// Every residual layer and every flow head adds a 1x1 projection of the SAME upsampled
// conditioning `enc` (256 channels) to its pre-activation (parallel_wavenet.py:240-244,
// :258-262); the reference itself evaluates those projections in bulk for the AR path
// (`Fastgen.cond_vars`, wavenet.py:353-377).  Reading `enc` in every layer costs
// 1024 B/sample/layer of the 1536 the fused layer kernel (wn_iaf_h.hip) moves.  Here one GEMM
// per deconv stack (iaf_cond_h_kernel) reads `enc` ONCE and writes, for every layer and head
// ("row block" = 64 output channels), the projected term C in the MFMA accumulator layout;
// the layer kernel then streams l (256 B read + 256 B write) and C (256 B read) per sample:
// 768 + 256 (C write) = 1024 B/sample/layer instead of 1536, and its weight image shrinks to
// 57 KB of LDS so two workgroups share a CU.
//
// C layout: [batch][row block][column block cb = t/16][mb = 16-row block][lane][4 floats] --
// exactly the D registers of v_mfma_f32_16x16x32_f16 (row 16mb + 4(lane>>4) + r, column
// 16cb + (lane&15)), so both sides move it with one 16-byte access per lane, 1 KB per wave.
// Values are the raw fp32 accumulators of the pre-scaled weights (same scale as the layer's
// dilated-conv fragments), so they are the C-in of the layer's first MFMA.
#include <algorithm>
#include <cstdlib>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_mfma_h.h"
#include "wn_iaf_c.h"

// Cache policy of the residual-stream stores of the layer kernels (aux: 2 = nt, 16 = sc1, 17 = sc0 sc1)
#ifndef WN_L_ST_AUX
#define WN_L_ST_AUX 0
#endif
#ifndef WN_ENC_STAGE_AUX
#define WN_ENC_STAGE_AUX 0
#endif

namespace {

constexpr int CK_NC = 128;                       // columns of one conditioning-GEMM tile
constexpr int CK_THREADS = 512;                  // 8 waves: two per SIMD hide each other's load / store latency
constexpr int CK_RS = CK_NC;                     // 16-byte words per (plane, group) row of the LDS tile: rows 2 KB apart, the
                                                 //   conflict-free kind of ds_read_b128 (scripts/ubench/lds_b128_stride.hip)
constexpr int CK_LDS_BYTES = 2 * 32 * CK_RS * 16;   // enc tile: [plane][group][column block][column] x 16 B = 128 KB
constexpr int CK_DEC = 32;                       // decimation of the row blocks that feed a "dec" layer group (wn_iaf_g.hip)

// LDS-DMA of 64 x 16 B (lane i's bytes land at lds_byte_addr + 16 i), issued from asm: through the builtin the compiler
// would put s_waitcnt vmcnt(0) in front of every later LDS access (wn_iaf_g.hip has the same helper and the reason)
__device__ inline void ck_dma16(const unsigned* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr)) : "memory");
}

// ---------------- conditioning GEMM: C[rb] = Wcond[rb] (64 x 256) . enc (256 x T) ----------------
// The enc tile of 128 columns sits in LDS as ready-made B operands; each wave owns whole row blocks: its A fragments
// come straight from L2 in fragment order (1 KB per wave load), every fragment is used against the 8 column blocks of
// the tile, every B operand read from LDS feeds 12 MFMAs.
// Row blocks come in two kinds (order[0 .. n_nat) natural, order[n_nat .. R) decimated):
//   natural   -- tile j = columns [128 j, 128 j + 128), column block nb = 16 consecutive samples, C block 8 j + nb;
//   decimated -- the layer (or head) runs in a decimated layer group of wn_iaf_g.hip, which walks the residue classes
//                t = r (mod 32): tile j = (k = j / 4, rg = j % 4) = residues 8 rg .. 8 rg + 7 x decimated columns
//                16 k .. 16 k + 15, column block nb = residue 8 rg + nb, C block (8 rg + nb) * (T / 512) + k.  The
//                tile is GATHERED from enc (64-byte runs: four residues x 16 B per decimated column) and holds the same
//                128 samples' worth of operands, so the MFMA loop is identical.
// Work list (round 5).  A UNIT is one sub-round: the eight waves of a workgroup take one row block each against one
// staged tile.  Units are ordered (kind, tile, sub-round) and every workgroup takes ONE CONTIGUOUS RANGE of them, equal to within one unit (4 800 units on
// 256 workgroups at one utterance: 18 or 19 each, where rounds of whole (tile, 32-row-block chunk) tasks cost the fullest
// workgroup 20).  A range crosses tile boundaries; the tile is restaged whenever (tile, kind) changes, by LDS-DMA: 16
// requests of 1 KB per wave issued back to back and ONE wait, instead of four batches of global loads -> registers -> LDS
// stores (the staging bubble was ~10 of the ~78 us of a four-sub-round task).
__global__ __launch_bounds__(CK_THREADS) void iaf_cond_h_kernel(
    const unsigned* __restrict__ enc, const unsigned* __restrict__ wblob, const unsigned* __restrict__ rb_off,
    const unsigned* __restrict__ order, int n_nat, float* __restrict__ C, int64_t c_bstride, int64_t TE, int c0, int R,
    int sr_nat, int sr_dec, int tiles_per_row, int ntiles, int64_t NCB) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    constexpr int NW = CK_THREADS / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const wn_u4* Bt = reinterpret_cast<const wn_u4*>(ldsw);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ldsw);
    const int NBD = (int)(NCB / CK_DEC);          // decimated column blocks per residue
    const int upt = sr_nat + sr_dec;              // units per tile
    const int64_t U = (int64_t)ntiles * upt;

    // contiguous unit range of this workgroup; every XCD takes one contiguous eighth (its L2 then holds one stretch of enc)
    int64_t u_lo, u_hi;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3, nx = gridDim.x >> 3;
        const int64_t x_lo = U * xcd / 8, x_hi = U * (xcd + 1) / 8;
        u_lo = x_lo + (x_hi - x_lo) * i / nx;
        u_hi = x_lo + (x_hi - x_lo) * (i + 1) / nx;
    } else {
        u_lo = U * blockIdx.x / gridDim.x;
        u_hi = U * (blockIdx.x + 1) / gridDim.x;
    }
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wblob, 0, 0x7ffffff0, 0x00020000);
    int staged_tile = -1, staged_kind = -1;
    // row block of this wave in unit u (clamped to the last one of its kind where a sub-round is short: `have` false)
    // unit u -> (kind, tile, sub-round): all natural units first, then the decimated ones -- with contiguous eighths per XCD
    // an XCD then works on ONE kind, and its L2 holds that kind's fragments (2 of the 4 MB) beside its stretch of enc
    // (with the kinds interleaved per tile the counters showed 655 MB of fragment re-fetches per call)
    const int64_t U_nat = (int64_t)ntiles * sr_nat;
    auto unit_of = [&](int64_t u, int& tile, int& s) -> bool {
        const bool dec = u >= U_nat;
        const int64_t r = dec ? u - U_nat : u;
        const int sr = dec ? sr_dec : sr_nat;
        tile = (int)(r / sr);
        s = (int)(r - (int64_t)tile * sr);
        return dec;
    };
    auto unit_rb = [&](int64_t u, bool& have) -> int {
        int tile, s;
        const bool dec = unit_of(u, tile, s);
        const int p_end = dec ? R : n_nat;
        const int pos = (dec ? n_nat : 0) + s * NW + wave;
        have = pos < p_end;
        return (int)order[have ? pos : p_end - 1];
    };
    wn_u4 a[2][4][2];
    if (u_lo < u_hi) {      // first A fragments of the wave's first row block: in flight while the tile is staged
        bool hv;
        const int ao = (int)rb_off[unit_rb(u_lo, hv)] * 4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            a[0][mb][0] = buf_ld4(rw, lane * 16, ao + (mb * 2 + 0) * 1024);
            a[0][mb][1] = buf_ld4(rw, lane * 16, ao + (mb * 2 + 1) * 1024);
        }
    }
    for (int64_t u = u_lo; u < u_hi; ++u) {
        int tile, su;
        const bool dec = unit_of(u, tile, su);
        const int b = tile / tiles_per_row;
        const int j = tile - b * tiles_per_row;
        bool have, have_next;
        const int rb = unit_rb(u, have);
        const int ao = (int)rb_off[rb] * 4;
        // the NEXT unit's first fragments are requested during this unit's last K-step (a short sub-round's idle wave: now)
        const int an = (int)rb_off[unit_rb(u + 1 < u_hi ? u + 1 : u, have_next)] * 4;
        if (tile != staged_tile || (int)dec != staged_kind) {
            // ---- stage the enc tile: 64 rows (plane, group) x 8 column blocks x 16 columns x 16 B, one DMA request per
            // half row; word position p = 64 half + lane of a row is column 16 nb + n of the tile (nb = p >> 4, n = p & 15)
            const unsigned* eb = enc + (size_t)b * IAF_CD * TE;      // (32-bit words: [2 planes][32 groups][TE][4])
            __syncthreads();                      // the previous unit's operand reads are done
#pragma unroll
            for (int k = 0; k < 64 * 2 / NW; ++k) {
                const int req = wave * (64 * 2 / NW) + k, row = req >> 1, p = (req & 1) * 64 + lane;
                int src_col = dec ? CK_DEC * (16 * (j >> 2) + (p & 15)) + 8 * (j & 3) + (p >> 4) : CK_NC * j + p;
                src_col = min(c0 + src_col, (int)TE - 1);      // (columns past the utterance: their results are dropped below)
                ck_dma16(eb + ((size_t)row * TE + src_col) * 4, lds_base + row * (CK_RS * 16) + (req & 1) * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (also the eight fragment loads above)
            __syncthreads();
            staged_tile = tile;
            staged_kind = (int)dec;
        }
        if (!have) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                a[0][mb][0] = buf_ld4(rw, lane * 16, an + (mb * 2 + 0) * 1024);
                a[0][mb][1] = buf_ld4(rw, lane * 16, an + (mb * 2 + 1) * 1024);
            }
            continue;
        }
        const __amdgpu_buffer_rsrc_t rcb = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(C + (size_t)b * c_bstride + ((size_t)rb * NCB) * 1024), 0, (int)NCB * 4096, 0x00020000);
        f4 acc[4][8];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[mb][nb] = (f4){0.f, 0.f, 0.f, 0.f};
        // B operands one (K-step, column block) ahead of the MFMAs that use them
        wn_u4 bb[2][2];
        bb[0][0] = Bt[q * CK_RS + n];
        bb[0][1] = Bt[(32 + q) * CK_RS + n];
        __builtin_amdgcn_sched_barrier(0);     // not part of the first K-step's (2 LDS reads, 12 MFMAs) groups
        auto store_nb = [&](int nb) {
            // column blocks past the end of the row fall outside the descriptor and are dropped
            const int cb = dec ? (8 * (j & 3) + nb) * NBD + (j >> 2) : (CK_NC / 16) * j + nb;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                buf_st4<WN_C_ST_AUX>(__builtin_bit_cast(wn_u4, acc[mb][nb]), rcb, lane * 16, (cb * 4 + mb) * 1024);   // (guarded store: wn_mfma_h.h)
        };
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            // next K-step's fragments (the next unit's first ones at the end)
            const int an1 = ks + 1 < 8 ? ao + (ks + 1) * 8 * 1024 : an;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                a[(ks + 1) & 1][mb][0] = buf_ld4(rw, lane * 16, an1 + (mb * 2 + 0) * 1024);
                a[(ks + 1) & 1][mb][1] = buf_ld4(rw, lane * 16, an1 + (mb * 2 + 1) * 1024);
            }
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                const int cur = nb & 1;
                if (ks * 8 + nb + 1 < 64) {
                    const int ks1 = (ks * 8 + nb + 1) >> 3, nb1 = (nb + 1) & 7;
                    bb[cur ^ 1][0] = Bt[(4 * ks1 + q) * CK_RS + 16 * nb1 + n];
                    bb[cur ^ 1][1] = Bt[(32 + 4 * ks1 + q) * CK_RS + 16 * nb1 + n];
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = mfma_h(a[ks & 1][mb][0], bb[cur][0], acc[mb][nb]);
#if !(WN_F16X2 & 1)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = mfma_h(a[ks & 1][mb][0], bb[cur][1], acc[mb][nb]);
#endif
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = mfma_h(a[ks & 1][mb][1], bb[cur][0], acc[mb][nb]);
                // results of a column block leave while the next one is being computed
                if (ks == 7 && nb >= 1) store_nb(nb - 1);
                // the eight fragment loads of the next K-step go out FIRST (left alone the scheduler sinks them
                // to the end of the K-step and the next one starts by waiting a full L2 round trip for them)
                if (nb == 0) __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, (WN_F16X2 & 1) ? 8 : 12, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        store_nb(7);
    }
}

struct CSrc {
    __amdgpu_buffer_rsrc_t rl, rc, rh;
    int vo[3];
    int vc;
};

// ---------------- residual layer with hoisted conditioning ----------------
// Same contraction as iaf_layer_h_kernel minus its eight enc K-steps: the accumulators start
// from the C tile.  Column map of a wave: 16 HN columns, column block e = columns 16e..16e+15.
// LAST: the layer is the last one of its flow -- the flow head (out1 over relu(l), relu, mean / scale
// projections, x <- x*s + m, running mean_tot / scale_tot; parallel_wavenet.py:256-277, :319-324) runs
// in the epilogue on the accumulator registers: the layer's output is neither written nor read back.
struct HeadArgs {
    const float* Ch;            // hoisted conditioning rows of the head
    const unsigned* wpack;      // head weight image
    float* x;
    float* Mt;
    float* St;
    int XR;
    int64_t T;
    int first;
    unsigned* status;           // range-guard word of the call (every variant; wn_codec.h)
};

// W2: ONE workgroup of 512 threads per CU instead of two of 256: its two halves walk tiles independently (like two
// workgroups) but share one weight image -- half the staging traffic, half the workgroups to dispatch.
// DMA: the weight image arrives by LDS-DMA, requested ahead of the first tile (short launches: few tiles per workgroup).
// NOPF (HN = 2): no register double buffer for the next tile's operands, which lets two workgroups of the 128-column
// form share a CU (each fragment read from LDS then feeds two column blocks: half the LDS traffic of HN = 1).
template <int HN, bool LAST = false, bool W2 = false, bool DMA = false, bool NOPF = false>
__global__ __launch_bounds__(W2 ? 512 : 256, ((HN == 1 && !W2) || NOPF) ? 2 : 1) void iaf_layer_c_kernel(
    const unsigned* __restrict__ lin, unsigned* __restrict__ lout, const float* __restrict__ C, int64_t c_bstride,
    const unsigned* __restrict__ wpack, int64_t RS, int d, int tiles_per_row, int ntiles, HeadArgs ha) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    constexpr int TILE = 64 * HN;
    constexpr int NT = W2 ? 512 : 256;
    float amax = 0.f;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int n = lane & 15, q = lane >> 4;
    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(ldsw) + lane;          // [((s*4+mb)*2+plane)*64]
    const wn_u4* PRl = Pl + 6 * 4 * 2 * 64;
    const float* ldsf = reinterpret_cast<const float*>(ldsw);
    const float* bg = ldsf + LC_A_WORDS + IAF_PR_FLOATS + q * 16;
    const float* br = bg + 64;
    const int RS16 = (int)RS * 16;
    const int lane_l = q * RS16 + (wave * 16 * HN + n + IAF_LP) * 16;
    const int lane_c = wave * HN * 4096 + lane * 16;

    auto tile_src = [&](int tile) -> CSrc {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * TILE;
        CSrc s;
        s.rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lin + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        s.rc = __builtin_amdgcn_make_buffer_rsrc((void*)(C + (size_t)b * c_bstride), 0, 0x7ffffff0, 0x00020000);
        s.vo[0] = lane_l + (tt - 2 * d) * 16;
        s.vo[1] = lane_l + (tt - d) * 16;
        s.vo[2] = lane_l + tt * 16;
        s.vc = lane_c + tt * 256;
        if (LAST) s.rh = __builtin_amdgcn_make_buffer_rsrc((void*)(ha.Ch + (size_t)b * c_bstride), 0, 0x7ffffff0, 0x00020000);
        return s;
    };
    // K-steps 0-5: taps t-2d, t-d, t (two 32-channel steps each)
    auto loadK = [&](const CSrc& s, int ks) -> KOp<HN> {
        KOp<HN> o;
#pragma unroll
        for (int e = 0; e < HN; ++e) {
            o.h[e] = buf_ld4(s.rl, s.vo[ks >> 1] + 256 * e, (4 * (ks & 1)) * RS16);
            o.l[e] = buf_ld4(s.rl, s.vo[ks >> 1] + 256 * e, (8 + 4 * (ks & 1)) * RS16);
        }
        return o;
    };

    // Operand double buffer: ALL loads of tile k+1 are issued before tile k is computed, so a
    // workgroup keeps one full tile (48 KB) in flight for the whole tile period.
    auto load_tile = [&](int tile, KOp<HN> (&bc)[6], f4 (&cp)[4][HN], f4 (&ch)[4][HN]) {
        const CSrc s = tile_src(tile);
#pragma unroll
        for (int e = 0; e < HN; ++e)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) cp[mb][e] = buf_ldf4(s.rc, s.vc + (e * 4 + mb) * 1024, 0);
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) bc[ks] = loadK(s, ks);
        if (LAST) {
#pragma unroll
            for (int e = 0; e < HN; ++e)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) ch[mb][e] = buf_ldf4(s.rh, s.vc + (e * 4 + mb) * 1024, 0);
        }
    };
    const TileWalk tw = W2 ? tile_walk_parts(ntiles, 2, (int)(threadIdx.x >> 8)) : tile_walk(ntiles);
    const int tstep = tw.step, tend = tw.end;
    float inv_m = 0.f, inv_r = 0.f;

    // head weights / constants (LAST): second image behind the layer's
    const wn_u4* PHl = reinterpret_cast<const wn_u4*>(ldsw + LC_LDS_WORDS) + lane;
    const float* bo = ldsf + LC_LDS_WORDS + HC_A_WORDS + q * 16;
    const float* wm = bo + 64;
    const float* wsc = wm + 64;
    float bmean = 0.f, bscale = 0.f, inv_h = 0.f;

    auto body = [&](int tile, KOp<HN> (&bc)[6], f4 (&cp)[4][HN], f4 (&ch)[4][HN], KOp<HN> (&bn)[6], f4 (&cn)[4][HN],
                    f4 (&chn)[4][HN]) {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * TILE;
        if (!NOPF && tile + tstep < tend) load_tile(tile + tstep, bn, cn, chn);
        f4 acc[4][HN];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < HN; ++e) acc[mb][e] = cp[mb][e];
        wn_u4 a[2][4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            a[0][mb][0] = Pl[((0 * 4 + mb) * 2 + 0) * 64];
            a[0][mb][1] = Pl[((0 * 4 + mb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            if (ks + 1 < 6) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    a[(ks + 1) & 1][mb][0] = Pl[(((ks + 1) * 4 + mb) * 2 + 0) * 64];
                    a[(ks + 1) & 1][mb][1] = Pl[(((ks + 1) * 4 + mb) * 2 + 1) * 64];
                }
            }
#pragma unroll
            for (int e = 0; e < HN; ++e)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    acc[mb][e] = mfma3(a[ks & 1][mb][0], a[ks & 1][mb][1], bc[ks].h[e], bc[ks].l[e], acc[mb][e]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue per column block: gate, residual 1x1, split, store
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
                    // l_old of channels 16mb+4q+2rp(+1): tap-t K-step 4+(mb>>1), slot 2(mb&1)+rp
                    float l0, l1;
                    wn_join_pair(bc[4 + (mb >> 1)].h[e][(mb & 1) * 2 + rp], bc[4 + (mb >> 1)].l[e][(mb & 1) * 2 + rp], l0, l1);
                    const float v0 = l0 + fmaf(rc[2 * rp], inv_r, br[mb * 4 + 2 * rp]);
                    const float v1 = l1 + fmaf(rc[2 * rp + 1], inv_r, br[mb * 4 + 2 * rp + 1]);
                    unsigned hw, lw;
                    wn_split_pair_t(v0, v1, hw, lw, amax);
                    oh[mb >> 1][(mb & 1) * 2 + rp] = hw;
                    ol[mb >> 1][(mb & 1) * 2 + rp] = lw;
                }
            }
            if (!LAST) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    buf_st4<WN_L_ST_AUX>(oh[s2], ro, vo_out + 256 * e, (4 * s2) * RS16);
                    buf_st4<WN_L_ST_AUX>(ol[s2], ro, vo_out + 256 * e, (8 + 4 * s2) * RS16);
                }
            } else {
                // ---- flow head on this column block (same arithmetic as iaf_head_c_kernel) ----
                f4 hacc[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) hacc[mb] = ch[mb][e];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    wn_u4 bh = oh[ks], bl = ol[ks];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {             // relu(l) (:256) on the reconstructed value
                        float v0, v1;
                        wn_join_pair(bh[i], bl[i], v0, v1);
                        unsigned hw, lw;
                        wn_split_pair(fmaxf(v0, 0.f), fmaxf(v1, 0.f), hw, lw);
                        bh[i] = hw;
                        bl[i] = lw;
                    }
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        hacc[mb] = mfma3(PHl[((ks * 4 + mb) * 2 + 0) * 64], PHl[((ks * 4 + mb) * 2 + 1) * 64], bh, bl, hacc[mb]);
                }
                float pm = 0.f, ps = 0.f;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float o = fmaxf(fmaf(hacc[mb][r], inv_h, bo[mb * 4 + r]), 0.f);
                        pm = fmaf(wm[mb * 4 + r], o, pm);
                        ps = fmaf(wsc[mb * 4 + r], o, ps);
                    }
                pm += __shfl_xor(pm, 16);
                ps += __shfl_xor(ps, 16);
                pm += __shfl_xor(pm, 32);
                ps += __shfl_xor(ps, 32);
                if (q == 0) {
                    const int64_t t = tt + wave * 16 * HN + 16 * e + n;
                    const float mean = pm + bmean;
                    const float sc = fminf(fmaxf(softplus_tf(ps + bscale), EXP_M9), EXP_7);   // :105-114
                    float* xp = ha.x + (size_t)b * ha.XR + IAF_XP + t;
                    *xp = *xp * sc + mean;                                                    // :277
                    float* mp = ha.Mt + (size_t)b * ha.T + t;
                    float* sp = ha.St + (size_t)b * ha.T + t;
                    if (ha.first) { *mp = mean; *sp = sc; }
                    else { *mp = mean + *mp * sc; *sp = *sp * sc; }                           // :322-323
                }
            }
        }
    };

    KOp<HN> bA[6], bB[6];
    f4 cA[4][HN], cB[4][HN], hA[4][HN], hB[4][HN];
    int tile = tw.first;
    if (DMA && !LAST && !W2) {
        // The weight image goes to LDS by LDS-DMA and is requested BEFORE the first tile's operands: loads return in
        // order, so behind the tile's loads the image would arrive a full fabric round trip late (3.3 us of a 13 us
        // workgroup life); in front of them it comes out of L2 while the tile is still on its way.
        auto dma = [&](const unsigned* src, unsigned* dst, int nchunks) {
            for (int i = wave; i < nchunks; i += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(i * 64 + lane) * 4),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
        };
        constexpr int TAIL_CH = LC_TAIL_WORDS / 256, TAIL_REM = LC_TAIL_WORDS / 4 - TAIL_CH * 64;   // 8 chunks + 33 lanes
        dma(wpack, ldsw, LC_A_WORDS / 256);
        dma(wpack + IAF_P_FLOATS, ldsw + LC_A_WORDS, TAIL_CH);
        if (wave == 0 && lane < TAIL_REM)       // the last 33 16-byte words of the image
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(wpack + IAF_P_FLOATS + TAIL_CH * 256 + lane * 4),
                (__attribute__((address_space(3))) void*)(ldsw + LC_A_WORDS + TAIL_CH * 256), 16, 0, 0);
        const bool has_tile = tile < tend;
        __builtin_amdgcn_sched_barrier(0);      // the vmcnt below counts on the image's loads being OLDER than the tile's
        if (has_tile) load_tile(tile, bA, cA, hA);
        // the 16 HN operand loads of the tile may stay in flight; everything older (the image) has landed.  A plain
        // s_barrier: __syncthreads() carries a fence that makes the compiler wait for vmcnt(0) here.
        if (has_tile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * HN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
    if (tile < tend) load_tile(tile, bA, cA, hA);
    // the weight image is staged AFTER the first tile's operand loads are in flight
    stage_words<LC_A_WORDS, NT>(wpack, ldsw);
    stage_words<LC_TAIL_WORDS, NT>(wpack + IAF_P_FLOATS, ldsw + LC_A_WORDS);
    }
    if (LAST) {
        stage_words<HC_A_WORDS, NT>(ha.wpack, ldsw + LC_LDS_WORDS);
        stage_words<HC_TAIL_WORDS, NT>(ha.wpack + IAF_PH_FLOATS, ldsw + LC_LDS_WORDS + HC_A_WORDS);
        bmean = ldsf[LC_LDS_WORDS + HC_A_WORDS + 192];
        bscale = ldsf[LC_LDS_WORDS + HC_A_WORDS + 193];
        inv_h = ldsf[LC_LDS_WORDS + HC_A_WORDS + 194];
    }
    inv_m = ldsf[LC_A_WORDS + IAF_PR_FLOATS + 128];
    inv_r = ldsf[LC_A_WORDS + IAF_PR_FLOATS + 129];
    if (NOPF) {
        while (tile < tend) {
            body(tile, bA, cA, hA, bA, cA, hA);
            tile += tstep;
            if (tile < tend) load_tile(tile, bA, cA, hA);
        }
    } else
    while (tile < tend) {
        body(tile, bA, cA, hA, bB, cB, hB);
        tile += tstep;
        if (tile >= tend) break;
        body(tile, bB, cB, hB, bA, cA, hA);
        tile += tstep;
    }
    wn_range_flag(amax, ha.status);
}

// ---------------- two residual layers in one launch (small dilations) ----------------
// Layers (A, B) with dilations (d, 2d), 4d <= 16.  A wave walks a contiguous run of 16-column
// blocks: layer A's block output (the four 16-byte operand words a lane holds after the epilogue)
// IS layer B's tap-t operand in the same lane, and B's taps t-2d, t-4d are column shifts of at most
// one block -- a DPP row shift of the current block's words merged with a row rotate of the previous
// block's, all in registers.  Layer A's output never goes to memory: 256 (l) + 2 x 256 (C) + 256
// (l out) = 1024 B/sample for two layers instead of 2 x 768, and one launch floor instead of two.
// A run starts one block early (layer A only) to have a previous block; left of the utterance the
// previous block is zero like the padded l rows.  Eight waves per workgroup (two per SIMD): one
// wave's gate/split VALU work overlaps the other's MFMAs; operands are reloaded for the next block
// into the registers that were just consumed.
// FIRST: layer A is the first layer of a flow and its operands are computed from the flow input x
// (start conv fused in, see first_layer_operands).
constexpr int PC_THREADS = 512;
constexpr int PC_LDS_WORDS = 2 * LC_LDS_WORDS + IAF_START_LDS_WORDS;

// column shift by SH (1..16) of a block's words: lane n gets cur[n - SH], or prev[n - SH + 16]
template <int SH>
__device__ inline wn_u4 pair_shift(wn_u4 prev, wn_u4 cur) {
    if (SH == 16) return prev;
    wn_u4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rot = __builtin_amdgcn_update_dpp(0, (int)prev[i], 0x120 + (SH & 15), 0xf, 0xf, true);   // row_ror:SH
        o[i] = (unsigned)__builtin_amdgcn_update_dpp(rot, (int)cur[i], 0x110 + (SH & 15), 0xf, 0xf, false);  // row_shr:SH
    }
    return o;
}

template <int DB, bool FIRST>
__global__ __launch_bounds__(PC_THREADS, 1) void iaf_pair_c_kernel(
    const unsigned* __restrict__ lin, unsigned* __restrict__ lout, const float* __restrict__ CA,
    const float* __restrict__ CB, int64_t c_bstride, const unsigned* __restrict__ wA, const unsigned* __restrict__ wB,
    int64_t RS, int NB, int rl, int rpr, int ntasks, const float* __restrict__ x, int XR,
    const float* __restrict__ wstart, unsigned* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    constexpr int DA = DB / 2, NW = PC_THREADS / 64;
    float amax = 0.f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const float* ldsf = reinterpret_cast<const float*>(ldsw);
    PairLayer LA, LB;
    LA.Pl = reinterpret_cast<const wn_u4*>(ldsw) + lane;
    LA.PRl = LA.Pl + 6 * 4 * 2 * 64;
    LA.bg = ldsf + LC_A_WORDS + IAF_PR_FLOATS + q * 16;
    LA.br = LA.bg + 64;
    LB.Pl = reinterpret_cast<const wn_u4*>(ldsw + LC_LDS_WORDS) + lane;
    LB.PRl = LB.Pl + 6 * 4 * 2 * 64;
    LB.bg = ldsf + LC_LDS_WORDS + LC_A_WORDS + IAF_PR_FLOATS + q * 16;
    LB.br = LB.bg + 64;
    const f4* wq = reinterpret_cast<const f4*>(ldsw + 2 * LC_LDS_WORDS);
    const int RS16 = (int)RS * 16;
    const int lane_l = q * RS16 + (n + IAF_LP) * 16;

    // task walk: each XCD takes one contiguous eighth of the runs
    int first, end, step;
    if ((gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, per = (ntasks + 7) >> 3;
        first = xcd * per + (int)(blockIdx.x >> 3) * NW + wave;
        end = min(ntasks, (xcd + 1) * per);
        step = (int)(gridDim.x >> 3) * NW;
    } else {
        first = (int)blockIdx.x * NW + wave;
        end = ntasks;
        step = (int)gridDim.x * NW;
    }

    // operands of the block being computed; every register is reloaded for the next block right after its
    // last use.  The loads are unconditional: "no next block" is an offset past the end of the descriptor
    // (returns zeros, moves nothing) -- a branch around a load makes the compiler drain vmcnt at the join.
    constexpr int OOB = 0x40000000;
    KOp<1> bc[6];
    f4 caA[4], cbA[4], caB[4], cbB[4];          // C tiles ping-pong (the accumulators take over their registers)
    float xvA[5], xvB[5];
    int b = 0, s = 0, e = 0;
    __amdgpu_buffer_rsrc_t rl_, rca, rcb, ro, rx;
    auto set_run = [&](int task) {
        b = task / rpr;
        s = (task - b * rpr) * rl;
        e = min(NB, s + rl);
        rl_ = __builtin_amdgcn_make_buffer_rsrc((void*)(lin + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        ro = __builtin_amdgcn_make_buffer_rsrc((void*)(lout + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        rca = __builtin_amdgcn_make_buffer_rsrc((void*)(CA + (size_t)b * c_bstride), 0, NB * 4096, 0x00020000);
        rcb = __builtin_amdgcn_make_buffer_rsrc((void*)(CB + (size_t)b * c_bstride), 0, NB * 4096, 0x00020000);
        if (FIRST) rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)b * XR), 0, XR * 4, 0x00020000);
    };
    auto load_c = [&](const __amdgpu_buffer_rsrc_t& r, f4 (&c)[4], int k, int pred) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) c[mb] = buf_ldf4(r, lane * 16 + k * 4096 + pred, mb * 1024);
    };
    auto load_bc = [&](int k, int ks, int pred) {
        const int vo = lane_l + (16 * k - (2 - (ks >> 1)) * DA) * 16 + pred;
        bc[ks].h[0] = buf_ld4(rl_, vo, (4 * (ks & 1)) * RS16);
        bc[ks].l[0] = buf_ld4(rl_, vo, (8 + 4 * (ks & 1)) * RS16);
    };
    auto load_x = [&](float (&xv)[5], int k, int pred) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
            xv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  rx, (IAF_XP + 16 * k + n + j - 5) * 4 + pred, 0, 0));
    };
    auto load_all = [&](int k) {
        load_c(rca, caA, k, 0);
        load_c(rcb, cbA, k, k >= s ? 0 : OOB);
        if (FIRST) {
            load_x(xvA, k, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) load_bc(k, ks, 0);
        }
    };
    // K loop of one layer: acc starts from c; B operands through op(ks); after(ks) runs once the K-step is issued
    auto contract = [&](const PairLayer& w, f4 (&acc)[4], auto&& op, auto&& after) {
        wn_u4 a[2][4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            a[0][mb][0] = w.Pl[((0 * 4 + mb) * 2 + 0) * 64];
            a[0][mb][1] = w.Pl[((0 * 4 + mb) * 2 + 1) * 64];
        }
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            if (ks + 1 < 6) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    a[(ks + 1) & 1][mb][0] = w.Pl[(((ks + 1) * 4 + mb) * 2 + 0) * 64];
                    a[(ks + 1) & 1][mb][1] = w.Pl[(((ks + 1) * 4 + mb) * 2 + 1) * 64];
                }
            }
            wn_u4 bh, bl;
            op(ks, bh, bl);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb] = mfma3(a[ks & 1][mb][0], a[ks & 1][mb][1], bh, bl, acc[mb]);
            after(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (first < end) {
        set_run(first);
        load_all(s > 0 ? s - 1 : 0);
    }
    // both weight images are staged AFTER the first block's operand loads are in flight
    if (FIRST) stage_start_weights(wstart, reinterpret_cast<f4*>(ldsw + 2 * LC_LDS_WORDS));
    {
        // 16-byte words of one layer image: fragments and tail are contiguous in LDS; in the blob the tail
        // sits at IAF_P_FLOATS.  All loads of a thread are issued before its first LDS store.
        constexpr int NV = LC_LDS_WORDS / 4, NCH = (NV + PC_THREADS - 1) / PC_THREADS;
        static_assert(LC_LDS_WORDS % 4 == 0, "");
        const wn_u4* sa = reinterpret_cast<const wn_u4*>(wA);
        const wn_u4* sb = reinterpret_cast<const wn_u4*>(wB);
        wn_u4* dst = reinterpret_cast<wn_u4*>(ldsw);
        wn_u4 ta[NCH], tb[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = c * PC_THREADS + (int)threadIdx.x;
            const int src = i < LC_A_WORDS / 4 ? i : i - LC_A_WORDS / 4 + IAF_P_FLOATS / 4;
            if (i < NV) { ta[c] = sa[src]; tb[c] = sb[src]; }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int i = c * PC_THREADS + (int)threadIdx.x;
            if (i < NV) { dst[i] = ta[c]; dst[NV + i] = tb[c]; }
        }
        __syncthreads();
    }
    LA.inv_m = ldsf[LC_A_WORDS + IAF_PR_FLOATS + 128];
    LA.inv_r = ldsf[LC_A_WORDS + IAF_PR_FLOATS + 129];
    LB.inv_m = ldsf[LC_LDS_WORDS + LC_A_WORDS + IAF_PR_FLOATS + 128];
    LB.inv_r = ldsf[LC_LDS_WORDS + LC_A_WORDS + IAF_PR_FLOATS + 129];

    for (int task = first; task < end; task += step) {
        if (task != first) {
            set_run(task);
            load_all(s > 0 ? s - 1 : 0);
        }
        wn_u4 ph[2], pl[2];                     // layer A's output of the previous block
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) ph[s2] = pl[s2] = (wn_u4){0u, 0u, 0u, 0u};
        auto block = [&](int k, f4 (&ca)[4], f4 (&cb)[4], float (&xv)[5], f4 (&can)[4], f4 (&cbn)[4], float (&xvn)[5]) {
            const int pred = k + 1 < e ? 0 : OOB;
            // ---- layer A ----
            if (FIRST) {
                first_layer_operands(xv, (long long)16 * k + n, q, wq, bc, amax);
                load_x(xvn, k + 1, pred);
            }
            f4 acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb] = ca[mb];
            load_c(rca, can, k + 1, pred);
            wn_u4 th[2], tl[2];                 // tap-t words: the residual's skip input
            contract(LA, acc,
                     [&](int ks, wn_u4& bh, wn_u4& bl) {
                         bh = bc[ks].h[0];
                         bl = bc[ks].l[0];
                         if (ks >= 4) { th[ks - 4] = bh; tl[ks - 4] = bl; }
                     },
                     [&](int ks) { if (!FIRST) load_bc(k + 1, ks, pred); });
            wn_u4 oh[2], ol[2];
            pair_epilogue(LA, acc, th, tl, oh, ol, amax);
            // ---- layer B ----
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb] = cb[mb];
            load_c(rcb, cbn, k + 1, pred);
            if (k >= s) {
                contract(LB, acc,
                         [&](int ks, wn_u4& bh, wn_u4& bl) {
                             const int s2 = ks & 1;
                             if (ks < 2) { bh = pair_shift<2 * DB>(ph[s2], oh[s2]); bl = pair_shift<2 * DB>(pl[s2], ol[s2]); }
                             else if (ks < 4) { bh = pair_shift<DB>(ph[s2], oh[s2]); bl = pair_shift<DB>(pl[s2], ol[s2]); }
                             else { bh = oh[s2]; bl = ol[s2]; }
                         },
                         [&](int) {});
                wn_u4 qh[2], ql[2];
                pair_epilogue(LB, acc, oh, ol, qh, ql, amax);
                const int vo_out = lane_l + 16 * k * 16;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    buf_st4<WN_L_ST_AUX>(qh[s2], ro, vo_out, (4 * s2) * RS16);
                    buf_st4<WN_L_ST_AUX>(ql[s2], ro, vo_out, (8 + 4 * s2) * RS16);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { ph[s2] = oh[s2]; pl[s2] = ol[s2]; }
        };
        int k = s > 0 ? s - 1 : 0;
        while (k < e) {
            block(k, caA, cbA, xvA, caB, cbB, xvB);
            if (++k >= e) break;
            block(k, caB, cbB, xvB, caA, cbA, xvA);
            ++k;
        }
    }
    wn_range_flag(amax, status);
}

// ---------------- flow head with hoisted conditioning (parallel_wavenet.py:256-277, :319-324) ----------------
template <int HN>
__global__ __launch_bounds__(256, 2) void iaf_head_c_kernel(
    const unsigned* __restrict__ lin, const float* __restrict__ C, int64_t c_bstride, const unsigned* __restrict__ wpack,
    float* __restrict__ x, float* __restrict__ Mt, float* __restrict__ St,
    int64_t RS, int XR, int64_t T, int first, int tiles_per_row, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned ldsw[];
    constexpr int TILE = 64 * HN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const wn_u4* Pl = reinterpret_cast<const wn_u4*>(ldsw) + lane;
    const float* ldsf = reinterpret_cast<const float*>(ldsw);
    const float* bo = ldsf + HC_A_WORDS + q * 16;
    const float* wm = bo + 64;
    const float* wsc = wm + 64;
    const int RS16 = (int)RS * 16;
    const int lane_l = q * RS16 + (wave * 16 * HN + n + IAF_LP) * 16;
    const int lane_c = wave * HN * 4096 + lane * 16;

    auto tile_src = [&](int tile) -> CSrc {
        const int b = tile / tiles_per_row;
        const int tt = (tile - b * tiles_per_row) * TILE;
        CSrc s;
        s.rl = __builtin_amdgcn_make_buffer_rsrc((void*)(lin + (size_t)b * IAF_W * RS), 0, IAF_W * (int)RS * 4, 0x00020000);
        s.rc = __builtin_amdgcn_make_buffer_rsrc((void*)(C + (size_t)b * c_bstride), 0, 0x7ffffff0, 0x00020000);
        s.vo[0] = s.vo[1] = s.vo[2] = lane_l + tt * 16;
        s.vc = lane_c + tt * 256;
        return s;
    };
    // K-steps 0-1: out1 over relu(l)
    auto loadK = [&](const CSrc& s, int ks) -> KOp<HN> {
        KOp<HN> o;
#pragma unroll
        for (int e = 0; e < HN; ++e) {
            o.h[e] = buf_ld4(s.rl, s.vo[2] + 256 * e, (4 * ks) * RS16);
            o.l[e] = buf_ld4(s.rl, s.vo[2] + 256 * e, (8 + 4 * ks) * RS16);
        }
        return o;
    };
    KOp<HN> bc[2];
    f4 cpre[4][HN];
    const TileWalk tw = tile_walk(ntiles);
    const int tile0 = tw.first, tstep = tw.step, tend = tw.end;
    if (tile0 < tend) {
        const CSrc s0 = tile_src(tile0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bc[ks] = loadK(s0, ks);
#pragma unroll
        for (int e = 0; e < HN; ++e)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) cpre[mb][e] = buf_ldf4(s0.rc, s0.vc + (e * 4 + mb) * 1024, 0);
    }
    stage_words<HC_A_WORDS>(wpack, ldsw);
    stage_words<HC_TAIL_WORDS>(wpack + IAF_PH_FLOATS, ldsw + HC_A_WORDS);
    const float bmean = ldsf[HC_A_WORDS + 192], bscale = ldsf[HC_A_WORDS + 193];
    const float inv_m = ldsf[HC_A_WORDS + 194];
    for (int tile = tile0; tile < tend; tile += tstep) {
        const int b = tile / tiles_per_row;
        const int t0 = (tile - b * tiles_per_row) * TILE + wave * 16 * HN;
        const int next = tile + tstep;
        const bool has_next = next < tend;
        const CSrc sn = tile_src(has_next ? next : tile);
        f4 acc[4][HN];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int e = 0; e < HN; ++e) acc[mb][e] = cpre[mb][e];
        if (has_next) {
#pragma unroll
            for (int e = 0; e < HN; ++e)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) cpre[mb][e] = buf_ldf4(sn.rc, sn.vc + (e * 4 + mb) * 1024, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wn_u4 a[4][2];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                a[mb][0] = Pl[((ks * 4 + mb) * 2 + 0) * 64];
                a[mb][1] = Pl[((ks * 4 + mb) * 2 + 1) * 64];
            }
#pragma unroll
            for (int e = 0; e < HN; ++e) {
                wn_u4 bh = bc[ks].h[e], bl = bc[ks].l[e];
#pragma unroll
                for (int i = 0; i < 4; ++i) {             // relu(l) (:256) on the reconstructed value
                    float v0, v1;
                    wn_join_pair(bh[i], bl[i], v0, v1);
                    unsigned hw, lw;
                    wn_split_pair(fmaxf(v0, 0.f), fmaxf(v1, 0.f), hw, lw);
                    bh[i] = hw;
                    bl[i] = lw;
                }
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb][e] = mfma3(a[mb][0], a[mb][1], bh, bl, acc[mb][e]);
            }
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
                const int64_t t = t0 + 16 * e + n;
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

int pick_hn_c(int B, int64_t T, int slots) {
    const char* force = getenv("WN_HN");
    if (force) return atoi(force) == 1 ? 1 : 2;
    if (T % 128) return 1;
    const int64_t n1 = B * (T / 64), n2 = B * (T / 128);
    const int64_t c1 = (n1 + slots - 1) / slots, c2 = 2 * ((n2 + slots - 1) / slots);
    return c1 <= c2 ? 1 : 2;
}

}  // namespace

int wn_iaf_c_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_cond_h_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, CK_LDS_BYTES));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_c_kernel<1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (LC_LDS_WORDS + HC_LDS_WORDS) * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_layer_c_kernel<1, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (LC_LDS_WORDS + HC_LDS_WORDS) * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_pair_c_kernel<2, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_WORDS * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_pair_c_kernel<2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_WORDS * 4));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(iaf_pair_c_kernel<8, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS_WORDS * 4));
    return WN_OK;
}

// floats of C per batch row for R row blocks over T samples
size_t wn_iaf_c_floats(int R, int64_t T) { return (size_t)R * (size_t)(T / 16) * 1024; }

// Conditioning GEMM of the R row blocks listed in rb_off (word offsets of their 8-K-step fragment arrays inside the
// weight blob) over enc columns [c0, c0+T).  order: the row-block indices, the n_nat natural ones first, then the
// decimated ones (n_nat == R: every row block in natural column order).
void wn_iaf_c_cond(const float* enc, const float* wblob, const unsigned* rb_off, const unsigned* order, int n_nat,
                   float* C, int64_t c_bstride, int64_t TE, int c0, int R, int B, int64_t T, int num_cu, hipStream_t st) {
    const int tiles_per_row = (int)((T + CK_NC - 1) / CK_NC), ntiles = B * tiles_per_row;
    constexpr int NW = CK_THREADS / 64;
    const int n_dec = R - n_nat;
    const int sr_nat = (n_nat + NW - 1) / NW, sr_dec = (n_dec + NW - 1) / NW;   // sub-rounds of eight row blocks per tile
    const int64_t units = (int64_t)ntiles * (sr_nat + sr_dec);
    int grid = (int)std::min<int64_t>(units, num_cu);
    if (grid >= 8) grid = grid / 8 * 8;                                          // XCD-aware ranges: a multiple of 8
    hipLaunchKernelGGL(iaf_cond_h_kernel, dim3(grid), dim3(CK_THREADS), CK_LDS_BYTES, st,
                       reinterpret_cast<const unsigned*>(enc), reinterpret_cast<const unsigned*>(wblob), rb_off, order,
                       n_nat, C, c_bstride, TE, c0, R, sr_nat, sr_dec, tiles_per_row, ntiles, T / 16);
}

// workgroups per CU of the hoisted layer / head kernels (57 KB of LDS; the 128-column variant
// needs the whole register file)
static int lc_slots(int hn) { return hn == 1 ? 2 : 1; }
// 64-column tiles: one 512-thread workgroup per CU (two halves, one weight image) instead of two of 256
static bool lc_w2() {
    static const bool on = getenv("WN_LC_W2") && atoi(getenv("WN_LC_W2")) != 0;   // measured slower (46.3 vs 49.2 M samples/s at one utterance): off
    return on;
}

void wn_iaf_c_layer(const float* lin, float* lout, const float* C, int64_t c_bstride, const float* wpack, int64_t RS,
                    int d, int B, int64_t T, int num_cu, hipStream_t st, unsigned* status) {
    HeadArgs hs{};
    hs.status = status;
    const int hn = pick_hn_c(B, T, 2 * num_cu);
    const int tiles_per_row = (int)(T / (64 * hn)), ntiles = B * tiles_per_row;
    const int slots = lc_slots(hn) * num_cu;
    const int grid = ntiles < slots ? ntiles : slots;
    if (hn == 1 && lc_w2()) {
        const int g2 = std::min(num_cu, (ntiles + 1) / 2);
        hipLaunchKernelGGL((iaf_layer_c_kernel<1, false, true>), dim3(g2), dim3(512), LC_LDS_WORDS * 4, st,
                           reinterpret_cast<const unsigned*>(lin), reinterpret_cast<unsigned*>(lout), C, c_bstride,
                           reinterpret_cast<const unsigned*>(wpack), RS, d, tiles_per_row, ntiles, hs);
        return;
    }
    // many tiles per workgroup (several utterances): 128-column tiles WITHOUT the register double buffer, two workgroups
    // per CU -- every fragment read from LDS feeds two column blocks (0.61 against 0.593 of the HBM peak at eight
    // utterances); with few tiles it loses (0.35 against 0.435 at one utterance: the second tile's loads are exposed)
    static const int nopf_env = getenv("WN_LC_NOPF") ? atoi(getenv("WN_LC_NOPF")) : -1;
    const bool nopf = nopf_env >= 0 ? nopf_env != 0 : (int64_t)B * (T / 128) >= 8 * (int64_t)num_cu;
    if (nopf && T % 128 == 0 && !getenv("WN_HN")) {
        const int tpr = (int)(T / 128), nt2 = B * tpr, g2 = std::min(nt2, 2 * num_cu);
        hipLaunchKernelGGL((iaf_layer_c_kernel<2, false, false, true, true>), dim3(g2), dim3(256), LC_LDS_WORDS * 4, st,
                           reinterpret_cast<const unsigned*>(lin), reinterpret_cast<unsigned*>(lout), C, c_bstride,
                           reinterpret_cast<const unsigned*>(wpack), RS, d, tpr, nt2, hs);
        return;
    }
    // few tiles per workgroup (one utterance): the start-up is a third of the workgroup's life and the DMA-staged
    // image takes 5 % off the launch; with many tiles (eight utterances) the register-staged form is 2 % faster
    static const int dma_env = getenv("WN_LC_DMA") ? atoi(getenv("WN_LC_DMA")) : -1;
    const bool dma = dma_env >= 0 ? dma_env != 0 : ntiles <= 4 * grid;
    auto kern = hn == 1 ? (dma ? iaf_layer_c_kernel<1, false, false, true> : iaf_layer_c_kernel<1>)
                        : (dma ? iaf_layer_c_kernel<2, false, false, true> : iaf_layer_c_kernel<2>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LC_LDS_WORDS * 4, st, reinterpret_cast<const unsigned*>(lin),
                       reinterpret_cast<unsigned*>(lout), C, c_bstride, reinterpret_cast<const unsigned*>(wpack), RS, d,
                       tiles_per_row, ntiles, hs);
}

// Last layer of a flow with the flow head in its epilogue (64-sample tiles, two workgroups per CU).
bool wn_iaf_c_last_ok() { return !getenv("WN_NO_HEADFUSE"); }

void wn_iaf_c_layer_head(const float* lin, const float* C, const float* Ch, int64_t c_bstride, const float* wpack,
                         const float* wpack_head, float* x, float* Mt, float* St, int64_t RS, int XR, int d, int first,
                         int B, int64_t T, int num_cu, hipStream_t st, unsigned* status) {
    const int tiles_per_row = (int)(T / 64), ntiles = B * tiles_per_row;
    const int grid = ntiles < 2 * num_cu ? ntiles : 2 * num_cu;
    HeadArgs ha{Ch, reinterpret_cast<const unsigned*>(wpack_head), x, Mt, St, XR, T, first, status};
    if (lc_w2()) {
        hipLaunchKernelGGL((iaf_layer_c_kernel<1, true, true>), dim3(std::min(num_cu, (ntiles + 1) / 2)), dim3(512),
                           (LC_LDS_WORDS + HC_LDS_WORDS) * 4, st, reinterpret_cast<const unsigned*>(lin),
                           static_cast<unsigned*>(nullptr), C, c_bstride, reinterpret_cast<const unsigned*>(wpack), RS, d,
                           tiles_per_row, ntiles, ha);
        return;
    }
    hipLaunchKernelGGL((iaf_layer_c_kernel<1, true>), dim3(grid), dim3(256), (LC_LDS_WORDS + HC_LDS_WORDS) * 4, st,
                       reinterpret_cast<const unsigned*>(lin), static_cast<unsigned*>(nullptr), C, c_bstride,
                       reinterpret_cast<const unsigned*>(wpack), RS, d, tiles_per_row, ntiles, ha);
}

// Two layers (dilations da, db = 2 da, 4 da <= 16) in one launch; x != nullptr: layer A is the first
// layer of a flow (da = 1) and the start conv runs inside.
bool wn_iaf_c_pair_ok(int da, int db) { return db == 2 * da && (db == 2 || db == 8) && !getenv("WN_NO_PAIR"); }

void wn_iaf_c_pair(const float* lin, float* lout, const float* CA, const float* CB, int64_t c_bstride, const float* wA,
                   const float* wB, int64_t RS, int da, int db, int B, int64_t T, int num_cu, hipStream_t st,
                   const float* x, int XR, const float* wstart, unsigned* status) {
    const int NB = (int)(T / 16);
    const int64_t nw = (int64_t)num_cu * (PC_THREADS / 64);
    const int64_t want = std::max<int64_t>(1, nw / B);       // runs per row that give every wave one run
    const char* mr = getenv("WN_PAIR_MINRUN");
    int rl = (int)((NB + want - 1) / want);
    rl = std::max(rl, mr ? atoi(mr) : 3);                    // bounds the one-block warm-up of a run to a third
    const int rpr = (NB + rl - 1) / rl;
    const int64_t ntasks = (int64_t)B * rpr;
    int grid = (int)std::min<int64_t>(num_cu, (ntasks + PC_THREADS / 64 - 1) / (PC_THREADS / 64));
    if (grid >= 8) grid = std::min(num_cu, (grid + 7) / 8 * 8);
    auto kern = db == 2 ? (x ? iaf_pair_c_kernel<2, true> : iaf_pair_c_kernel<2, false>) : iaf_pair_c_kernel<8, false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PC_THREADS), PC_LDS_WORDS * 4, st, reinterpret_cast<const unsigned*>(lin),
                       reinterpret_cast<unsigned*>(lout), CA, CB, c_bstride, reinterpret_cast<const unsigned*>(wA),
                       reinterpret_cast<const unsigned*>(wB), RS, NB, rl, rpr, (int)ntasks, x, XR, wstart, status);
}

void wn_iaf_c_head(const float* lin, const float* C, int64_t c_bstride, const float* wpack, float* x, float* Mt,
                   float* St, int64_t RS, int XR, int64_t T, int first, int B, int num_cu, hipStream_t st) {
    const int hn = pick_hn_c(B, T, 2 * num_cu);
    const int tiles_per_row = (int)(T / (64 * hn)), ntiles = B * tiles_per_row;
    const int grid = ntiles < 2 * num_cu ? ntiles : 2 * num_cu;
    auto kern = hn == 1 ? iaf_head_c_kernel<1> : iaf_head_c_kernel<2>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), HC_LDS_WORDS * 4, st, reinterpret_cast<const unsigned*>(lin), C,
                       c_bstride, reinterpret_cast<const unsigned*>(wpack), x, Mt, St, RS, XR, T, first, tiles_per_row,
                       ntiles);
}
