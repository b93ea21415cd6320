// fp32-MFMA GEMMs of the reference-precision (fp32) student path, round 6: C = A . B with the ACTIVATIONS (B, 256 input
// channels x a tile of time) resident in LDS and the WEIGHTS (A) streamed from L2 in MFMA-fragment order.
//
//   cond   -- the hoisted conditioning term of the fp32 form: C[rb] = b[rb] + Wcond[rb] (64 x 256) . enc (256 x T) for the 64
//             row blocks (60 residual layers + 4 flow heads; parallel_wavenet.py:237-244, :258-262) of a deconv stack in ONE
//             launch, written in the accumulator layout the layer / head kernels start from (wn_iaf.hip, HOIST).  The
//             reference evaluates these projections in bulk itself on the AR path (Fastgen.cond_vars, wavenet.py:353-377).
//             161 GFLOP per 4.8 s utterance -- 37 % of the student's MACs -- leave sixty 49 us launches that ran at 0.61 of the
//             fp32-MFMA peak for a GEMM whose inner loop is one LDS operand word per 16 MFMAs.
//   deconv -- the last (256 -> 256, K = 80, stride 20) transposed-conv layer of the upsampler (masked.py:235-291,
//             wavenet.py:46-73): per output phase p a dense GEMM over (tap j, input channel) on the FRAME axis,
//             y[S f + p] = sum_j W[S j + r] . x[f + d - j], r = (p + pL) mod S, d = (p + pL) div S -- 80 row blocks
//             (20 phases x 4 channel groups) against the same x tile, taps as column shifts of the LDS tile; output
//             phase-major, woven into time order by deconv_interleave_kernel (bias + activation there).
//
// One kernel template.  A TASK is one row block (64 output rows) x one column tile (NB blocks of 16 columns) = one wave's
// work: acc[4 row blocks of 16][NB] in registers, per K-group (16 input channels = four K-steps of v_mfma_f32_16x16x4_f32)
// ONE 16-byte fragment load per 16-row block from L2 (fragment order: 1 KB contiguous per wave) and four ds_read_b32 per
// column block, each feeding 4 MFMAs of 32 cycles.  Tasks are ordered (tile, row block); every workgroup (8 waves, two
// per SIMD: one wave's loads, stores and waits under the other's MFMAs) takes one contiguous range, equal to within one
// task, stages a tile when its range enters it and deals the tile's tasks round-robin to its waves, so that a SIMD's two
// waves differ by at most one task.
#include <algorithm>
#include <type_traits>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_mfma_h.h"
#include "wn_iaf_c.h"

namespace {

constexpr int GF_THREADS = 512;                   // default: 8 waves, two per SIMD (template parameter NW of the kernel)
constexpr int GF_CIN = 256;                       // input channels of both GEMMs (enc / the upsampler's hidden layer)
constexpr int GF_HALO = 4;                        // deconv: frames staged left AND right of a tile's 64 (taps <= 5)

struct GfArgs {
    const float* src;          // activation rows, planar fp32: [batch][256][src_rs]
    int64_t src_bstride;       // floats per batch element
    int src_rs;                // floats per row
    int src_col0;              // source column of tile 0's first staged column (multiple of 4)
    const float* wblob;
    const unsigned* tab;       // per row block: {float offset of its first K-group, aux} -- aux: cond = float offset of its 64
                               // biases in lane order; deconv = (phase << 8) | input column shift d
    int a_kstride;             // floats between consecutive K-groups of a row block
    int nkg;                   // K-groups (16 x taps)
    int R;                     // row blocks
    int tiles_per_row;
    int64_t ntasks;
    int parts;                 // > 0: workgroup g takes part g % parts of tile g / parts (few tiles: no range crosses a tile, every
                               // SIMD gets the same number of tasks); 0: contiguous ranges of the (tile, row block) list
    // cond output
    float* C;
    int64_t c_bstride, NCB;
    // deconv output: phase-major yp[batch][phase][cout][Lp]
    float* yp;
    int S, cout, Lp;
};

__device__ inline f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int NB, bool DECONV, int NW = GF_THREADS / 64>
__global__ __launch_bounds__(NW * 64) void gemm_f32_kernel(const GfArgs A) {
    constexpr int GF_THREADS = NW * 64, GF_WAVES = NW;
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    constexpr int NC = 16 * NB;                                   // columns of a tile
    constexpr int PITCH = DECONV ? NC + 2 * GF_HALO : NC;         // floats per LDS row (one input channel)
    constexpr int QUADS = PITCH / 4;                              // column quads per row
    constexpr int NI = (GF_CIN / 4) * QUADS;                      // staging items per tile: (channel group of 4) x (column quad)
    constexpr int NST = (NI + GF_THREADS - 1) / GF_THREADS;       // ... per thread
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;

    // contiguous task range of this workgroup; every XCD takes one contiguous eighth (its L2 then holds one stretch of
    // the activations and, for the cond GEMM, sees the 4 MB fragment table tile after tile)
    int64_t t_lo, t_hi;
    const int64_t U = A.ntasks;
    if (A.parts > 0) {
        const int tile = blockIdx.x / A.parts, part = blockIdx.x - tile * A.parts;
        t_lo = (int64_t)tile * A.R + A.R * part / A.parts;
        t_hi = (int64_t)tile * A.R + A.R * (part + 1) / A.parts;
    } else {
        // ranges are cut in OCTETS of tasks (R is a multiple of 8, so a tile is a whole number of octets): a round of the
        // eight waves is then always full, where a range that starts and ends anywhere pays two partial rounds of ~19
        const int64_t NO = U / GF_WAVES;
        if ((gridDim.x & 7) == 0) {
            const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3, nx = gridDim.x >> 3;
            const int64_t x_lo = NO * xcd / 8, x_hi = NO * (xcd + 1) / 8;
            t_lo = (x_lo + (x_hi - x_lo) * i / nx) * GF_WAVES;
            t_hi = (x_lo + (x_hi - x_lo) * (i + 1) / nx) * GF_WAVES;
        } else {
            t_lo = NO * blockIdx.x / gridDim.x * GF_WAVES;
            t_hi = NO * (blockIdx.x + 1) / gridDim.x * GF_WAVES;
        }
        if (blockIdx.x + 1 == gridDim.x) t_hi = U;       // (R not a multiple of 8: the remainder goes to the last workgroup)
    }
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)A.wblob, 0, 0x7ffffff0, 0x00020000);

    for (int64_t t0 = t_lo; t0 < t_hi;) {
        const int tile = (int)(t0 / A.R);
        const int64_t seg_end = min(t_hi, (int64_t)(tile + 1) * A.R);
        const int b = tile / A.tiles_per_row;
        const int jt = tile - b * A.tiles_per_row;
        // ---- stage the tile: 256 channel rows x PITCH columns, all loads of a thread first, then its LDS stores ----
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(A.src + (size_t)b * A.src_bstride), 0, GF_CIN * A.src_rs * 4, 0x00020000);
            const int col0 = A.src_col0 + NC * jt;
            // LDS layout [K4 group][column][4]: the four input channels one lane multiplies in the four K-steps of a K-group
            // side by side -- ONE ds_read_b128 per 16 MFMAs in the inner loop.  Group g holds channels 4 g + (0..3) of the
            // student packs (K-step jj <-> channel 16 cg + 4 q + jj) / channels 16 (g / 4) + 4 (0..3) + g % 4 of the
            // upsampler pack (16 cg + 4 jj + q).  An item = four rows x four columns, transposed in registers.
            f4 tmp[NST][4];
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int i = k * GF_THREADS + (int)threadIdx.x;
                const int grp = i / QUADS, qd = i - grp * QUADS;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ch = DECONV ? 16 * (grp >> 2) + 4 * c + (grp & 3) : 4 * grp + c;
                    tmp[k][c] = (f4){0.f, 0.f, 0.f, 0.f};
                    if (i < NI) tmp[k][c] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (ch * A.src_rs + col0 + 4 * qd) * 4, 0, 0));
                }
            }
            __syncthreads();                      // the previous tile's operand reads are done
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int i = k * GF_THREADS + (int)threadIdx.x;
                const int grp = i / QUADS, qd = i - grp * QUADS;
                if (i < NI) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<f4*>(ldsf + (size_t)((grp * PITCH + 4 * qd + e) * 4)) =
                            (f4){tmp[k][0][e], tmp[k][1][e], tmp[k][2][e], tmp[k][3][e]};
                }
            }
            __syncthreads();
        }
        // ---- the tile's tasks of this range, round-robin over the waves ----
        for (int64_t t = t0 + wave; t < seg_end; t += GF_WAVES) {
            const int rbi = (int)(t - (int64_t)tile * A.R);
            const unsigned a_off = A.tab[2 * rbi], aux = A.tab[2 * rbi + 1];
            const int ao = (int)a_off * 4 + lane * 16;            // byte offset of this lane's fragment words
            const int ks4 = A.a_kstride * 4;
            const int dshift = DECONV ? (int)(aux & 255u) : 0;
            f4 acc[4][NB];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = (f4){0.f, 0.f, 0.f, 0.f};
            // fragments NBUF - 1 K-groups ahead of their use: one K-group of the narrow (64-column) deconv task is 2 000
            // cycles of MFMA per wave, less than an L2 round trip under this kernel's own 5 TB/s of fragment traffic
            constexpr int NBUF = DECONV ? 4 : 2;
            f4 a[NBUF][4];
#pragma unroll
            for (int p = 0; p < NBUF - 1; ++p)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    a[p][mb] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rw, ao + mb * 1024, p * ks4, 0));
            // one K-group: 16 input channels x the tile's NB column blocks = 16 NB MFMAs.  K-step jj of the group multiplies
            // channel 16 cg + 4 q + jj (the student packs, wn_pack_iaf) or 16 cg + 4 jj + q (the upsampler pack,
            // wn_pack_deconv) of lane group q: word jj of LDS group 4 cg + q; tap j = kg / 16 of the deconv reads x[f + d - j]
            auto brow_of = [&](int kg) -> const f4* {
                const int colb = DECONV ? GF_HALO + dshift - (kg >> 4) : 0;
                return reinterpret_cast<const f4*>(ldsf) + (4 * (kg & 15) + q) * PITCH + colb + n;
            };
            f4 bwn = brow_of(0)[0];               // operand word of the NEXT column block, one block (16 MFMAs) ahead of its use
            auto kgroup = [&](int kg, auto cur_c) {
                constexpr int cur = decltype(cur_c)::value;
                {
                    // UNCONDITIONAL (the last NBUF - 1 groups reload the task's last fragments): behind a branch the compiler
                    // cannot count the loads in flight and waits for ALL of them -- the ones just issued included -- before the
                    // group's first MFMA (s_waitcnt vmcnt(0) where vmcnt(4 (NBUF - 1)) is meant): an L2 round trip per K-group
                    const int kn = min(kg + NBUF - 1, A.nkg - 1);
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        a[(cur + NBUF - 1) % NBUF][mb] =
                            __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rw, ao + mb * 1024, kn * ks4, 0));
                }
                // ... and pinned at the head of the group (left alone the scheduler sinks them to their first use)
                __builtin_amdgcn_sched_barrier(0);
                const f4* brow = brow_of(kg);
                const f4* brow1 = brow_of(min(kg + 1, A.nkg - 1));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const f4 bw = bwn;
                    bwn = nb + 1 < NB ? brow[16 * (nb + 1)] : brow1[0];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = mfma4(a[cur][mb][jj], bw[jj], acc[mb][nb]);
                    // the next column block's operand word first (ds_read_b128), then this block's 16 MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            for (int kg = 0; kg < A.nkg; kg += NBUF) {            // (nkg is a multiple of 16)
                kgroup(kg, std::integral_constant<int, 0>{});
                kgroup(kg + 1, std::integral_constant<int, 1>{});
                if (NBUF == 4) {
                    kgroup(kg + 2, std::integral_constant<int, 2 % NBUF>{});
                    kgroup(kg + 3, std::integral_constant<int, 3 % NBUF>{});
                }
            }
            if (!DECONV) {
                // bias of the row block (the layer's dilated-conv + cond biases / the head's out1 + cond biases, in lane order)
                // and the store in accumulator layout: [row block][t / 16][mb][lane][4], written once and read once a
                // gigabyte later -> non-temporal like the split-fp16 form's C
                const f4* bp = reinterpret_cast<const f4*>(A.wblob + aux) + q * 4;
                const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(A.C + (size_t)b * A.c_bstride + ((size_t)rbi * A.NCB) * 1024), 0, (int)A.NCB * 4096, 0x00020000);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const f4 bias = bp[mb];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const f4 v = acc[mb][nb] + bias;
                        buf_st4<WN_C_ST_AUX>(__builtin_bit_cast(wn_u4, v), rc, lane * 16, ((NB * jt + nb) * 4 + mb) * 1024);
                    }
                }
            } else {
                const int p = (int)(aux >> 8);                    // output phase; its 64-channel group follows from rbi
                const int cgo = rbi % (A.cout / 64);
                float* yb = A.yp + (((size_t)b * A.S + p) * A.cout + 64 * cgo) * A.Lp + NC * jt + n;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* yr = yb + (size_t)(16 * mb + 4 * q + r) * A.Lp;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) yr[16 * nb] = acc[mb][nb][r];
                    }
            }
        }
        t0 = seg_end;
    }
}

// (measured, A/B on one box: 64-column tiles with sixteen waves of 128 registers -- four per SIMD -- 1 168 us against 1 167 us for
// this form: the wave count is not what the GEMM waits for)
constexpr int GF_COND_LDS = GF_CIN * 128 * 4;                       // 128 KB
constexpr int GF_DC_LDS = GF_CIN * (64 + 2 * GF_HALO) * 4;          // 72 KB

int gf_grid(int64_t ntasks, int num_cu) {
    int grid = (int)std::min<int64_t>(ntasks, num_cu);
    if (grid >= 8) grid = grid / 8 * 8;                             // XCD-aware ranges: a multiple of 8
    return grid;
}

}  // namespace

int wn_iaf_f_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<8, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GF_COND_LDS));
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<4, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GF_DC_LDS));
    return WN_OK;
}

// Conditioning GEMM of the fp32 form: R row blocks (tab: {fragment offset, bias offset} pairs in natural row-block
// order) over enc columns [c0, c0 + T) -> C[batch][row block][T / 16][4][64][4].  enc: planar fp32 rows [B][256][TE].
void wn_iaf_f_cond(const wn_handle* h, const float* enc, const unsigned* tab, int R, float* C, int64_t c_bstride, int64_t TE,
                   int c0, int B, int64_t T, hipStream_t st) {
    GfArgs A{};
    A.src = enc;
    A.src_bstride = (int64_t)IAF_CD * TE;
    A.src_rs = (int)TE;
    A.src_col0 = c0;
    A.wblob = h->d_blob;
    A.tab = tab;
    A.a_kstride = 1024;
    A.nkg = IAF_CD / 16;
    A.R = R;
    A.tiles_per_row = (int)(T / 128);
    A.ntasks = (int64_t)B * A.tiles_per_row * R;
    A.C = C;
    A.c_bstride = c_bstride;
    A.NCB = T / 16;
    hipLaunchKernelGGL((gemm_f32_kernel<8, false>), dim3(gf_grid(A.ntasks, h->num_cu)), dim3(GF_THREADS), GF_COND_LDS, st, A);
}

// T must be a multiple of 128 (every generate length is a multiple of 2^(num_stages-1) >= 64; 128 for the shipped 10 stages)
bool wn_iaf_f_cond_ok(int64_t T, int c0) { return T % 128 == 0 && c0 % 4 == 0; }

// Last upsampler layer as a frame-axis GEMM (see the head of this file).  x: planar rows [B][256][xs] with DC_XOFF zero
// columns in front; tab: R = S * cout / 64 pairs {fragment offset, (phase << 8) | d}; yp: [B][S][cout][Lp] phase-major.
void wn_deconv_f_gemm(const wn_handle* h, const float* x, int xs, int xoff, const unsigned* tab, int R, int taps, int a_kstride,
                      float* yp, int S, int cout, int L, int Lp, int B, hipStream_t st) {
    GfArgs A{};
    A.src = x;
    A.src_bstride = (int64_t)GF_CIN * xs;
    A.src_rs = xs;
    A.src_col0 = xoff - GF_HALO;
    A.wblob = h->d_blob;
    A.tab = tab;
    A.a_kstride = a_kstride;
    A.nkg = 16 * taps;
    A.R = R;
    A.tiles_per_row = Lp / 64;
    A.ntasks = (int64_t)B * A.tiles_per_row * R;
    A.yp = yp;
    A.S = S;
    A.cout = cout;
    A.Lp = Lp;
    (void)L;
    // One utterance of 4.8 s is 60 tiles x 80 row blocks = 4 800 tasks = 4.69 per SIMD.  Contiguous ranges of 18.75 tasks
    // cross a tile boundary in a quarter of the workgroups, and the barrier at the boundary makes the two sides' rounds add up
    // (6 task times instead of 5: 378 us measured for 273 us of matrix-pipe work).  With fewer tiles than CUs every tile is cut
    // into `parts` equal pieces instead -- 4 x 20 tasks here: 240 workgroups, each SIMD exactly 3 + 2 tasks.
    const int tiles = B * A.tiles_per_row;
    int grid;
    if (tiles <= h->num_cu && R >= 16) {
        A.parts = std::max(1, std::min(h->num_cu / tiles, R / 8));
        grid = tiles * A.parts;
    } else {
        A.parts = 0;
        grid = gf_grid(A.ntasks, h->num_cu);
    }
    hipLaunchKernelGGL((gemm_f32_kernel<4, true>), dim3(grid), dim3(GF_THREADS), GF_DC_LDS, st, A);
}
