// Full-sequence (teacher-forced) forward of the WaveNet teacher: wavenet/wavenet.py:180-291.
//
// The autoregressive path (wn_ar.hip) evaluates one sample per step; scoring or checking a
// whole utterance that way costs T dependent steps.  Given the audio, every layer is a dense
// GEMM over time, so the same network runs here as a chain of split-fp16 MFMA GEMMs on
// channel-major activations (the layouts of the IAF path):
//   l, m, enc : G4 words (wn_iaf_h.hip) -- the MFMA B operand of a lane is one 16-byte load
//   s, out1   : fp32 in the MFMA accumulator layout [t/16][16-row block][lane][4] -- written and
//               read-modify-written with one 16-byte access per lane, and the accumulator
//               registers of two row blocks ARE the B operand of a K-step of the next GEMM
// One kernel template: C[64 or 128 rows][256 columns] per workgroup, K walked over up to four operand
// segments (three dilated taps of l + enc; m; relu(s) + enc; relu(out1)), weights as A fragments
// staged through LDS in double-buffered chunks shared by the four waves.  Epilogues: gate ->
// m; residual add -> l and skip accumulate -> s; plain store; time-major out_params.
#include <algorithm>
#include <cmath>

#include "wn_internal.h"
#include "wn_codec.h"
#include "wn_pack_h.h"
#include "wn_mfma_h.h"

namespace {

constexpr int TG_NT = 4;                 // 16-column blocks per wave
constexpr int TG_TN = 4 * 16 * TG_NT;    // columns per workgroup
constexpr int TG_KC = 4;                 // K-steps of weights per LDS stage
constexpr int TG_XP = 64;                // zero left pad of the scaled input row

enum { TG_SRC_G4 = 0, TG_SRC_ACC_RELU = 1 };
enum { TG_EPI_GATE = 0, TG_EPI_RS = 1, TG_EPI_ACC = 2, TG_EPI_OUT = 3 };

struct TgSeg {
    const unsigned* base;   // G4 words or accumulator-layout floats
    long long bstride;      // words per batch element
    int rowlen;             // G4: columns per group row; ACC: 16-row blocks per column block
    int col0;               // G4: column of t = 0 (left pad, tap shift, centre crop)
    int nks;                // 32-channel K-steps in this segment
    int ng;                 // G4: group rows per plane
    int kind;
};

struct TgArgs {
    TgSeg seg[4];
    int nseg, nks;
    const unsigned* wp;     // A fragments [m-tile][K-step][4 row blocks][plane][lane][4]
    const float* bias;      // [m-tile][64], tile-local row order
    float inv_scale;
    long long T;            // valid columns (only the time-major store is guarded)
    unsigned* og4;          // GATE: m;  RS: l (updated in place)
    long long og4_bstride;
    int og4_rowlen, og4_col0, og4_ng;
    float* oacc;            // RS: s (accumulated);  ACC: destination
    long long oacc_bstride;
    int oacc_nmb;
    int res_mtiles;         // RS: m-tiles below this are residual rows, the rest skip rows
    float* otm;             // OUT: [B][T][ow]
    int ow;
};

// U = 64-row m-tiles per workgroup (1 or 2): two tiles halve the re-reads of the activation operand,
// which bound the kernel (each operand word is fetched once per workgroup row of the grid).
template <int EPI, int U>
__global__ __launch_bounds__(256, 2) void tg_gemm_kernel(const TgArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2][TG_KC * 4 * 512];
    constexpr int KC = TG_KC / U;                 // K-steps per LDS stage (64 KB in both shapes)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int mt0 = blockIdx.y * U, b = blockIdx.z;
    const int t0 = blockIdx.x * TG_TN + wave * 16 * TG_NT;      // first column of this wave
    const int nchunk = (a.nks + KC - 1) / KC;

    f4 acc[4 * U][TG_NT];
#pragma unroll
    for (int mb = 0; mb < 4 * U; ++mb)
#pragma unroll
        for (int e = 0; e < TG_NT; ++e) acc[mb][e] = (f4){0.f, 0.f, 0.f, 0.f};

    const wn_u4* wsrc = reinterpret_cast<const wn_u4*>(a.wp);
    auto stage = [&](int chunk, int buf) {
#pragma unroll
        for (int kl = 0; kl < KC; ++kl) {
            const int ks = chunk * KC + kl;
            if (ks < a.nks) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const wn_u4* src = wsrc + ((size_t)(mt0 + u) * a.nks + ks) * 512;
                    wn_u4* dst = reinterpret_cast<wn_u4*>(lds[buf]) + (kl * U + u) * 512;
                    dst[threadIdx.x] = src[threadIdx.x];
                    dst[threadIdx.x + 256] = src[threadIdx.x + 256];
                }
            }
        }
    };
    // operand words of K-step ks for the TG_NT column blocks of this lane
    auto loadB = [&](int ks, wn_u4 (&vh)[TG_NT], wn_u4 (&vl)[TG_NT]) {
        int si = 0, ksl = ks;
        while (si + 1 < a.nseg && ksl >= a.seg[si].nks) { ksl -= a.seg[si].nks; ++si; }
        const TgSeg& s = a.seg[si];
        if (s.kind == TG_SRC_G4) {
            const wn_u4* p = reinterpret_cast<const wn_u4*>(s.base + (size_t)b * s.bstride) +
                             (size_t)(4 * ksl + q) * s.rowlen + s.col0 + t0 + n;
            const size_t lo = (size_t)s.ng * s.rowlen;
#pragma unroll
            for (int e = 0; e < TG_NT; ++e) {
                vh[e] = p[16 * e];
                vl[e] = p[lo + 16 * e];
            }
        } else {
            // accumulator layout: the registers of row blocks 2ksl, 2ksl+1 of lane (q, n) are the 8 k-slots
            const f4* p = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(s.base) + (size_t)b * s.bstride) +
                          ((size_t)(t0 >> 4) * s.rowlen + 2 * ksl) * 64 + lane;
#pragma unroll
            for (int e = 0; e < TG_NT; ++e) {
                const f4 v0 = p[(size_t)e * s.rowlen * 64], v1 = p[(size_t)e * s.rowlen * 64 + 64];
                unsigned h0, l0, h1, l1, h2, l2, h3, l3;
                wn_split_pair(fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f), h0, l0);
                wn_split_pair(fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f), h1, l1);
                wn_split_pair(fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f), h2, l2);
                wn_split_pair(fmaxf(v1[2], 0.f), fmaxf(v1[3], 0.f), h3, l3);
                vh[e] = (wn_u4){h0, h1, h2, h3};
                vl[e] = (wn_u4){l0, l1, l2, l3};
            }
        }
    };

    stage(0, 0);
    wn_u4 b1h[TG_NT], b1l[TG_NT];
    loadB(0, b1h, b1l);
    __syncthreads();
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunk) stage(chunk + 1, buf ^ 1);
        const wn_u4* Al = reinterpret_cast<const wn_u4*>(lds[buf]) + lane;
#pragma unroll
        for (int kl = 0; kl < KC; ++kl) {
            const int ks = chunk * KC + kl;
            if (ks < a.nks) {
                wn_u4 vh[TG_NT], vl[TG_NT];
#pragma unroll
                for (int e = 0; e < TG_NT; ++e) { vh[e] = b1h[e]; vl[e] = b1l[e]; }
                if (ks + 1 < a.nks) loadB(ks + 1, b1h, b1l);
#ifndef WN_TG_APIPE
#define WN_TG_APIPE 1
#endif
                if (WN_TG_APIPE) {
                    // weight fragments one row block (12 MFMAs) ahead of their use: read where they are used, each pair of
                    // ds_read_b128 sat in front of its own MFMAs and the wave waited out the LDS latency twelve MFMAs at a time
                    wn_u4 ahn = Al[(kl * 4 * U) * 128], aln = Al[(kl * 4 * U) * 128 + 64];
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                    for (int mb = 0; mb < 4 * U; ++mb) {
                        const wn_u4 ah = ahn, al = aln;
                        if (mb + 1 < 4 * U) {
                            ahn = Al[(kl * 4 * U + mb + 1) * 128];
                            aln = Al[(kl * 4 * U + mb + 1) * 128 + 64];
                        }
#pragma unroll
                        for (int e = 0; e < TG_NT; ++e) acc[mb][e] = mfma3(ah, al, vh[e], vl[e], acc[mb][e]);
                        if (mb + 1 < 4 * U) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3 * TG_NT, 0);
                    }
                } else {
#pragma unroll
                for (int mb = 0; mb < 4 * U; ++mb) {
                    const wn_u4 ah = Al[(kl * 4 * U + mb) * 128], al = Al[(kl * 4 * U + mb) * 128 + 64];
#pragma unroll
                    for (int e = 0; e < TG_NT; ++e) acc[mb][e] = mfma3(ah, al, vh[e], vl[e], acc[mb][e]);
                }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue per 64-row tile: row 16 mb + 4 q + r of the tile, column t0 + 16 e + n ----
    const float inv = a.inv_scale;
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int mt = mt0 + u;
    const f4* bias4 = reinterpret_cast<const f4*>(a.bias + (size_t)mt * 64) + q;
    f4 bv[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) bv[mb] = bias4[mb * 4];
    f4 (&ac)[4][TG_NT] = *reinterpret_cast<f4 (*)[4][TG_NT]>(&acc[4 * u]);
    if (EPI == TG_EPI_GATE) {
        // rows 0-31: sigmoid half of gate channels 32 mt .. +31, rows 32-63: their tanh half
        wn_u4* o = reinterpret_cast<wn_u4*>(a.og4 + (size_t)b * a.og4_bstride) +
                   (size_t)(4 * mt + q) * a.og4_rowlen + a.og4_col0 + t0 + n;
        const size_t lo = (size_t)a.og4_ng * a.og4_rowlen;
#pragma unroll
        for (int e = 0; e < TG_NT; ++e) {
            wn_u4 gh, gl;
#pragma unroll
            for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    float g[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        g[k] = sigmoidf_(fmaf(ac[mg][e][2 * rp + k], inv, bv[mg][2 * rp + k])) *
                               tanhf_(fmaf(ac[mg + 2][e][2 * rp + k], inv, bv[mg + 2][2 * rp + k]));
                    unsigned hw, lw;
                    wn_split_pair(g[0], g[1], hw, lw);
                    gh[2 * mg + rp] = hw;
                    gl[2 * mg + rp] = lw;
                }
            o[16 * e] = gh;
            o[lo + 16 * e] = gl;
        }
    } else if (EPI == TG_EPI_RS && mt < a.res_mtiles) {
        // residual rows 64 mt .. +63: l += res  (wavenet.py:272-274), two 32-channel operand groups
        const size_t lo = (size_t)a.og4_ng * a.og4_rowlen;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            wn_u4* o = reinterpret_cast<wn_u4*>(a.og4 + (size_t)b * a.og4_bstride) +
                       (size_t)(4 * (2 * mt + st) + q) * a.og4_rowlen + a.og4_col0 + t0 + n;
#pragma unroll
            for (int e = 0; e < TG_NT; ++e) {
                const wn_u4 oh = o[16 * e], ol = o[lo + 16 * e];
                wn_u4 nh, nl;
#pragma unroll
                for (int mg = 0; mg < 2; ++mg)
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        const int mb = 2 * st + mg;
                        float l0, l1;
                        wn_join_pair(oh[2 * mg + rp], ol[2 * mg + rp], l0, l1);
                        l0 += fmaf(ac[mb][e][2 * rp], inv, bv[mb][2 * rp]);
                        l1 += fmaf(ac[mb][e][2 * rp + 1], inv, bv[mb][2 * rp + 1]);
                        unsigned hw, lw;
                        wn_split_pair(l0, l1, hw, lw);
                        nh[2 * mg + rp] = hw;
                        nl[2 * mg + rp] = lw;
                    }
                o[16 * e] = nh;
                o[lo + 16 * e] = nl;
            }
        }
    } else if (EPI == TG_EPI_RS || EPI == TG_EPI_ACC) {
        // accumulator-layout destination: s += skip (wavenet.py:275-277) or a plain store
        const int mrow = EPI == TG_EPI_RS ? mt - a.res_mtiles : mt;
        f4* o = reinterpret_cast<f4*>(a.oacc + (size_t)b * a.oacc_bstride) +
                ((size_t)(t0 >> 4) * a.oacc_nmb + 4 * mrow) * 64 + lane;
#pragma unroll
        for (int e = 0; e < TG_NT; ++e)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f4* p = o + ((size_t)e * a.oacc_nmb + mb) * 64;
                f4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(ac[mb][e][r], inv, bv[mb][r]);
                if (EPI == TG_EPI_RS) v += *p;
                *p = v;
            }
    } else {
        // out_params, the reference's [B][T][out_width]
#pragma unroll
        for (int e = 0; e < TG_NT; ++e) {
            const long long t = t0 + 16 * e + n;
            if (t >= a.T) continue;
            float* o = a.otm + ((size_t)b * a.T + t) * a.ow;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 64 * mt + 16 * mb + 4 * q + r;
                    if (c < a.ow) o[c] = fmaf(ac[mb][e][r], inv, bv[mb][r]);
                }
        }
    }
    }
}

// scaled input row: zero pad | encode(wav)  (wavenet.py:412-418 encoding, masked.py:39-52 shift by reading t-1)
__global__ void tg_input_kernel(const float* __restrict__ wav, float* __restrict__ xs, long long T, long long Tp,
                                int mu) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TG_XP + Tp) return;
    const long long t = i - TG_XP;
    float v = 0.f;
    if (t >= 0 && t < T) {
        v = wav[(size_t)b * T + t];
        if (mu) v = wn_mu_law_scaled(v);
    }
    xs[(size_t)b * (TG_XP + Tp) + i] = v;
}

// conv_start over shift_right(x) (wavenet.py:223-226) -> l in G4, plus the zero left pad of the rows
__global__ __launch_bounds__(256) void tg_start_kernel(const float* __restrict__ xs, const float* __restrict__ wb,
                                                       unsigned* __restrict__ l, int W, long long Tp, long long RS) {
    const int b = blockIdx.z, g = blockIdx.y, NG = W / 8;
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // column incl. left pad
    if (c >= RS) return;
    const long long t = c - IAF_LP;
    wn_u4 hw = (wn_u4){0u, 0u, 0u, 0u}, lw = hw;
    if (t >= 0) {
        const float* xp = xs + (size_t)b * (TG_XP + Tp) + TG_XP + t;
        const float x0 = xp[-3], x1 = xp[-2], x2 = xp[-1];
        const int s = g >> 2, kg = g & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = 2 * (16 * s + 8 * (i >> 1) + 2 * kg + (i & 1));
            float o[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                o[hh] = wb[3 * W + ch + hh] + wb[ch + hh] * x0 + wb[W + ch + hh] * x1 + wb[2 * W + ch + hh] * x2;
            unsigned a, c2;
            wn_split_pair(o[0], o[1], a, c2);
            hw[i] = a;
            lw[i] = c2;
        }
    }
    unsigned* base = l + (size_t)b * W * RS;
    *reinterpret_cast<wn_u4*>(base + ((size_t)g * RS + c) * 4) = hw;
    *reinterpret_cast<wn_u4*>(base + ((size_t)(NG + g) * RS + c) * 4) = lw;
}

struct TLayout {
    long long T, Tp, TE, RS;
    int c0;
    size_t enc, l, m, s, h1, xs, scratch, total;
};

TLayout t_layout(const wn_handle* h, int B, int F, long long T) {
    const wn_config& c = h->cfg;
    TLayout L;
    L.T = T;
    L.Tp = (T + TG_TN - 1) / TG_TN * TG_TN;
    L.TE = (long long)F * h->frame_shift;
    L.c0 = (int)((L.TE - T) / 2);                           // wavenet.py:76-85
    L.RS = IAF_LP + L.Tp;
    size_t o = 0;
    auto carve = [&](size_t floats) { size_t r = o; o += align_up(floats * sizeof(float), 256); return r; };
    L.enc = carve((size_t)B * c.deconv_width * (L.TE + TG_TN) + 64);
    L.l = carve((size_t)B * c.width * L.RS);
    L.m = carve((size_t)B * (c.gate_width / 2) * L.Tp);
    L.s = carve((size_t)B * c.skip_width * L.Tp);
    L.h1 = carve((size_t)B * c.skip_width * L.Tp);
    L.xs = carve((size_t)B * (TG_XP + L.Tp));
    L.scratch = o;
    o += wn_deconv_scratch_bytes(h, B, F);
    L.total = o;
    return L;
}

template <int EPI>
void tg_launch(const TgArgs& a, int mtiles, int B, long long Tp, hipStream_t st) {
    if (mtiles % 2 == 0) {
        dim3 g((unsigned)(Tp / TG_TN), mtiles / 2, B);
        hipLaunchKernelGGL((tg_gemm_kernel<EPI, 2>), g, dim3(256), 0, st, a);
    } else {
        dim3 g((unsigned)(Tp / TG_TN), mtiles, B);
        hipLaunchKernelGGL((tg_gemm_kernel<EPI, 1>), g, dim3(256), 0, st, a);
    }
}

}  // namespace

// ---- packing: A fragments of a row-major [M][K] matrix, 64-row tiles, rows picked by rowfn ----
template <class RowFn>
static void pack_tiles(std::vector<float>& blob, size_t dst_off, const float* src, int ld, int K, int mtiles, float scale,
                       RowFn rowfn) {
    const int nks = K / 32;
    unsigned* P = reinterpret_cast<unsigned*>(blob.data() + dst_off);
    for (int mt = 0; mt < mtiles; ++mt)
        for (int ks = 0; ks < nks; ++ks)
            for (int mb = 0; mb < 4; ++mb)
                pack_afrag(P + (((size_t)mt * nks + ks) * 4 + mb) * 512, [&](int e, int kg, int i16) {
                    const int row = rowfn(mt * 64 + 16 * mb + i16);
                    if (row < 0) return 0.f;
                    return scale * src[(size_t)row * ld + 32 * ks + 16 * (e >> 2) + 4 * kg + (e & 3)];
                });
}

int wn_pack_teacher(wn_handle* h, std::vector<float>& blob) {
    const wn_config& c = h->cfg;
    const int W = c.width, S = c.skip_width, G = c.gate_width, H = G / 2, Cd = c.deconv_width, OW = c.out_width;
    TeacherPack& T = h->teacher;
    const ArPack& A = h->ar;
    auto reserve = [&](size_t words) {
        blob.resize(align_up(blob.size(), 64));
        const size_t off = blob.size();
        blob.resize(off + words);
        return off;
    };
    // one GEMM: fragments + tile-ordered bias from the row-major matrices wn_pack_ar already built
    auto gemm = [&](size_t w_off, size_t b_off, int M, int K, int mtiles, auto rowfn) {
        TeacherGemmPack g;
        std::vector<float> src(blob.begin() + w_off, blob.begin() + w_off + (size_t)M * K);
        std::vector<float> bsrc(blob.begin() + b_off, blob.begin() + b_off + M);
        const float sc = pick_scale(src.data(), src.size());
        g.inv_scale = 1.0f / sc;
        g.nks = K / 32;
        g.mtiles = mtiles;
        g.w_off = reserve((size_t)mtiles * g.nks * 4 * 512);
        pack_tiles(blob, g.w_off, src.data(), K, K, mtiles, sc, rowfn);
        g.b_off = reserve((size_t)mtiles * 64);
        for (int i = 0; i < mtiles * 64; ++i) {
            const int row = rowfn(i);
            blob[g.b_off + i] = row < 0 ? 0.f : bsrc[row];
        }
        return g;
    };
    auto ident = [](int M) { return [M](int i) { return i < M ? i : -1; }; };
    T.skip_start = gemm(A.wss_off, A.bss_off, S, W, S / 64, ident(S));
    for (const ArLayerPack& lp : A.layers) {
        TeacherLayerPack tl;
        tl.dilation = lp.dilation;
        // m-tile j: sigmoid rows 32j..32j+31 then their tanh partners H+32j.. (wavenet.py:264-269)
        tl.gate = gemm(lp.wd_off, lp.bd_off, G, 3 * W + Cd, H / 32,
                       [H](int i) { const int j = i / 64, lr = i % 64; return lr < 32 ? 32 * j + lr : H + 32 * j + lr - 32; });
        tl.rs = gemm(lp.wrs_off, lp.brs_off, W + S, H, (W + S) / 64, ident(W + S));
        T.layers.push_back(tl);
    }
    T.out1 = gemm(A.wo1_off, A.bo1_off, S, S + Cd, S / 64, ident(S));
    T.out2 = gemm(A.wo2_off, A.bo2_off, OW, S, (OW + 63) / 64, ident(OW));
    return WN_OK;
}

namespace {
// Teacher scoring: log-likelihood per sample of the (encoded) audio under the output parameters of Wavenet.feed_forward --
// the per-sample term of Wavenet.calculate_loss (wavenet/wavenet.py:293-316): loss_func.mol_log_probs (loss_func.py:22-63),
// gauss_log_prob (:66-75,104-119) and the cross entropy of ce_loss (:128-133), on the targets Wavenet.encode_signal derives
// from the raw audio (wavenet.py:157-178).  One wave per sample; lane i owns mixture i / strides over the classes.
// The discretised-logistic mass cdf(x + 1/Q) - cdf(x - 1/Q) is evaluated as sigma(a) sigma(-b) (1 - exp(-(a - b))) with
// a - b = 2 inv_s / Q formed directly: the difference of two float32 sigmoids the reference's formula takes loses all but
// two digits of it at Q = 65 536 (the float64 evaluation of the reference's formula is what the tests compare with).
__device__ inline float tl_softplus(float v) { return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))); }
__device__ inline float tl_sigmoid(float v) {
    const float e = expf(-fabsf(v));
    return v >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}
__device__ inline float tl_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ inline float tl_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__global__ __launch_bounds__(256) void tg_log_prob_kernel(const float* __restrict__ out, const float* __restrict__ wav,
                                                          float* __restrict__ lp, long long n, int ow, int loss, int Q, int mu) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const float* o = out + (size_t)i * ow;
    const float x = wav[i];
    const float xt = mu ? wn_mu_law_scaled(x) : x;                 // real_targets (wavenet.py:165-174)
    const float NEG = -__builtin_inff();
    float res;
    if (loss == WN_LOSS_MOL) {
        const int M = ow / 3;
        float v = NEG, lg = NEG;
        if (lane < M) {
            lg = o[lane];
            const float mean = o[M + lane], ls = fmaxf(o[2 * M + lane], -7.0f);
            const float inv = expf(-ls), c = xt - mean, iq = 1.0f / (float)Q;
            const float plus = inv * (c + iq), mn = inv * (c - iq);
            const float max_thres = ((float)(Q - 1) - 0.5f) / ((float)Q * 0.5f) - 1.0f, min_thres = 0.5f / ((float)Q * 0.5f) - 1.0f;
            const float delta = tl_sigmoid(plus) * tl_sigmoid(-mn) * (-expm1f(-2.0f * inv * iq));
            v = xt < min_thres ? plus - tl_softplus(plus) : (xt > max_thres ? -tl_softplus(mn) : logf(fmaxf(delta, 1e-12f)));
        }
        const float lmax = tl_wave_max(lg);
        const float lse = lmax + logf(tl_wave_sum(lane < M ? expf(lg - lmax) : 0.f));
        v = lane < M ? v + (lg - lse) : NEG;
        const float vmax = tl_wave_max(v);
        res = vmax + logf(tl_wave_sum(lane < M ? expf(v - vmax) : 0.f));
    } else if (loss == WN_LOSS_GAUSS) {
        const float ls = fmaxf(o[1], -7.0f), z = (xt - o[0]) * expf(-ls);
        res = -0.5f * z * z - ls - 0.9189385332046727f;             // Normal(mean, exp(ls)).log_prob(x)
    } else {
        // cate_targets (wavenet.py:166-176): the quantised audio shifted to [0, Q)
        int label = (mu ? (int)floorf(wn_mu_law_scaled(x) * 128.0f) : (int)floorf(x * (float)Q * 0.5f)) + Q / 2;
        label = min(max(label, 0), Q - 1);
        float m = NEG;
        for (int k = lane; k < ow; k += 64) m = fmaxf(m, o[k]);
        m = tl_wave_max(m);
        float sum = 0.f;
        for (int k = lane; k < ow; k += 64) sum += expf(o[k] - m);
        res = o[label] - (m + logf(tl_wave_sum(sum)));
    }
    if (lane == 0) lp[i] = res;
}
}  // namespace

size_t wn_teacher_ws_bytes(const wn_handle* h, int B, int F, long long T) { return t_layout(h, B, F, T).total; }

extern "C" size_t wn_teacher_workspace_bytes(const wn_handle* h, int B, int F, int64_t T) {
    if (!h || !h->finalized || h->cfg.kind != WN_KIND_TEACHER || B < 1 || F < 1 || T < 1) return 0;
    return wn_teacher_ws_bytes(h, B, F, T);
}

extern "C" int wn_teacher_forward(wn_handle* h, const float* wav, const float* mel, int B, int F, int64_t T,
                                  float* out_params, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_teacher_forward: null handle");
    if (!h->finalized) return wn_fail(h, WN_ESTATE, "wn_teacher_forward: call wn_finalize first");
    const wn_config& c = h->cfg;
    if (c.kind != WN_KIND_TEACHER) return wn_fail(h, WN_EINVAL, "wn_teacher_forward: handle is not a Wavenet teacher");
    if (B < 1 || F < 1 || T < 1 || !wav || !mel || !out_params || !ws)
        return wn_fail(h, WN_EINVAL, "wn_teacher_forward: bad argument");
    const WnWork work(h);
    const long long TE = (long long)F * h->frame_shift, md = 1ll << (c.num_stages - 1);
    if (T > TE) return wn_fail(h, WN_EINVAL, "wn_teacher_forward: %lld samples need more than %d mel frames "
                               "(wavenet.py:79 assert cond_len >= x_len)", (long long)T, F);
    if (T % md) return wn_fail(h, WN_EINVAL, "wn_teacher_forward: length %lld is not a multiple of the largest "
                               "dilation %lld (masked.py:188)", (long long)T, md);
    if (TE > 2000000) return wn_fail(h, WN_EINVAL, "wn_teacher_forward: utterance too long (32-bit row offsets)");
    const TLayout L = t_layout(h, B, F, T);
    if (ws_bytes < L.total)
        return wn_fail(h, WN_ENOMEM, "wn_teacher_forward: workspace %zu < %zu bytes", ws_bytes, L.total);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(ws);
    float* enc = reinterpret_cast<float*>(base + L.enc);
    unsigned* l = reinterpret_cast<unsigned*>(base + L.l);
    unsigned* m = reinterpret_cast<unsigned*>(base + L.m);
    float* s = reinterpret_cast<float*>(base + L.s);
    float* h1 = reinterpret_cast<float*>(base + L.h1);
    float* xs = reinterpret_cast<float*>(base + L.xs);
    const int W = c.width, S = c.skip_width, H = c.gate_width / 2, Cd = c.deconv_width;
    const TeacherPack& P = h->teacher;

    // conditioning (wavenet.py:214-216), G4 rows of TE columns
    int rc = wn_run_deconv(h, 0, mel, B, F, enc, L.TE, base + L.scratch, st, true);
    if (rc) return rc;
    // l0 = conv_start(shift_right(x_scaled))  (wavenet.py:223-226)
    {
        dim3 g((unsigned)((TG_XP + L.Tp + 255) / 256), B);
        hipLaunchKernelGGL(tg_input_kernel, g, dim3(256), 0, st, wav, xs, (long long)T, L.Tp, c.use_mu_law);
        dim3 g2((unsigned)((L.RS + 255) / 256), W / 8, B);
        hipLaunchKernelGGL(tg_start_kernel, g2, dim3(256), 0, st, xs, h->d_blob + h->ar.start_off, l, W, L.Tp, L.RS);
    }
    auto seg_g4 = [](const unsigned* p, long long bstride, long long rowlen, int col0, int nks, int ng) {
        TgSeg sg;
        sg.base = p; sg.bstride = bstride; sg.rowlen = (int)rowlen; sg.col0 = col0; sg.nks = nks; sg.ng = ng;
        sg.kind = TG_SRC_G4;
        return sg;
    };
    auto seg_acc = [&](const float* p, int nks) {
        TgSeg sg;
        sg.base = reinterpret_cast<const unsigned*>(p); sg.bstride = (long long)S * L.Tp; sg.rowlen = S / 16;
        sg.col0 = 0; sg.nks = nks; sg.ng = 0; sg.kind = TG_SRC_ACC_RELU;
        return sg;
    };
    auto base_args = [&](const TeacherGemmPack& g) {
        TgArgs a{};
        a.wp = reinterpret_cast<const unsigned*>(h->d_blob + g.w_off);
        a.bias = h->d_blob + g.b_off;
        a.inv_scale = g.inv_scale;
        a.nks = g.nks;
        a.T = T;
        return a;
    };
    const TgSeg seg_l = seg_g4(l, (long long)W * L.RS, L.RS, IAF_LP, W / 32, W / 8);
    const TgSeg seg_enc = seg_g4(reinterpret_cast<const unsigned*>(enc), (long long)Cd * L.TE, L.TE, L.c0, Cd / 32, Cd / 8);
    const TgSeg seg_m = seg_g4(m, (long long)H * L.Tp, L.Tp, 0, H / 32, H / 8);
    // s = skip_start(l)  (wavenet.py:231-233)
    {
        TgArgs a = base_args(P.skip_start);
        a.seg[0] = seg_l; a.nseg = 1;
        a.oacc = s; a.oacc_bstride = (long long)S * L.Tp; a.oacc_nmb = S / 16;
        tg_launch<TG_EPI_ACC>(a, P.skip_start.mtiles, B, L.Tp, st);
    }
    for (const TeacherLayerPack& tl : P.layers) {
        {   // d = dilated_conv(l) + mel_cond(enc); m = sigmoid(d[:H]) * tanh(d[H:])  (wavenet.py:243-269)
            TgArgs a = base_args(tl.gate);
            for (int tap = 0; tap < 3; ++tap) {
                a.seg[tap] = seg_l;
                a.seg[tap].col0 = IAF_LP - (2 - tap) * tl.dilation;
            }
            a.seg[3] = seg_enc; a.nseg = 4;
            a.og4 = m; a.og4_bstride = (long long)H * L.Tp; a.og4_rowlen = (int)L.Tp; a.og4_col0 = 0; a.og4_ng = H / 8;
            tg_launch<TG_EPI_GATE>(a, tl.gate.mtiles, B, L.Tp, st);
        }
        {   // l += res(m); s += skip(m)  (wavenet.py:271-277)
            TgArgs a = base_args(tl.rs);
            a.seg[0] = seg_m; a.nseg = 1;
            a.og4 = l; a.og4_bstride = (long long)W * L.RS; a.og4_rowlen = (int)L.RS; a.og4_col0 = IAF_LP; a.og4_ng = W / 8;
            a.oacc = s; a.oacc_bstride = (long long)S * L.Tp; a.oacc_nmb = S / 16;
            a.res_mtiles = W / 64;
            tg_launch<TG_EPI_RS>(a, tl.rs.mtiles, B, L.Tp, st);
        }
    }
    {   // out1(relu(s)) + mel_cond_out1(enc)  (wavenet.py:283-289)
        TgArgs a = base_args(P.out1);
        a.seg[0] = seg_acc(s, S / 32); a.seg[1] = seg_enc; a.nseg = 2;
        a.oacc = h1; a.oacc_bstride = (long long)S * L.Tp; a.oacc_nmb = S / 16;
        tg_launch<TG_EPI_ACC>(a, P.out1.mtiles, B, L.Tp, st);
    }
    {   // out2(relu(.))  (wavenet.py:290-292)
        TgArgs a = base_args(P.out2);
        a.seg[0] = seg_acc(h1, S / 32); a.nseg = 1;
        a.otm = out_params; a.ow = c.out_width;
        tg_launch<TG_EPI_OUT>(a, P.out2.mtiles, B, L.Tp, st);
    }
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}

extern "C" int wn_teacher_log_prob(wn_handle* h, const float* out_params, const float* wav, int B, int64_t T, float* log_prob,
                                   void* stream) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_teacher_log_prob: null handle");
    const wn_config& c = h->cfg;
    if (c.kind != WN_KIND_TEACHER) return wn_fail(h, WN_EINVAL, "wn_teacher_log_prob: handle is not a Wavenet teacher");
    if (B < 1 || T < 1 || !out_params || !wav || !log_prob) return wn_fail(h, WN_EINVAL, "wn_teacher_log_prob: bad argument");
    const int Q = c.use_mu_law ? 256 : 65536;
    if (c.loss_type == WN_LOSS_CE && c.out_width != Q)
        return wn_fail(h, WN_EINVAL, "wn_teacher_log_prob: %d logits for %d classes", c.out_width, Q);
    const long long n = (long long)B * T;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(tg_log_prob_kernel, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, out_params, wav, log_prob,
                       n, c.out_width, c.loss_type, Q, c.use_mu_law);
    WN_HIP(h, hipGetLastError());
    return WN_OK;
}
