// Measurement-only variant of pair_epilogue (csrc/wn_iaf_c.h) used by epilogue_cost.hip and team_probe.hip; it did not pay
// inside the kernels and is not part of the product.
#pragma once
#include "wn_iaf_c.h"

namespace {

// The same for NE blocks of one wave in lock step: every stage (gate, split, residual 1x1, join + skip + split) runs
// over all blocks before the next begins, so the NE dependency chains exp -> rcp -> split -> MFMA -> join -> split are
// interleaved in the instruction stream (hipcc keeps two calls of pair_epilogue in source order: one chain after the
// other), and a residual fragment read from LDS serves NE products.
template <int NE>
__device__ inline void pair_epilogue_n(const PairLayer& w, const f4 (&acc)[NE][4], const wn_u4 (&lh)[NE][2],
                                       const wn_u4 (&ll)[NE][2], wn_u4 (&oh)[NE][2], wn_u4 (&ol)[NE][2], float& amax) {
    float g[NE][2][4];
#pragma unroll
    for (int mg = 0; mg < 2; ++mg)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float bs = w.bg[mg * 4 + r], bt = w.bg[(mg + 2) * 4 + r];
#pragma unroll
            for (int e = 0; e < NE; ++e)
                g[e][mg][r] = gate_scaled(fmaf(acc[e][mg][r], -WN_LOG2E * w.inv_m, bs),
                                          fmaf(acc[e][mg + 2][r], 2.f * WN_LOG2E * w.inv_m, bt));
        }
    wn_u4 gh[NE], gl[NE];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            unsigned hw, lw;
            wn_split_pair(g[e][i >> 1][(i & 1) * 2], g[e][i >> 1][(i & 1) * 2 + 1], hw, lw);
            gh[e][i] = hw;
            gl[e][i] = lw;
        }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const wn_u4 ph = w.PRl[(mb * 2 + 0) * 64], pl = w.PRl[(mb * 2 + 1) * 64];
        f4 rc[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) rc[e] = mfma3(ph, pl, gh[e], gl[e], (f4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            const float b0 = w.br[mb * 4 + 2 * rp], b1 = w.br[mb * 4 + 2 * rp + 1];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                float l0, l1;
                wn_join_pair(lh[e][mb >> 1][(mb & 1) * 2 + rp], ll[e][mb >> 1][(mb & 1) * 2 + rp], l0, l1);
                const float v0 = l0 + fmaf(rc[e][2 * rp], w.inv_r, b0);
                const float v1 = l1 + fmaf(rc[e][2 * rp + 1], w.inv_r, b1);
                unsigned hw, lw;
                wn_split_pair_t(v0, v1, hw, lw, amax);
                oh[e][mb >> 1][(mb & 1) * 2 + rp] = hw;
                ol[e][mb >> 1][(mb & 1) * 2 + rp] = lw;
            }
        }
    }
}

}  // namespace
