// Generic-width IAF student: the configuration surface beyond the shipped JSONs.
//
// The MFMA kernels of wn_iaf*.hip are specialised for width 64 / deconv_width 256 / num_stages >= 7 -- every
// parallel_wavenet*.json the reference ships.  masked.conv1d itself takes any num_filters (masked.py:160-232) and
// ParallelWavenet any width / deconv_config / num_stages (parallel_wavenet.py:124-141, :200-287), so a student of any
// other shape is served here instead of being refused: plain fp32 FMA kernels on channel-major rows, one workgroup per
// 64-sample tile, weights read as wave-uniform (scalar) operands straight from the TF layouts.  Same semantics, same
// C ABI, same tests against the float64 oracle; an order of magnitude slower than the specialised path (no MFMA, no
// layer fusion beyond gate + residual) -- a correctness path for odd shapes, not a tuned one.
//
//   layer (parallel_wavenet.py:227-254):
//     pre[co] = bd[co] + bc[co] + sum_k sum_ci Wd[k][ci][co] l[ci][t - (2-k) d] + sum_cj Wc[cj][co] enc[cj][t + c0]
//     g[c]    = sigmoid(pre[c]) tanh(pre[c + W/2]),  c < W/2
//     l'[co]  = l[co] + br[co] + sum_c Wr[c][co] g[c]
//   head (:256-277, :105-114, :319-324) and start conv (:222-225) likewise.
#include <algorithm>

#include "wn_internal.h"
#include "wn_codec.h"

namespace {

constexpr int XT = 64;                 // samples per workgroup tile
constexpr int XCH = 16;                // output channels per accumulation chunk
constexpr float X_EXP_M9 = 1.2340980408667956e-4f;
constexpr float X_EXP_7 = 1096.6331584284585f;

__device__ inline float x_softplus_tf(float p) {
    const float thr = -13.942384719848633f;      // tf.nn.softplus: log(eps) + 2
    if (p > -thr) return p;
    if (p < thr) return expf(p);
    return log1pf(expf(p));
}

// l0[c][t] = b[c] + w0[c] x[t-3] + w1[c] x[t-2] + w2[c] x[t-1]   (shift_right folded into the taps)
__global__ void x_start_kernel(const float* __restrict__ x, const float* __restrict__ wb, float* __restrict__ l,
                               int W, int64_t T, int XR, int64_t RS) {
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float* xp = x + (size_t)b * XR + IAF_XP + t;
    const float x0 = xp[-3], x1 = xp[-2], x2 = xp[-1];
    float* lp = l + (size_t)b * W * RS + IAF_LP + t;
    for (int c = 0; c < W; ++c) lp[(size_t)c * RS] = wb[3 * W + c] + wb[c] * x0 + wb[W + c] * x1 + wb[2 * W + c] * x2;
}

// one residual layer on a 64-sample tile.  Thread = (sample tt, channel group cg of 4); a wave shares its channel
// group, so the weights are wave-uniform.  pre and g go through LDS: [W][XT] + [W/2][XT] floats.
__global__ __launch_bounds__(256) void x_layer_kernel(
    const float* __restrict__ lin, float* __restrict__ lout, const float* __restrict__ enc, const float* __restrict__ Wd,
    const float* __restrict__ Wc, const float* __restrict__ Wr, const float* __restrict__ bd, const float* __restrict__ bc,
    const float* __restrict__ br, int W, int Cd, int64_t RS, int64_t TE, int c0, int d, int64_t T) {
    extern __shared__ float xs[];
    float* pre = xs;                       // [W][XT]
    float* g = xs + (size_t)W * XT;        // [W/2][XT]
    const int b = blockIdx.y, tt = threadIdx.x & (XT - 1), cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t t = (int64_t)blockIdx.x * XT + tt;
    const bool on = t < T;
    const float* lb = lin + (size_t)b * W * RS + IAF_LP + (on ? t : 0);
    const float* eb = enc + (size_t)b * Cd * TE + c0 + (on ? t : 0);
    for (int cb = cg * XCH; cb < W; cb += 4 * XCH) {
        float acc[XCH];
#pragma unroll
        for (int o = 0; o < XCH; ++o) acc[o] = cb + o < W ? bd[cb + o] + bc[cb + o] : 0.f;
        for (int k = 0; k < 3; ++k) {
            const float* lk = lb - (2 - k) * d;                    // the 2048 zero columns left of t = 0 make this branch-free
            const float* wk = Wd + (size_t)k * W * W + cb;
            for (int ci = 0; ci < W; ++ci) {
                const float v = lk[(size_t)ci * RS];
#pragma unroll
                for (int o = 0; o < XCH; ++o)
                    if (cb + o < W) acc[o] = fmaf(wk[(size_t)ci * W + o], v, acc[o]);
            }
        }
        for (int cj = 0; cj < Cd; ++cj) {
            const float v = eb[(size_t)cj * TE];
#pragma unroll
            for (int o = 0; o < XCH; ++o)
                if (cb + o < W) acc[o] = fmaf(Wc[(size_t)cj * W + cb + o], v, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < XCH; ++o)
            if (cb + o < W) pre[(size_t)(cb + o) * XT + tt] = acc[o];
    }
    __syncthreads();
    const int H = W / 2;
    for (int c = cg; c < H; c += 4) {
        const float a = pre[(size_t)c * XT + tt], bb = pre[(size_t)(c + H) * XT + tt];
        g[(size_t)c * XT + tt] = (1.f / (1.f + expf(-a))) * tanhf(bb);      // :246-250
    }
    __syncthreads();
    float* ob = lout + (size_t)b * W * RS + IAF_LP + (on ? t : 0);
    for (int cb = cg * XCH; cb < W; cb += 4 * XCH) {
        float acc[XCH];
#pragma unroll
        for (int o = 0; o < XCH; ++o) acc[o] = cb + o < W ? br[cb + o] : 0.f;
        for (int c = 0; c < H; ++c) {
            const float v = g[(size_t)c * XT + tt];
#pragma unroll
            for (int o = 0; o < XCH; ++o)
                if (cb + o < W) acc[o] = fmaf(Wr[(size_t)c * W + cb + o], v, acc[o]);
        }
        if (on)
#pragma unroll
            for (int o = 0; o < XCH; ++o)
                if (cb + o < W) ob[(size_t)(cb + o) * RS] = lb[(size_t)(cb + o) * RS] + acc[o];
    }
}

// flow head on a 64-sample tile: o = relu(bo + bco + Wo relu(l) + Wco enc); mean, p = projections; x <- x s + mean
__global__ __launch_bounds__(256) void x_head_kernel(
    const float* __restrict__ lin, const float* __restrict__ enc, const float* __restrict__ Wo, const float* __restrict__ Wco,
    const float* __restrict__ bo, const float* __restrict__ bco, const float* __restrict__ wm, const float* __restrict__ ws,
    float bmean, float bscale, float* __restrict__ x, float* __restrict__ Mt, float* __restrict__ St, int W, int Cd,
    int64_t RS, int64_t TE, int c0, int XR, int64_t T, int first) {
    extern __shared__ float xs[];
    float* pm = xs;                        // [4][XT] partial mean
    float* ps = xs + 4 * XT;               // [4][XT] partial scale parameter
    const int b = blockIdx.y, tt = threadIdx.x & (XT - 1), cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t t = (int64_t)blockIdx.x * XT + tt;
    const bool on = t < T;
    const float* lb = lin + (size_t)b * W * RS + IAF_LP + (on ? t : 0);
    const float* eb = enc + (size_t)b * Cd * TE + c0 + (on ? t : 0);
    float am = 0.f, as = 0.f;
    for (int cb = cg * XCH; cb < W; cb += 4 * XCH) {
        float acc[XCH];
#pragma unroll
        for (int o = 0; o < XCH; ++o) acc[o] = cb + o < W ? bo[cb + o] + bco[cb + o] : 0.f;
        for (int ci = 0; ci < W; ++ci) {
            const float v = fmaxf(lb[(size_t)ci * RS], 0.f);                     // :256
#pragma unroll
            for (int o = 0; o < XCH; ++o)
                if (cb + o < W) acc[o] = fmaf(Wo[(size_t)ci * W + cb + o], v, acc[o]);
        }
        for (int cj = 0; cj < Cd; ++cj) {
            const float v = eb[(size_t)cj * TE];
#pragma unroll
            for (int o = 0; o < XCH; ++o)
                if (cb + o < W) acc[o] = fmaf(Wco[(size_t)cj * W + cb + o], v, acc[o]);
        }
#pragma unroll
        for (int o = 0; o < XCH; ++o)
            if (cb + o < W) {
                const float r = fmaxf(acc[o], 0.f);
                am = fmaf(wm[cb + o], r, am);
                as = fmaf(ws[cb + o], r, as);
            }
    }
    pm[cg * XT + tt] = am;
    ps[cg * XT + tt] = as;
    __syncthreads();
    if (cg == 0 && on) {
        const float mean = pm[tt] + pm[XT + tt] + pm[2 * XT + tt] + pm[3 * XT + tt] + bmean;
        const float p = ps[tt] + ps[XT + tt] + ps[2 * XT + tt] + ps[3 * XT + tt] + bscale;
        const float sc = fminf(fmaxf(x_softplus_tf(p), X_EXP_M9), X_EXP_7);      // :105-114
        float* xp = x + (size_t)b * XR + IAF_XP + t;
        *xp = *xp * sc + mean;                                                   // :277
        float* mp = Mt + (size_t)b * T + t;
        float* sp = St + (size_t)b * T + t;
        if (first) { *mp = mean; *sp = sc; }
        else { *mp = mean + *mp * sc; *sp = *sp * sc; }                          // :322-323
    }
}

}  // namespace

// Plain copies of the TF tensors (HWIO conv kernels are already [k][ci][co]); offsets in floats into the blob.
int wn_pack_iaf_x(wn_handle* h, std::vector<float>& blob) {
    const wn_config& c = h->cfg;
    auto var = [&](const std::string& nme) -> const std::vector<float>& { return h->vars.at(nme).data; };
    auto put = [&](const std::vector<float>& v) {
        blob.resize(align_up(blob.size(), 64));
        const size_t off = blob.size();
        blob.insert(blob.end(), v.begin(), v.end());
        return off;
    };
    for (int k = 0; k < c.n_flows; ++k) {
        const std::string p = "iaf_" + std::to_string(k + 1);
        IafFlowX fx;
        fx.deconv_stack = c.share_deconv ? 0 : k;
        {
            std::vector<float> wb = wn_get_kernel(h, p + "/start_conv", "W", false);     // [3][1][W]
            const auto& b = var(p + "/start_conv/biases");
            wb.insert(wb.end(), b.begin(), b.end());
            fx.start = put(wb);
        }
        for (int i = 0; i < c.iaf_layers[k]; ++i) {
            const std::string s = std::to_string(i + 1);
            IafLayerX lx;
            lx.dilation = 1 << (i % c.num_stages);                                        // parallel_wavenet.py:228
            lx.wd = put(wn_get_kernel(h, p + "/dilated_conv_" + s, "W", false));
            lx.wc = put(wn_get_kernel(h, p + "/mel_cond_" + s, "W", false));
            lx.wr = put(wn_get_kernel(h, p + "/res_" + s, "W", false));
            lx.bd = put(var(p + "/dilated_conv_" + s + "/biases"));
            lx.bc = put(var(p + "/mel_cond_" + s + "/biases"));
            lx.br = put(var(p + "/res_" + s + "/biases"));
            fx.layers.push_back(lx);
        }
        fx.wo = put(wn_get_kernel(h, p + "/out1", "W", false));
        fx.wco = put(wn_get_kernel(h, p + "/mel_cond_out1", "W", false));
        fx.bo = put(var(p + "/out1/biases"));
        fx.bco = put(var(p + "/mel_cond_out1/biases"));
        fx.wm = put(wn_get_kernel(h, p + "/out2_mean", "W", false));
        fx.ws = put(wn_get_kernel(h, p + "/out2_scale", "W", false));
        fx.bmean = var(p + "/out2_mean/biases")[0];
        fx.bscale = var(p + "/out2_scale/biases")[0];
        h->flows_x.push_back(fx);
    }
    return WN_OK;
}

void wn_iaf_x_start(const wn_handle* h, const IafFlowX& fx, const float* x, float* l, int64_t T, int XR, int64_t RS, int B,
                    hipStream_t st) {
    dim3 g((unsigned)((T + 255) / 256), B);
    hipLaunchKernelGGL(x_start_kernel, g, dim3(256), 0, st, x, h->d_blob + fx.start, l, h->cfg.width, T, XR, RS);
}

void wn_iaf_x_layer(const wn_handle* h, const IafLayerX& lx, const float* lin, float* lout, const float* enc, int64_t RS,
                    int64_t TE, int c0, int B, int64_t T, hipStream_t st) {
    const int W = h->cfg.width, Cd = h->cfg.deconv_width;
    const float* bl = h->d_blob;
    dim3 g((unsigned)((T + XT - 1) / XT), B);
    hipLaunchKernelGGL(x_layer_kernel, g, dim3(256), (size_t)(W + W / 2) * XT * sizeof(float), st, lin, lout, enc,
                       bl + lx.wd, bl + lx.wc, bl + lx.wr, bl + lx.bd, bl + lx.bc, bl + lx.br, W, Cd, RS, TE, c0,
                       lx.dilation, T);
}

void wn_iaf_x_head(const wn_handle* h, const IafFlowX& fx, const float* lin, const float* enc, float* x, float* Mt,
                   float* St, int64_t RS, int64_t TE, int c0, int XR, int64_t T, int first, int B, hipStream_t st) {
    const int W = h->cfg.width, Cd = h->cfg.deconv_width;
    const float* bl = h->d_blob;
    dim3 g((unsigned)((T + XT - 1) / XT), B);
    hipLaunchKernelGGL(x_head_kernel, g, dim3(256), (size_t)8 * XT * sizeof(float), st, lin, enc, bl + fx.wo, bl + fx.wco,
                       bl + fx.bo, bl + fx.bco, bl + fx.wm, bl + fx.ws, fx.bmean, fx.bscale, x, Mt, St, W, Cd, RS, TE, c0,
                       XR, T, first);
}

int wn_iaf_x_set_attrs(wn_handle* h) {
    WN_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(x_layer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (h->cfg.width + h->cfg.width / 2) * XT * (int)sizeof(float)));
    return WN_OK;
}
