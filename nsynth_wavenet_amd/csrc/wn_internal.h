// Internal declarations shared by the translation units of libwnhip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "wnhip.h"

typedef float f4 __attribute__((ext_vector_type(4)));
// 16-byte vector that is only 4-byte aligned: dilated taps / centre crops shift
// time-contiguous rows by an arbitrary number of samples.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool set = false;
};

// ---- device-side parameter blocks (offsets in floats into wn_handle::d_blob) ----

// One transposed-conv layer packed for the MFMA kernel (wn_deconv.hip).
struct DeconvLayerPack {
    int cin, cout, K, S, pL, taps;     // taps = K / S
    size_t w_off;                      // packed A fragments [S][ks4][mb][64][4]
    size_t w_off_h;                    // split-fp16 A fragments (0 = not available for this shape)
    float inv_scale_h;
    size_t b_off;                      // bias [cout]
    // phase-group pack of deconv_pg_kernel (0 = not available): per group of four output phases, A fragments of
    // (4 phases x 32 channels) row groups over (input block, tap); see wn_deconv.hip
    size_t w_off_pg = 0;
    int pg_n = 0, pg_nrg = 0;                                         // phase groups, row groups (512 KB of fragments each)
    int pg_p0[8] = {0}, pg_nph[8] = {0}, pg_d[8] = {0}, pg_rg0[8] = {0};   // first phase, phases (4 | 2), input column offset, first row group
    // fp32 frame-axis GEMM (gemm_f32_kernel<4, true>, wn_iaf_f.hip): uint32 table of tab_f_R = S * cout / 64 pairs
    // {float offset of the row block's first K-group inside w_off's pack, (phase << 8) | input column shift d}; 0 = not available
    size_t tab_f_off = 0;
    int tab_f_R = 0;
};

struct DeconvStackPack {
    std::string prefix;                // "", "iaf_share", "iaf_k"
    std::vector<DeconvLayerPack> layers;
};

// launch group of wn_iaf_g.hip
struct WnGroup {
    int kind = 0;          // 0 natural time base, 1 decimated by 32
    int begin = 0, end = 0;   // layers [begin, end) of the flow
    int halo_cols = 0;     // 2 * sum of the (local) dilations
    int n() const { return end - begin; }
};

// Student flow (wn_iaf.hip)
struct IafLayerPack {
    size_t off;                        // fp32 path: LAYER_FLOATS floats: P | PR | bgate | bres
    size_t off_h;                      // split-fp16 path: IAF_LAYER_H_WORDS words (wn_iaf_h.hip)
    int dilation;
    unsigned id = 0;                   // 1-based number of the layer in the student (marker value, wn_iaf_g.hip)
};
struct IafFlowPack {
    size_t start_off;                  // w[3][W] | b[W]
    std::vector<IafLayerPack> layers;
    std::vector<WnGroup> groups;       // launch plan of wn_iaf_g.hip (empty: per-layer launches only)
    size_t head_off;                   // HEAD_FLOATS floats
    size_t head_off_h;                 // split-fp16 head pack
    int deconv_stack;                  // index into wn_handle::stacks
    int rb_base;                       // first row block of this flow in the hoisted-conditioning table
};

// Student of a shape the MFMA kernels are not specialised for (wn_iaf_x.hip): plain fp32 tensors in TF layout
struct IafLayerX {
    size_t wd, wc, wr, bd, bc, br;     // [3][W][W], [Cd][W], [W/2][W], biases
    int dilation;
};
struct IafFlowX {
    size_t start;                      // w[3][W] | b[W]
    std::vector<IafLayerX> layers;
    size_t wo, wco, bo, bco, wm, ws;   // out1 [W][W], mel_cond_out1 [Cd][W], biases, out2_mean / out2_scale [W]
    float bmean, bscale;
    int deconv_stack;
};

// Teacher (wn_ar.hip): plain [out][in] row-major matrices
struct ArLayerPack {
    size_t wd_off;    // [gate][3*width + deconv_width]   (taps t-2d, t-d, t, cond)
    size_t bd_off;    // [gate]  (dilated bias + cond bias)
    size_t bc_off = 0;   // [gate]  cond bias alone (wn_ar_cond_vars)
    size_t wrs_off;   // [width + skip][gate/2]            (res rows then skip rows)
    size_t brs_off;   // [width + skip]
    size_t wd_b_off, wrs_b_off;   // the same matrices in MFMA A-fragment order (batched step)
    size_t brs_gate_off;          // [res | skip | gate] biases for the batched res/skip kernel
    // merged GEMV step: d_j = wd_j.[ring(t-2d) | ring(t-d) | lin_{j-1} | enc] + wcomp_j.m_{j-1} + bm_j with
    // wcomp_j = Wd_j[tap t] . Wres_{j-1} (computed on the device after the upload), bm_j = bd_j + Wd_j[tap t].bres_{j-1}
    size_t wcomp_off = 0, bm_off = 0;
    size_t wdc_b_off = 0;         // A-fragment order of [wd_j | wcomp_j] for the batched merged step (built on the device)
    int dilation;
    size_t ring_off;  // float offset of this layer's ring inside the state (per batch elem)
};
struct ArPack {
    size_t start_off;     // w[3][width] | b[width]
    size_t wss_off, bss_off;   // skip_start [skip][width], [skip]
    std::vector<ArLayerPack> layers;
    size_t wo1_off, bo1_off;   // [skip][skip + deconv_width], [skip] (out1 | mel_cond_out1)
    size_t bco1_off = 0;       // [skip] mel_cond_out1 bias alone (wn_ar_cond_vars)
    size_t wo2_off, bo2_off;   // [out_width][skip], [out_width]
    size_t wss_b_off, wo1_b_off, wo2_b_off;   // A-fragment order copies
    size_t ring_floats;        // per batch element
};

// Full-sequence teacher forward (wn_teacher.hip): split-fp16 A fragments in 64-row tiles
struct TeacherGemmPack {
    size_t w_off = 0;      // [m-tile][K-step][4 row blocks][plane][lane][4] words
    size_t b_off = 0;      // [m-tile][64] bias, tile-local row order
    float inv_scale = 1.f;
    int nks = 0, mtiles = 0;
};
struct TeacherLayerPack {
    TeacherGemmPack gate, rs;
    int dilation;
};
struct TeacherPack {
    TeacherGemmPack skip_start, out1, out2;
    std::vector<TeacherLayerPack> layers;
};

struct wn_handle {
    wn_config cfg;
    std::map<std::string, HostTensor> vars;   // expected variables (+ data once set)
    bool finalized = false;

    int device = 0;
    float* d_blob = nullptr;
    size_t blob_floats = 0;
    std::vector<DeconvStackPack> stacks;
    std::vector<IafFlowPack> flows;
    std::vector<IafFlowX> flows_x;            // generic-width student (wn_iaf_x.hip) instead of `flows`
    bool generic_student = false;             // width / deconv_width / num_stages outside the MFMA kernels' shape
    ArPack ar;
    TeacherPack teacher;
    // hoisted conditioning (wn_iaf_c.hip): word offsets of the 8-K-step cond fragment arrays of
    // every layer and head ("row block"), flow after flow, stored as uint32 inside the blob
    size_t cond_tab_off = 0;
    int cond_rows = 0;
    // the same table for the fp32 form (wn_iaf_f.hip): {float offset of the row block's 16 cond K-groups, float offset of its
    // 64 biases in lane order} per row block, flow after flow (layers, then the head)
    size_t cond_tab_f_off = 0;
    // row-block orders of the conditioning GEMM (uint32 tables inside the blob, each cond_rows long): identity; all
    // flows' natural row blocks first, then the decimated ones (shared deconv stack, wn_iaf_g.hip's plan); the same
    // per flow with flow-local indices (private stacks).  n_nat: natural row blocks of the whole student / per flow
    size_t order_id_off = 0, order_all_off = 0, order_flow_off = 0;
    int n_nat_all = 0;
    std::vector<int> n_nat_flow;
    bool groups_ok = false;                   // every flow has a group plan
    int frame_shift = 1;
    int num_cu = 256;
    // resolved once in wn_create: WN_COND override of cond_mode 0 and the workspace limit of the hoisted form
    int cond_env_mode = 0;                    // WN_COND_AUTO or the form named by the environment
    bool dc_no_pg = false;                    // WN_DC_NO_PG=1 at wn_create: upsampler without the phase-group kernel (wn_deconv.hip)
    int groups_env = 0;                       // launch structure in force: +1 groups always, -1 never, 0 size policy (wn_iaf_set_groups)
    int groups_env0 = 0;                      // ... as wn_create resolved it (WN_GROUPS=1 / WN_NO_GROUPS=1): what mode 0 restores
    double hoist_limit_bytes = 96e9;          // a third of the device memory
    // Work calls (generate / deconv / AR / teacher) hold `sw` SHARED while they are inside the library and read every switch
    // once at their entry; the switches (wn_iaf_set_groups, wn_ar_set_graph, wn_profile_*) try to take it EXCLUSIVELY and
    // return WN_ESTATE when a work call of another thread is in flight (wnhip.h, "Concurrency").
    mutable std::shared_mutex sw;
    std::mutex list_mu;                       // the event lists of the measurement modes and the AR graph list below
    // hipGraphs of the wn_ar_generate calls in flight (wn_ar.hip): one entry per call, retired by later calls once its
    // stream has run it, all released in wn_destroy
    std::vector<void*> ar_graphs;
    void* ar_cap_stream = nullptr;     // private non-blocking stream wn_ar_generate CAPTURES on (wn_ar.hip; under its capture mutex)
    bool ar_use_graph = true;                 // wn_ar_set_graph
    // bench.py measurement aid (wn_profile_begin/end)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_events;     // begin/end pairs
    int64_t prof_launches = 0;
    // second mode (wn_profile_parts_begin/end): one event at every PART boundary of a generate call instead of the
    // brackets around the residual-stack launches; part_tags[i] = part that starts at part_events[i] (-1: call ends)
    int parts_mask = 0xf;                     // wn_profile_parts_only: parts a generate call runs (measurement only: bit = part tag)
    bool parts_on = false;
    std::vector<hipEvent_t> part_events;
    std::vector<int> part_tags;
    int64_t part_calls = 0;
};

// ---- error helpers ----
int wn_fail(const wn_handle* h, int code, const char* fmt, ...);   // message -> the calling thread's wn_last_error
// a work call is inside the library (shared) / a switch changes the handle (exclusive, refused while busy)
struct WnWork {
    std::shared_lock<std::shared_mutex> lk;
    explicit WnWork(const wn_handle* h) : lk(h->sw) {}
};
#define WN_SWITCH(h, fn)                                                                         \
    std::unique_lock<std::shared_mutex> sw_lk__((h)->sw, std::try_to_lock);                      \
    if (!sw_lk__.owns_lock())                                                                    \
        return wn_fail((h), WN_ESTATE, fn ": a work call of another thread is in flight on this handle (switches are "  \
                                          "refused while the handle is busy)")
#define WN_HIP(h, expr)                                                              \
    do {                                                                             \
        hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess)                                                       \
            return wn_fail((h), WN_EIO, "%s failed: %s (%s:%d)", #expr,              \
                           hipGetErrorString(e__), __FILE__, __LINE__);              \
    } while (0)

// ---- sizes of the IAF packs (width 64, deconv_width 256) ----
// layer: P 112 K-steps * 4 mb * 64 lanes | PR 8 * 4 * 64 | bgate 64 | bres 64
constexpr int IAF_W = 64;
constexpr int IAF_CD = 256;
constexpr int IAF_P_FLOATS = 112 * 4 * 64;       // 28672
constexpr int IAF_PR_FLOATS = 8 * 4 * 64;        // 2048
constexpr int IAF_LAYER_FLOATS = IAF_P_FLOATS + IAF_PR_FLOATS + 64 + 64;   // 30848
// head: PH 80 K-steps * 4 * 64 | bias 64 | wmean 64 | wscale 64 | bmean, bscale, pad
constexpr int IAF_PH_FLOATS = 80 * 4 * 64;       // 20480
constexpr int IAF_HEAD_FLOATS = IAF_PH_FLOATS + 64 * 3 + 4;                 // 20676

constexpr int IAF_LAYER_H_WORDS = IAF_P_FLOATS + IAF_PR_FLOATS + 128 + 4 + 4;   // + 1/scale_main, 1/scale_res, pad; + marker quad
                                                                            // (the layer's id x 4: wn_iaf_g.hip's DMA-delivered 'slot ready' word)
// precision of the IAF contractions (wn_config.precision)
constexpr int WN_PREC_F16X3 = 0;   // split-fp16 on the fp16 MFMA (default)
constexpr int WN_PREC_F32 = 1;     // fp32 MFMA
// where the f16x3 path evaluates the per-layer conditioning 1x1s (wn_config.cond_mode)
constexpr int WN_COND_AUTO = 0;    // hoisted once enc + l outgrow the 256 MB Infinity Cache, else fused
constexpr int WN_COND_FUSED = 1;   // inside every layer kernel (re-reads enc per layer)
constexpr int WN_COND_HOISTED = 2; // one GEMM per deconv stack writes them for all layers

constexpr int IAF_LP = 2048;   // zero left pad of activation rows (>= 2 * max dilation; >= 32 * 64: the DL layout of
                               // wn_iaf_g.hip keeps 64 zero columns in front of each of its 32 residue rows)
constexpr int IAF_XP = 64;     // zero left pad of the flow input x (>= filter_length)
// First bytes of every caller-owned workspace: the range-guard words of the generate calls made on it (status[0] of the
// current call, status[1] accumulated since wn_iaf_range_reset).  Nothing else -- wn_deconv included -- writes there.
constexpr size_t WN_WS_HEAD = 256;

// ---- implemented in the .hip units ----
int wn_pack_deconv(wn_handle* h, std::vector<float>& blob);
int wn_pack_iaf(wn_handle* h, std::vector<float>& blob);
int wn_pack_ar(wn_handle* h, std::vector<float>& blob);
int wn_pack_teacher(wn_handle* h, std::vector<float>& blob);   // after wn_pack_ar (reuses its matrices)

// Runs stack `si` on mel [B,F,n_mel]; writes channel-major enc [B][cout][enc_stride]
// (first valid sample at column 0).  Scratch carved from ws.
struct DeconvScratch {
    size_t bytes;
};
size_t wn_deconv_scratch_bytes(const wn_handle* h, int B, int F);
// split_out: write the LAST layer's output as split-fp16 words in the G4 layout (wn_iaf_h.hip)
// instead of fp32 rows
// status: the call's range-guard word (wn_codec.h), may be null; prec: WN_PREC_* of this call, -1 = the handle's
int wn_run_deconv(wn_handle* h, int si, const float* mel, int B, int F,
                  float* enc_cm, int64_t enc_stride, void* scratch, hipStream_t st, bool split_out = false,
                  unsigned* status = nullptr, int prec = -1);

int wn_pack_iaf_h(wn_handle* h, std::vector<float>& blob);
// ---- generic-width student (wn_iaf_x.hip) ----
int wn_pack_iaf_x(wn_handle* h, std::vector<float>& blob);
int wn_deconv_set_attrs(wn_handle* h);
int wn_iaf_x_set_attrs(wn_handle* h);
void wn_iaf_x_start(const wn_handle* h, const IafFlowX& fx, const float* x, float* l, int64_t T, int XR, int64_t RS, int B,
                    hipStream_t st);
void wn_iaf_x_layer(const wn_handle* h, const IafLayerX& lx, const float* lin, float* lout, const float* enc, int64_t RS,
                    int64_t TE, int c0, int B, int64_t T, hipStream_t st);
void wn_iaf_x_head(const wn_handle* h, const IafFlowX& fx, const float* lin, const float* enc, float* x, float* Mt,
                   float* St, int64_t RS, int64_t TE, int c0, int XR, int64_t T, int first, int B, hipStream_t st);
int wn_iaf_h_set_attrs(wn_handle* h);
void wn_iaf_h_start(const float* x, const float* wb, float* l, int64_t T, int XR, int64_t RS, int B, hipStream_t st,
                    unsigned* status);
void wn_iaf_h_layer(const float* lin, float* lout, const float* enc, const float* wpack, int64_t RS, int64_t TE,
                    int c0, int d, int B, int64_t T, int num_cu, hipStream_t st, unsigned* status,
                    const float* x = nullptr, int XR = 0, const float* wstart = nullptr);
void wn_iaf_h_head(const float* lin, const float* enc, const float* wpack, float* x, float* Mt, float* St,
                   int64_t RS, int64_t TE, int c0, int XR, int64_t T, int first, int B, int num_cu, hipStream_t st);
bool wn_iaf_hoisted(const wn_handle* h, int B, int64_t T);
int wn_iaf_c_set_attrs(wn_handle* h);
size_t wn_iaf_c_floats(int R, int64_t T);
void wn_iaf_c_cond(const float* enc, const float* wblob, const unsigned* rb_off, const unsigned* order, int n_nat,
                   float* C, int64_t c_bstride, int64_t TE, int c0, int R, int B, int64_t T, int num_cu, hipStream_t st);
void wn_iaf_c_layer(const float* lin, float* lout, const float* C, int64_t c_bstride, const float* wpack, int64_t RS,
                    int d, int B, int64_t T, int num_cu, hipStream_t st, unsigned* status);
bool wn_iaf_c_last_ok();
void wn_iaf_c_layer_head(const float* lin, const float* C, const float* Ch, int64_t c_bstride, const float* wpack,
                         const float* wpack_head, float* x, float* Mt, float* St, int64_t RS, int XR, int d, int first,
                         int B, int64_t T, int num_cu, hipStream_t st, unsigned* status);
bool wn_iaf_c_pair_ok(int da, int db);
// ---- layer groups resident in LDS (wn_iaf_g.hip) ----
bool wn_iaf_g_plan(const std::vector<int>& dilations, std::vector<WnGroup>& out);
int wn_iaf_g_set_attrs(wn_handle* h);
int wn_iaf_set_attrs(wn_handle* h);   // every student kernel's dynamic-LDS limit, once per handle (wn_finalize)
void wn_iaf_g_run(const wn_handle* h, const WnGroup& g, const IafLayerPack* layers, const float* Cg, size_t rb_floats,
                  int64_t c_bstride, const float* lin, float* lout, int64_t RS, int out_dec, int B, int64_t T,
                  const float* x, int XR, const float* wstart, bool last, const float* whead, const float* xin,
                  float* xout, float* Mt, float* St, int first_flow, unsigned* status, hipStream_t st);
void wn_iaf_c_pair(const float* lin, float* lout, const float* CA, const float* CB, int64_t c_bstride, const float* wA,
                   const float* wB, int64_t RS, int da, int db, int B, int64_t T, int num_cu, hipStream_t st,
                   const float* x, int XR, const float* wstart, unsigned* status);
void wn_iaf_c_head(const float* lin, const float* C, int64_t c_bstride, const float* wpack, float* x, float* Mt,
                   float* St, int64_t RS, int XR, int64_t T, int first, int B, int num_cu, hipStream_t st);
// ---- fp32 GEMMs with the activation tile in LDS (wn_iaf_f.hip) ----
int wn_iaf_f_set_attrs(wn_handle* h);
bool wn_iaf_f_cond_ok(int64_t T, int c0);
void wn_iaf_f_cond(const wn_handle* h, const float* enc, const unsigned* tab, int R, float* C, int64_t c_bstride, int64_t TE,
                   int c0, int B, int64_t T, hipStream_t st);
void wn_deconv_f_gemm(const wn_handle* h, const float* x, int xs, int xoff, const unsigned* tab, int R, int taps, int a_kstride,
                      float* yp, int S, int cout, int L, int Lp, int B, hipStream_t st);
int wn_iaf_form(const wn_handle* h, int B, int64_t T, int form);   // WN_COND_FUSED / _HOISTED for this call (form: WN_FORM_*)
bool wn_iaf_use_groups(const wn_handle* h, int B, int64_t T, int cond_form);   // layer-group kernel for this call?
int wn_form_precision(const wn_handle* h, int form);              // WN_PREC_* a call of this form computes in
std::vector<float> wn_get_kernel(const wn_handle* h, const std::string& scope, const char* name, bool deconv);
size_t wn_iaf_workspace_bytes(const wn_handle* h, int B, int F, int form = WN_FORM_DEFAULT);
size_t wn_ar_workspace_bytes(const wn_handle* h, int B, int F);
void wn_ar_release(wn_handle* h);
int wn_ar_post_upload(wn_handle* h);   // device-side part of the AR packing (composite matrices)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
