// Handle lifecycle, variable table and weight upload of libwnhip.so.
// Replaces graph construction + Saver.restore of the reference
// (wavenet/parallelgen.py:11-41, wavenet/fastgen.py:61-88,118-147).
#include <algorithm>
#include <cmath>
#include <cstdarg>

#include <cstdlib>
#include "wn_internal.h"

// Last error message of the calling thread: concurrent callers of one handle never share (or tear) a string.
static thread_local char t_err[1024] = "";

int wn_fail(const wn_handle*, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}

static void expect(wn_handle* h, const std::string& name, std::vector<int64_t> shape) {
    HostTensor t;
    t.shape = std::move(shape);
    h->vars[name] = t;
}

// conv variable pair (masked.py:190-200) or its weight-norm split (masked.py:145-153)
static void expect_conv(wn_handle* h, const std::string& scope, int K, int cin, int cout) {
    if (h->cfg.use_weight_norm) {
        expect(h, scope + "/W_V", {1, K, cin, cout});
        expect(h, scope + "/W_g", {cout});
    } else {
        expect(h, scope + "/W", {1, K, cin, cout});
    }
    expect(h, scope + "/biases", {cout});
}

// upsampler stack variables: transposed conv (masked.py:249-260) or resize + conv
// (masked.py:294-322, an ordinary conv1d variable pair); names wavenet.py:37-44,54-57
static void expect_deconv(wn_handle* h, const std::string& prefix) {
    const wn_config& c = h->cfg;
    int cin = c.n_mel;
    for (int j = 0; j < c.n_deconv; ++j) {
        std::string scope = (prefix.empty() ? std::string() : prefix + "/") +
                            (c.use_resize_conv ? "resize_conv_" : "trans_conv_") + std::to_string(j + 1);
        if (c.use_resize_conv) {
            expect_conv(h, scope, c.deconv_filter[j], cin, c.deconv_width);
            cin = c.deconv_width;
            continue;
        }
        if (c.use_weight_norm) {
            expect(h, scope + "/kernel_V", {1, c.deconv_filter[j], c.deconv_width, cin});
            expect(h, scope + "/kernel_g", {c.deconv_width});
        } else {
            expect(h, scope + "/kernel", {1, c.deconv_filter[j], c.deconv_width, cin});
        }
        expect(h, scope + "/bias", {c.deconv_width});
        cin = c.deconv_width;
    }
    DeconvStackPack sp;
    sp.prefix = prefix;
    h->stacks.push_back(sp);
}

static int validate(const wn_config& c) {
    if (c.kind != WN_KIND_STUDENT && c.kind != WN_KIND_TEACHER)
        return wn_fail(nullptr, WN_EINVAL, "config: unknown kind %d", c.kind);
    if (c.filter_length != 3)
        return wn_fail(nullptr, WN_EINVAL, "config: filter_length must be 3 (masked.py:349), got %d",
                       c.filter_length);
    if (c.n_deconv < 1 || c.n_deconv > WN_MAX_DECONV)
        return wn_fail(nullptr, WN_EINVAL, "config: deconv_config needs 1..%d layers", WN_MAX_DECONV);
    for (int j = 0; j < c.n_deconv; ++j) {
        int K = c.deconv_filter[j], S = c.deconv_stride[j];
        if (c.use_resize_conv) {
            // nearest-neighbour resize + SAME conv == per-phase GEMM with ceil((K-1)/S)+1 taps
            if (S < 1 || S > 32 || K < 1 || (K - 1 + S - 1) / S + 1 > 7)
                return wn_fail(nullptr, WN_EINVAL, "config: resize_conv layer %d: stride <= 32 and "
                               "(filter-1)/stride <= 6 are the supported range, got filter %d stride %d", j, K, S);
            continue;
        }
        if (S < 1 || K < S || K % S != 0 || ((K - S) & 1))
            return wn_fail(nullptr, WN_EINVAL,
                           "config: deconv layer %d (filter %d, stride %d) unsupported: need "
                           "filter %% stride == 0 and (filter-stride) even", j, K, S);
        if (S > 32 || K / S > 8)
            return wn_fail(nullptr, WN_EINVAL, "config: deconv layer %d: stride <= 32 and filter/stride <= 8 "
                           "are the supported range, got filter %d stride %d", j, K, S);
    }
    if (c.n_mel < 4 || c.n_mel % 4 || c.deconv_width % 64 || c.deconv_width < 64)
        return wn_fail(nullptr, WN_EINVAL, "config: n_mel %% 4 and deconv_width %% 64 must be 0");
    if (c.num_stages < 1 || c.num_stages > 10)
        return wn_fail(nullptr, WN_EINVAL, "config: num_stages must be in 1..10");
    if (c.use_resize_conv != 0 && c.use_resize_conv != 1)
        return wn_fail(nullptr, WN_EINVAL, "config: use_resize_conv must be 0 or 1");
    for (int i = 0; i < 5; ++i)
        if (c.reserved[i]) return wn_fail(nullptr, WN_EINVAL, "config: reserved fields must be 0");
    if (c.upsample_act < 0 || c.upsample_act > 2)
        return wn_fail(nullptr, WN_EINVAL, "config: bad upsample_act");
    if (c.precision != WN_PREC_F16X3 && c.precision != WN_PREC_F32)
        return wn_fail(nullptr, WN_EINVAL, "config: unknown precision mode %d", c.precision);
    if (c.cond_mode < WN_COND_AUTO || c.cond_mode > WN_COND_HOISTED)
        return wn_fail(nullptr, WN_EINVAL, "config: unknown conditioning mode %d (0 default, 1 fused, 2 hoisted)", c.cond_mode);
    if (c.kind == WN_KIND_STUDENT) {
        // width 64 / deconv_width 256 / num_stages >= 7 (every shipped parallel_wavenet*.json) run on the MFMA kernels;
        // any other shape on the generic fp32 kernels of wn_iaf_x.hip (same results, much slower)
        if (c.gate_width != c.width)
            return wn_fail(nullptr, WN_EINVAL, "config: student gate_width must equal width (parallel_wavenet.py:209), "
                           "got width %d gate %d", c.width, c.gate_width);
        if (c.num_stages < 3)
            return wn_fail(nullptr, WN_EINVAL, "config: student num_stages must be >= 3 (the output length is a multiple of "
                           "2^(num_stages-1) and the noise / quantiser kernels move four samples at a time), got %d", c.num_stages);
        // generic kernels (wn_iaf_x.hip): a 64-column tile of `pre` (width rows) and `g` (width / 2 rows) lives in LDS:
        // 1.5 * width * 256 B <= 160 KB
        if (c.width < 2 || (c.width & 1) || c.width > 416 || c.deconv_width > 2048)
            return wn_fail(nullptr, WN_EINVAL, "config: student width must be even and <= 416 (one tile of the generic layer "
                           "kernel must fit the 160 KB of LDS), deconv_width <= 2048; got width %d deconv %d", c.width,
                           c.deconv_width);
        if (c.n_flows < 1 || c.n_flows > WN_MAX_FLOWS)
            return wn_fail(nullptr, WN_EINVAL, "config: num_iaf_layers needs 1..%d flows", WN_MAX_FLOWS);
        if (c.loss_type != WN_LOSS_LOGISTIC && c.loss_type != WN_LOSS_GAUSS)
            return wn_fail(nullptr, WN_EINVAL, "config: student loss_type must be logistic or gauss");
    } else {
        if (c.width % 64 || c.skip_width % 64 || c.gate_width % 128)
            return wn_fail(nullptr, WN_EINVAL, "config: teacher widths must be multiples of 64");
        if (c.gate_width != c.width && c.gate_width != 2 * c.width)
            return wn_fail(nullptr, WN_EINVAL, "config: gate_width must be width or 2*width");
        if (c.loss_type != WN_LOSS_CE && c.loss_type != WN_LOSS_MOL && c.loss_type != WN_LOSS_GAUSS)
            return wn_fail(nullptr, WN_EINVAL, "config: teacher loss_type must be ce, mol or gauss");
        int Q = c.use_mu_law ? 256 : 65536;
        int ow = c.loss_type == WN_LOSS_CE ? Q : c.loss_type == WN_LOSS_MOL ? 3 * c.mol_mix : 2;
        if (c.out_width != ow)
            return wn_fail(nullptr, WN_EINVAL, "config: out_width %d != %d implied by loss_type", c.out_width, ow);
        if (c.num_layers < 1 || c.num_layers > 256)
            return wn_fail(nullptr, WN_EINVAL, "config: bad num_layers");
        // register tiles of the AR step kernels (wn_ar.hip): a wave holds one weight row in chunks of 256 floats -- 8 / 4
        // chunks for the shipped shapes (3 * width + deconv_width <= 2048, gate_width / 2 <= 1024), 16 / 8 in the wide
        // instantiation; the MoL selection scores live in a 64-entry LDS array
        if (3 * c.width + c.deconv_width > 4096 || c.skip_width > 2048 || c.gate_width / 2 > 2048)
            return wn_fail(nullptr, WN_EINVAL,
                           "config: the autoregressive step kernels hold a weight row in registers and need "
                           "3*width + deconv_width <= 4096, skip_width <= 2048, gate_width/2 <= 2048; "
                           "got width %d skip %d gate %d deconv %d", c.width, c.skip_width, c.gate_width, c.deconv_width);
        if (c.loss_type == WN_LOSS_MOL && (c.mol_mix < 1 || c.mol_mix > 64))
            return wn_fail(nullptr, WN_EINVAL, "config: mol_mix must be in 1..64, got %d", c.mol_mix);
    }
    return WN_OK;
}

extern "C" int wn_abi_version(void) { return WN_ABI_VERSION; }

extern "C" int wn_create(const wn_config* cfg, wn_handle** out) {
    if (!cfg || !out) return wn_fail(nullptr, WN_EINVAL, "wn_create: null argument");
    *out = nullptr;
    int rc = validate(*cfg);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return wn_fail(nullptr, WN_EIO, "wn_create: no HIP device (libwnhip has no CPU path)");
    wn_handle* h = new wn_handle();
    h->cfg = *cfg;
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return wn_fail(nullptr, WN_EIO, "wn_create: hipGetDevice failed");
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) {
        h->num_cu = prop.multiProcessorCount;
        h->hoist_limit_bytes = (double)prop.totalGlobalMem / 3.0;   // 96 GB on a 288 GB MI355X
    }
    {
        // ... and never more than half of what is FREE right now: on a GPU another process already fills, the default
        // placement becomes the fused form (no projected-term workspace) instead of an allocation failure later.
        // Resolved ONCE here: wn_workspace_bytes and wn_iaf_generate must agree on the form.
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0)
            h->hoist_limit_bytes = std::min(h->hoist_limit_bytes, (double)free_b * 0.5);
    }
    {   // A/B switches of the layer-group kernel, read ONCE (a handle may be shared by concurrent callers)
        const char* np = getenv("WN_DC_NO_PG");
        h->dc_no_pg = np && atoi(np) != 0;
        const char* ng = getenv("WN_NO_GROUPS");
        const char* fg = getenv("WN_GROUPS");
        if (ng && atoi(ng) != 0) h->groups_env = -1;
        else if (fg && atoi(fg) != 0) h->groups_env = 1;
        h->groups_env0 = h->groups_env;
    }
    if (const char* e = getenv("WN_COND")) {                         // read ONCE: sizing and generate calls must agree
        if (!strcmp(e, "fused")) h->cond_env_mode = WN_COND_FUSED;
        else if (!strcmp(e, "hoisted")) h->cond_env_mode = WN_COND_HOISTED;
    }
    h->frame_shift = 1;
    for (int j = 0; j < cfg->n_deconv; ++j) h->frame_shift *= cfg->deconv_stride[j];
    h->generic_student = cfg->kind == WN_KIND_STUDENT &&
                         (cfg->width != IAF_W || cfg->deconv_width != IAF_CD || cfg->num_stages < 7);

    const wn_config& c = h->cfg;
    if (c.kind == WN_KIND_STUDENT) {
        // parallel_wavenet.py:217-220,311-314
        if (c.share_deconv) expect_deconv(h, "iaf_share");
        for (int k = 0; k < c.n_flows; ++k) {
            std::string p = "iaf_" + std::to_string(k + 1);
            if (!c.share_deconv) expect_deconv(h, p);
            expect_conv(h, p + "/start_conv", 3, 1, c.width);
            for (int i = 0; i < c.iaf_layers[k]; ++i) {
                std::string s = std::to_string(i + 1);
                expect_conv(h, p + "/dilated_conv_" + s, 3, c.width, c.gate_width);
                expect_conv(h, p + "/mel_cond_" + s, 1, c.deconv_width, c.gate_width);
                expect_conv(h, p + "/res_" + s, 1, c.gate_width / 2, c.width);
            }
            expect_conv(h, p + "/out1", 1, c.width, c.width);
            expect_conv(h, p + "/mel_cond_out1", 1, c.deconv_width, c.width);
            expect_conv(h, p + "/out2_mean", 1, c.width, 1);
            expect_conv(h, p + "/out2_scale", 1, c.width, 1);
        }
    } else {
        // wavenet.py:142-155,426-501
        expect_deconv(h, "");
        expect_conv(h, "conv_start", 3, 1, c.width);
        expect_conv(h, "skip_start", 1, c.width, c.skip_width);
        for (int i = 0; i < c.num_layers; ++i) {
            std::string s = std::to_string(i + 1);
            expect_conv(h, "dilated_conv_" + s, 3, c.width, c.gate_width);
            expect_conv(h, "mel_cond_" + s, 1, c.deconv_width, c.gate_width);
            expect_conv(h, "res_" + s, 1, c.gate_width / 2, c.width);
            expect_conv(h, "skip_" + s, 1, c.gate_width / 2, c.skip_width);
        }
        expect_conv(h, "out1", 1, c.skip_width, c.skip_width);
        expect_conv(h, "mel_cond_out1", 1, c.deconv_width, c.skip_width);
        expect_conv(h, "out2", 1, c.skip_width, c.out_width);
    }
    *out = h;
    return WN_OK;
}

extern "C" int wn_set_weight(wn_handle* h, const char* name, const float* data,
                             const int64_t* shape, int rank) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_set_weight: null handle");
    if (h->finalized) return wn_fail(h, WN_ESTATE, "wn_set_weight: handle already finalized");
    if (!name || !data || (rank > 0 && !shape)) return wn_fail(h, WN_EINVAL, "wn_set_weight: null argument");
    auto it = h->vars.find(name);
    if (it == h->vars.end())
        return wn_fail(h, WN_ENOENT, "wn_set_weight: '%s' is not a variable of this model", name);
    HostTensor& t = it->second;
    // Saver(reshape=True) (parallelgen.py:40) lets a checkpoint tensor with the
    // same element count but a different shape restore; mirror that.
    int64_t want = 1, got = 1;
    for (int64_t d : t.shape) want *= d;
    for (int i = 0; i < rank; ++i) got *= shape[i];
    if (want != got) {
        std::string ws, gs;
        for (int64_t d : t.shape) ws += std::to_string(d) + ",";
        for (int i = 0; i < rank; ++i) gs += std::to_string(shape[i]) + ",";
        return wn_fail(h, WN_EINVAL, "wn_set_weight: '%s' expects shape [%s] got [%s]", name, ws.c_str(),
                       gs.c_str());
    }
    t.data.assign(data, data + want);
    t.set = true;
    return WN_OK;
}

extern "C" int wn_finalize(wn_handle* h) {
    if (!h) return wn_fail(nullptr, WN_EINVAL, "wn_finalize: null handle");
    if (h->finalized) return wn_fail(h, WN_ESTATE, "wn_finalize: already finalized");
    std::string missing;
    int nmiss = 0;
    for (auto& kv : h->vars)
        if (!kv.second.set) {
            if (nmiss < 6) missing += kv.first + " ";
            ++nmiss;
        }
    if (nmiss) return wn_fail(h, WN_ENOENT, "wn_finalize: %d variables missing: %s%s", nmiss, missing.c_str(),
                              nmiss > 6 ? "..." : "");
    std::vector<float> blob;
    int rc = wn_pack_deconv(h, blob);
    if (rc) return rc;
    rc = h->cfg.kind != WN_KIND_STUDENT ? wn_pack_ar(h, blob) : h->generic_student ? wn_pack_iaf_x(h, blob) : wn_pack_iaf(h, blob);
    if (rc) return rc;
    if (h->cfg.kind == WN_KIND_STUDENT) {
        if (!h->generic_student) rc = wn_pack_iaf_h(h, blob);
        if (rc) return rc;
    } else {
        rc = wn_pack_teacher(h, blob);
        if (rc) return rc;
    }
    WN_HIP(h, hipSetDevice(h->device));
    h->blob_floats = blob.size();
    WN_HIP(h, hipMalloc((void**)&h->d_blob, blob.size() * sizeof(float)));
    WN_HIP(h, hipMemcpy(h->d_blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    if (h->cfg.kind == WN_KIND_TEACHER) {
        rc = wn_ar_post_upload(h);
        if (rc) return rc;
    }
    // dynamic-LDS limits of the student kernels belong to the handle's device: raised here, once, so that generate calls
    // write nothing into the handle (a finalized handle may be shared by concurrent callers)
    if (h->cfg.kind == WN_KIND_STUDENT) {
        rc = wn_iaf_set_attrs(h);
        if (rc) return rc;
    }
    rc = wn_deconv_set_attrs(h);
    if (rc) return rc;
    // host copies are no longer needed
    for (auto& kv : h->vars) std::vector<float>().swap(kv.second.data);
    h->finalized = true;
    return WN_OK;
}

extern "C" int64_t wn_iaf_length(const wn_handle* h, int F) {
    if (!h || F < 0) return WN_EINVAL;
    int64_t md = 1ll << (h->cfg.num_stages - 1);
    return ((int64_t)F * h->frame_shift / md) * md;   // parallel_wavenet.py:293-302
}

extern "C" int64_t wn_ar_length(const wn_handle* h, int F) {
    if (!h || F < 0) return WN_EINVAL;
    return (int64_t)F * h->frame_shift;               // fastgen.py:136
}

extern "C" size_t wn_workspace_bytes(const wn_handle* h, int B, int F) {
    if (!h || !h->finalized || B < 1 || F < 1) return 0;
    return h->cfg.kind == WN_KIND_STUDENT ? wn_iaf_workspace_bytes(h, B, F) : wn_ar_workspace_bytes(h, B, F);
}

extern "C" size_t wn_iaf_workspace_bytes_form(const wn_handle* h, int form, int B, int F) {
    if (!h || !h->finalized || h->cfg.kind != WN_KIND_STUDENT || B < 1 || F < 1) return 0;
    if (form < WN_FORM_DEFAULT || form > WN_FORM_F16X3_FUSED) return 0;
    return wn_iaf_workspace_bytes(h, B, F, form);
}

extern "C" const char* wn_last_error(const wn_handle*) { return t_err; }

extern "C" void wn_destroy(wn_handle* h) {
    if (!h) return;
    wn_ar_release(h);
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->part_events) (void)hipEventDestroy(e);
    if (h->d_blob) (void)hipFree(h->d_blob);
    delete h;
}

// ---- shared by the pack functions: fetch a (weight-norm folded) kernel ----
// masked.py:131-157: W = V / ||V||_axes * g ; axes (0,1,2) conv, (0,1,3) deconv.
std::vector<float> wn_get_kernel(const wn_handle* h, const std::string& scope, const char* name, bool deconv) {
    if (!h->cfg.use_weight_norm) return h->vars.at(scope + "/" + name).data;
    const HostTensor& V = h->vars.at(scope + "/" + name + "_V");
    const HostTensor& g = h->vars.at(scope + "/" + name + "_g");
    const int64_t K = V.shape[1], d2 = V.shape[2], d3 = V.shape[3];
    std::vector<float> W(V.data.size());
    const int64_t nout = deconv ? d2 : d3;
    std::vector<double> ss(nout, 0.0);
    for (int64_t k = 0; k < K; ++k)
        for (int64_t a = 0; a < d2; ++a)
            for (int64_t b = 0; b < d3; ++b) {
                double v = V.data[(k * d2 + a) * d3 + b];
                ss[deconv ? a : b] += v * v;
            }
    for (int64_t k = 0; k < K; ++k)
        for (int64_t a = 0; a < d2; ++a)
            for (int64_t b = 0; b < d3; ++b) {
                int64_t o = deconv ? a : b;
                // tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))
                float inv = 1.0f / std::sqrt((float)std::max(ss[o], 1e-12));
                W[(k * d2 + a) * d3 + b] = V.data[(k * d2 + a) * d3 + b] * inv * g.data[o];
            }
    return W;
}

// ---- CRC32C (Castagnoli) for the TensorFlow checkpoint reader/writer (tf_bundle.py) ----
// Slicing-by-8 table lookup on the host (wnhip.h: wn_crc32c).
extern "C" uint32_t wn_crc32c(const void* data, size_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xff];
    });
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = crc ^ 0xffffffffu;
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = T[7][v & 0xff] ^ T[6][(v >> 8) & 0xff] ^ T[5][(v >> 16) & 0xff] ^ T[4][(v >> 24) & 0xff] ^
            T[3][(v >> 32) & 0xff] ^ T[2][(v >> 40) & 0xff] ^ T[1][(v >> 48) & 0xff] ^ T[0][(v >> 56) & 0xff];
        p += 8;
        n -= 8;
    }
    while (n--) c = T[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}
