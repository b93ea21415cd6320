/*
 * wnhip.h — C ABI of libwnhip.so, the MI355X (gfx950) generation engine that
 * replaces the TensorFlow graph execution on the generation path of
 * bfs18/nsynth_wavenet.
 *
 * The reference exposes NO native/FFI interface for this path (it is pure
 * Python over TensorFlow 1.x).  Its seam is Python:
 *     wavenet/parallelgen.py:22   synthesis(hparams, mel, save_paths, checkpoint_path)
 *     wavenet/fastgen.py:69       encode(hparams, wav_data, checkpoint_path)
 *     wavenet/fastgen.py:128      synthesis(hparams, mel_encoding, save_paths, checkpoint_path)
 * and inside those, one `sess.run` on the graphs built by
 *     wavenet/parallel_wavenet.py:289-359   ParallelWavenet.feed_forward + _clip_quant_scale
 *     wavenet/wavenet.py:142-155            Wavenet.deconv_stack
 *     wavenet/wavenet.py:379-514            Fastgen.sample
 * Each entry point below states which of those `sess.run` calls it replaces.
 * The ctypes binding a maintainer of the reference would add is shown in
 * INTEGRATION.md.
 *
 * Conventions
 *   - Plain C, no torch / HIP types in signatures: a stream is passed as the
 *     raw hipStream_t value in a `void*` (NULL = the null stream).
 *   - Every pointer is a DEVICE pointer unless its name ends in `_host`.
 *   - The caller owns every input, output and workspace buffer; the library
 *     owns only the handle and its packed weights.  No allocation and no host
 *     synchronisation happens inside the generate calls (wn_ar_generate, whose
 *     sample loop is driven from the host, is the documented exception: it
 *     enqueues work and returns without synchronising, but it instantiates a
 *     hipGraph).
 *   - Return value: 0 on success, a negative errno-style code otherwise
 *     (WN_EINVAL shape/config, WN_ENOENT unknown/missing weight, WN_ENOMEM
 *     workspace too small, WN_EIO HIP runtime error, WN_ESTATE wrong call
 *     order).  Nothing throws or aborts across the ABI; the message is
 *     available from wn_last_error().
 *   - Concurrency (SURVEY 8(b) "Threading / streams"; tests/test_gpu_threads.py drives one handle from two
 *     host threads on two streams).  After wn_finalize() the weights and every table of a handle are read-only.
 *     The WORK calls -- wn_deconv, wn_iaf_generate*, wn_iaf_range_*, wn_clip_quant, wn_ar_reset / wn_ar_step /
 *     wn_ar_generate / wn_ar_cond_vars, wn_teacher_forward / wn_teacher_log_prob -- keep all per-call state (the
 *     range-guard word, the autoregressive queues and step counter) in the caller's workspace / state buffer:
 *     any number of host threads may issue them on ONE handle at the same time, each with its own workspace
 *     and stream.  (wn_ar_generate's hipGraphs are per call; the handle only keeps them alive until their
 *     stream has run them.)
 *     The SWITCHES -- wn_iaf_set_groups, wn_ar_set_graph, wn_profile_begin / _pause / _end,
 *     wn_profile_parts_begin / _only / _end -- change how later work calls of the handle run.  They take the
 *     handle exclusively: while a work call of another thread is inside the library they return WN_ESTATE
 *     ("busy") and change nothing; a work call reads every switch once, at its entry.  The two measurement
 *     modes (wn_profile_*) are for single-caller benchmarking: with several callers the recorded events
 *     interleave (the lists themselves are locked).
 *     wn_last_error() returns the message of the CALLING THREAD's last failed call.
 */
#ifndef WNHIP_H_
#define WNHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_ABI_VERSION 1

/* libwnhip.so is linked with -fvisibility=hidden: the functions declared here are its whole dynamic symbol table
 * (tests/test_host.py checks `nm -D --defined-only` against this header, both ways). */
#define WN_API __attribute__((visibility("default")))

#define WN_OK       0
#define WN_EINVAL  (-22)
#define WN_ENOENT  (-2)
#define WN_ENOMEM  (-12)
#define WN_EIO     (-5)
#define WN_ESTATE  (-1)
#define WN_ERANGE  (-34)   /* wn_iaf_range_status: an activation left the fp16 range of the split-fp16 arithmetic */

/* `form` of wn_iaf_generate_form / wn_iaf_workspace_bytes_form: which arithmetic ONE call runs in */
#define WN_FORM_DEFAULT     (-1)  /* what the handle was created for (wn_config.precision / cond_mode) */
#define WN_FORM_F16X3        0    /* split-fp16 MFMA, conditioning placed by the handle's cond_mode policy */
#define WN_FORM_F32          1    /* fp32 MFMA (the reference's own arithmetic): no fp16 range limit, ~2.3x slower */
#define WN_FORM_F16X3_FUSED  2    /* split-fp16 MFMA without the hoisted-conditioning workspace */

#define WN_MAX_DECONV 4
#define WN_MAX_FLOWS  8

/* kind */
#define WN_KIND_STUDENT 0   /* ParallelWavenet (IAF), parallel_wavenet.py */
#define WN_KIND_TEACHER 1   /* Wavenet / Fastgen (autoregressive), wavenet.py */
/* loss_type */
#define WN_LOSS_CE       0
#define WN_LOSS_MOL      1
#define WN_LOSS_GAUSS    2
#define WN_LOSS_LOGISTIC 3  /* student only: noise source (parallel_wavenet.py:303-306) */
/* upsample_act (masked.py:28-36) */
#define WN_ACT_TANH       0
#define WN_ACT_RELU       1
#define WN_ACT_LEAKY_RELU 2 /* alpha = 0.4 */

/* Plain-C mirror of the config JSON (config_jsons/NAME.json) after the
 * reference's per-class getattr defaults have been applied by the caller
 * (wavenet.py:105-129,326-345; parallel_wavenet.py:124-141). */
typedef struct wn_config {
    int32_t kind;
    int32_t n_mel;                          /* 80 (mel_extractor.py:17) */
    int32_t width;                          /* residual channels */
    int32_t skip_width;                     /* teacher only */
    int32_t gate_width;                     /* student: width; teacher: width or 2*width */
    int32_t deconv_width;
    int32_t n_deconv;                       /* len(deconv_config) */
    int32_t deconv_filter[WN_MAX_DECONV];
    int32_t deconv_stride[WN_MAX_DECONV];
    int32_t filter_length;                  /* 3 (masked.py:349 asserts it) */
    int32_t num_stages;
    int32_t num_layers;                     /* teacher */
    int32_t n_flows;                        /* student: len(num_iaf_layers) */
    int32_t iaf_layers[WN_MAX_FLOWS];
    int32_t use_mu_law;
    int32_t loss_type;
    int32_t mol_mix;
    int32_t out_width;                      /* teacher: 256/65536 (ce), 3*mol_mix, 2; student: 2 */
    int32_t share_deconv;                   /* use_share_deconv || use_teacher_deconv */
    int32_t use_weight_norm;
    int32_t upsample_act;
    int32_t precision;                      /* IAF / upsampler contractions: 0 split-fp16 x3 on the
                                               fp16 MFMA (default), 1 fp32 MFMA */
    int32_t cond_mode;                      /* where the split-fp16 path evaluates the per-layer
                                               conditioning 1x1s: 0 default (= 2 unless the
                                               environment says WN_COND=fused or the projected term
                                               of the call would exceed a third of the device memory
                                               / half of what was free at wn_create), 1 inside every
                                               layer kernel, 2 one GEMM per deconv stack */
    int32_t use_resize_conv;                /* upsampler = nearest-neighbour resize + SAME conv
                                               (masked.py:294-322) instead of transposed conv;
                                               variables <prefix>resize_conv_i/{W,biases} */
    int32_t reserved[5];                    /* must be 0 */
} wn_config;

typedef struct wn_handle wn_handle;

WN_API int wn_abi_version(void);

/* Build an empty engine for one model on the current HIP device.
 * Replaces graph construction (parallelgen.py:11-19, fastgen.py:61-66,118-125).
 * Students of the shipped shape (width 64, deconv_width 256, num_stages >= 7) run on the MFMA kernels; any other even
 * width <= 416 (one tile of the generic layer kernel must fit the 160 KB of LDS) / deconv_width % 64 == 0 /
 * num_stages >= 3 on generic fp32 kernels (same results, much slower).
 * Teachers: 3 * width + deconv_width <= 2048 runs on the tuned step kernels (every shipped wavenet_*.json), up to 4096 on
 * a wide instantiation of the same kernels (a correctness path: below four utterances, and at every batch size when
 * 3*width + deconv_width + gate_width/2 > 3072, one utterance at a time; otherwise the batched step); beyond that wn_create refuses. */
WN_API int wn_create(const wn_config* cfg_host, wn_handle** out);

/* Provide one variable under its TensorFlow name WITHOUT the
 * '/ExponentialMovingAverage' suffix and ':0' (fastgen.py:12-14), in the
 * reference's shape: conv kernels HWIO [1,K,Cin,Cout] '<scope>/W', biases
 * '<scope>/biases' (masked.py:190-200); transposed-conv kernels [1,K,Cout,Cin]
 * '<scope>/kernel', '<scope>/bias' (masked.py:249-260); weight-norm pairs
 * '<name>_V','<name>_g' (masked.py:145-153).  Replaces Saver.restore
 * (parallelgen.py:29-41, fastgen.py:80-84,142-147).  Unknown names -> WN_ENOENT. */
WN_API int wn_set_weight(wn_handle* h, const char* tf_var_name,
                  const float* data_host, const int64_t* shape_host, int rank);

/* Check that every variable the config needs is present, fold weight-norm
 * (W = V/||V||*g), repack into the kernels' MFMA-fragment order, upload. */
WN_API int wn_finalize(wn_handle* h);

/* Length helpers: samples produced for F mel frames.
 *   IAF:  T  = (F*frame_shift // 2^(num_stages-1)) * 2^(num_stages-1)  (parallel_wavenet.py:293-302)
 *   AR :  Tn = F*frame_shift                                          (fastgen.py:136,156)      */
WN_API int64_t wn_iaf_length(const wn_handle* h, int F);
WN_API int64_t wn_ar_length(const wn_handle* h, int F);

/* Bytes of caller-provided scratch needed by wn_deconv / wn_iaf_generate /
 * wn_ar_generate for batch B and F frames (Tn = wn_ar_length for AR).
 * The first 256 bytes of a workspace hold the range-guard words of the generate calls made on it
 * (wn_iaf_range_status*); every call -- wn_deconv included -- leaves them to those functions, so one buffer can
 * serve wn_iaf_generate and wn_deconv in any order. */
WN_API size_t wn_workspace_bytes(const wn_handle* h, int B, int F);

/* Wavenet.deconv_stack (wavenet.py:46-73,142-155): mel [B,F,n_mel] ->
 * enc [B, F*frame_shift, deconv_width] in the REFERENCE layout (time-major),
 * i.e. what fastgen.encode returns (fastgen.py:69-88).  `scope` is the
 * variable-name prefix: "" (teacher), "iaf_share", "iaf_1", ... */
WN_API int wn_deconv(wn_handle* h, const char* scope, const float* mel, int B, int F,
              float* enc, void* ws, size_t ws_bytes, void* stream);

/* The single sess.run of parallelgen.synthesis (parallelgen.py:43-45):
 * ParallelWavenet.feed_forward (parallel_wavenet.py:289-345) followed by
 * _clip_quant_scale (:347-359).
 *   mel        [B,F,n_mel]
 *   noise      [B,T] injected logistic/normal draws, or NULL -> drawn on the
 *              device from `seed` (Philox4x32-10; logistic via log u - log(1-u),
 *              u ~ U(1e-5,1-1e-5), or N(0,1) for loss_type gauss)
 *   wav        [B,T]  float32 on the 2/Q grid (or the inverse-mu-law table)
 *   idx        [B,T]  int32 quantisation index in [-Q/2,Q/2)        (optional)
 *   x_raw      [B,T]  feed_forward's 'x' before clipping           (optional)
 *   mean_tot, scale_tot [B,T] as returned by feed_forward          (optional)
 *   rand_out   [B,T]  the noise actually used ('rand_input')       (optional) */
WN_API int wn_iaf_generate(wn_handle* h, const float* mel, int B, int F,
                    const float* noise, uint64_t seed,
                    float* wav, int32_t* idx, float* x_raw,
                    float* mean_tot, float* scale_tot, float* rand_out,
                    void* ws, size_t ws_bytes, void* stream);

/* The same call in an explicitly named arithmetic (every student handle carries the weights in both packings), and
 * its workspace size.  wn_iaf_generate(h, ...) == wn_iaf_generate_form(h, WN_FORM_DEFAULT, ...); wn_workspace_bytes
 * (default form) also covers WN_FORM_F32 and WN_FORM_F16X3_FUSED, so a re-run never needs a larger workspace.
 *
 * fp16 range guard.  The split-fp16 forms carry activations as fp16 hi+lo pairs: 22-bit significand, but the fp16
 * EXPONENT range -- |value| >= 65504 would become inf where the reference's fp32 graph is still finite (scale may reach
 * e^7 per flow, parallel_wavenet.py:105-114).  Every kernel that produces such pairs (upsampler, start conv, residual
 * layers) checks the magnitudes it converts and raises a status word at the head of the call's workspace; the call then
 * writes NaN to every float output (idx = 0) instead of audio computed from saturated operands.  The generate call
 * itself stays asynchronous; wn_iaf_range_status(h, ws, stream) SYNCHRONISES the stream, reads that word and returns
 * WN_OK or WN_ERANGE -- on WN_ERANGE re-run the call with wn_iaf_generate_form(h, WN_FORM_F32, ...) (same workspace).
 * The Python Engine does exactly that by itself (Engine.iaf_generate(check_range=True), the default).  Never silent. */
WN_API int wn_iaf_generate_form(wn_handle* h, int form, const float* mel, int B, int F,
                         const float* noise, uint64_t seed,
                         float* wav, int32_t* idx, float* x_raw,
                         float* mean_tot, float* scale_tot, float* rand_out,
                         void* ws, size_t ws_bytes, void* stream);
WN_API size_t wn_iaf_workspace_bytes_form(const wn_handle* h, int form, int B, int F);
WN_API int wn_iaf_range_status(wn_handle* h, const void* ws, void* stream);
/* A run of calls kept asynchronous (a serving or timing loop that does not want one host synchronisation per call):
 * the second word of the workspace head accumulates the status words of every call made on that workspace.
 * wn_iaf_range_reset(h, ws, stream) zeroes the 64-byte head (do it once for a fresh workspace and before a run);
 * wn_iaf_range_status_since_reset(h, ws, stream) SYNCHRONISES, returns WN_ERANGE if ANY call since the reset left the
 * fp16 range (each such call NaN-poisoned its own outputs, as above) and resets.  Both touch only the workspace. */
WN_API int wn_iaf_range_reset(wn_handle* h, void* ws, void* stream);
WN_API int wn_iaf_range_status_since_reset(wn_handle* h, void* ws, void* stream);

/* _clip_quant_scale on its own (parallel_wavenet.py:347-359 with
 * utils.cast_quantize / inv_cast_quantize / inv_mu_law, utils.py:108-159):
 * x[n] -> wav[n], idx[n].  Bit-exact integer index. */
WN_API int wn_clip_quant(wn_handle* h, const float* x, int64_t n,
                  float* wav, int32_t* idx, void* stream);

/* ---- log-mel featuriser of the generation drivers (auxilaries/mel_extractor.py:31-44 melspectrogram,
 * :65-90 stft / mel basis / amp_to_db / normalise; called by eval_wavenet.py / eval_parallel_wavenet.py on the
 * host through librosa).  Needs no model handle; errors are reported through wn_last_error(NULL). ---- */

/* Frames of an utterance of n_samples: 1 + n_samples / 200 (centred frames, hop 12.5 ms at 16 kHz). */
WN_API int64_t wn_mel_frames(int64_t n_samples);

/* wav [B][L] float32 (device) -> mel [B][wn_mel_frames(L)][80] float32 (device), values in [0, 1]:
 * reflect-padded centred 2048-point frames, periodic Hann window of 800 samples, |STFT|, Slaney mel basis
 * (80 bands, 125-7600 Hz), 20 log10(max(1e-5, .)), clip((S + 140) / 140, 0, 1).  L must exceed 1024 (the reflect
 * padding, as in numpy.pad). */
WN_API int wn_mel_spectrogram(const float* wav, int B, int64_t L, float* mel, void* stream);

/* ---- autoregressive path (wavenet.py:379-514, fastgen.py:118-169) ---- */

/* Number of injected random values per sample and per batch element:
 * mol: mol_mix uniforms for the Gumbel-max + 1 uniform for the logistic;
 * gauss: 1 standard normal; ce: 1 uniform in [0,1) (inverse-CDF draw). */
WN_API int wn_ar_n_rand(const wn_handle* h);

/* Bytes of the explicit FIFO state (two queues per causal layer,
 * masked.py:352-355) for batch B. */
WN_API size_t wn_ar_state_bytes(const wn_handle* h, int B);

/* sess.run(init_ops) (fastgen.py:150): zero the queues, step counter := 0. */
WN_API int wn_ar_reset(wn_handle* h, void* state, int B, void* stream);

/* One sess.run([sample, push_ops]) (fastgen.py:158-161):
 *   wav_in   [B]      previous audio sample (float, already de-quantised)
 *   enc_t    [B,deconv_width]  conditioning for this step
 *   rnd      [B,n_rand] injected randoms, or NULL -> Philox(seed, step)
 *   sample   [B]      int32 in [-Q/2,Q/2)
 *   out_params [B,out_width] pre-sampling network output            (optional) */
WN_API int wn_ar_step(wn_handle* h, void* state, int B,
               const float* wav_in, const float* enc_t,
               const float* rnd, uint64_t seed,
               int32_t* sample, float* out_params, void* stream);

/* The whole loop of fastgen.synthesis (fastgen.py:150-168) for Tn steps:
 *   enc   [B,Tn,deconv_width]  (reference layout, e.g. from wn_deconv)
 *   rnd   [Tn,B,n_rand] or NULL -> Philox(seed)
 *   idx   [B,Tn] int32, wav [B,Tn] float (de-quantised feedback signal)
 *   forced_wav [B,Tn] optional teacher forcing: when non-NULL the network input
 *              at step t is forced_wav[:,t-1] instead of its own sample
 *   out_params [B,Tn,out_width] optional */
WN_API int wn_ar_generate(wn_handle* h, const float* enc, int B, int Tn,
                   const float* rnd, uint64_t seed,
                   int32_t* idx, float* wav,
                   const float* forced_wav, float* out_params,
                   void* ws, size_t ws_bytes, void* stream);

/* Fastgen.cond_vars / fastgen.calculate_cond_vars (wavenet/wavenet.py:353-377, wavenet/fastgen.py:91-115): the 1x1
 * conditioning projections of every layer and of the output stage, in bulk over time, biases included.
 * enc [B,Tn,deconv_width] (wn_deconv's output); out: num_layers tensors [B,Tn,gate_width] ('mel_cond_1' ..
 * 'mel_cond_<num_layers>') back to back, then 'mel_cond_out1' [B,Tn,skip_width] -- wn_ar_cond_vars_floats(h,B,Tn) floats. */
WN_API size_t wn_ar_cond_vars_floats(const wn_handle* h, int B, int Tn);
WN_API int wn_ar_cond_vars(wn_handle* h, const float* enc, int B, int Tn, float* out, void* stream);

/* wn_ar_generate replays the step from a hipGraph (16 steps per graph) when the caller's stream can be
 * captured (any stream but the legacy null stream); enable = 0 makes it issue plain launches instead
 * (A/B measurements, graph-vs-launch parity tests).  Default: enabled.  The steps are captured on a private
 * stream of the handle and replayed on the caller's, which is never in capture mode; a capture that fails or is
 * invalidated from outside (a device-wide synchronise of another thread) falls back to plain launches by itself. */
WN_API int wn_ar_set_graph(wn_handle* h, int enable);

/* Full-sequence teacher forward, `Wavenet.feed_forward` (wavenet/wavenet.py:180-291) for a teacher
 * handle: wav [B,T] raw audio in [-1,1] (the input encoding of wavenet.py:412-418 -- mu-law/128 when
 * use_mu_law -- is applied on the device), mel [B,F,n_mel]; out_params [B,T,out_width] are the
 * logits / mixture parameters the reference returns under 'out_params'.  T <= F*frame_shift (the
 * conditioning is centre-cropped to T, wavenet.py:76-85) and T must be a multiple of
 * 2^(num_stages-1) (masked.py:188).  Same result as wn_ar_generate with forced_wav (the reference's
 * incremental == full-sequence identity), T times fewer dependent launches. */
WN_API size_t wn_teacher_workspace_bytes(const wn_handle* h, int B, int F, int64_t T);
WN_API int wn_teacher_forward(wn_handle* h, const float* wav, const float* mel, int B, int F, int64_t T,
                       float* out_params, void* ws, size_t ws_bytes, void* stream);

/* Teacher scoring, the per-sample term of `Wavenet.calculate_loss` (wavenet/wavenet.py:293-316): log-likelihood of the
 * audio wav [B,T] under out_params [B,T,out_width] (what wn_teacher_forward wrote) -- loss_func.mol_log_probs
 * (wavenet/loss_func.py:22-63), gauss_log_prob (:104-119) or the negated cross entropy of ce_loss (:128-133), by the
 * handle's loss_type, on the targets `Wavenet.encode_signal` derives from the raw audio (wavenet.py:157-178: mu-law / 128
 * and the class index when use_mu_law, the audio itself otherwise).  log_prob [B,T]; the reference's scalar loss is
 * minus its mean.  Asynchronous on `stream`; no workspace.
 * Accuracy contract (what "parity" means here): the values are the float64-accurate ones of the reference's FORMULAS.  The
 * mixture-of-logistics bin mass is formed as sigma(a) sigma(-b) (1 - e^-(a-b)) with a - b computed directly; the reference's
 * float32 graph evaluates sigmoid(plus) - sigmoid(min), which keeps about two digits of a mass of 1/65 536 -- a loss printed
 * by a float32 TensorFlow run of the reference can therefore differ from this one in its third digit, by the reference's own
 * cancellation, not by this path's error (tests hold 2e-5 against the float64 evaluation of the reference's code).  The
 * mu-law class of a sample closer than ~1e-5 of a bin to a bin edge is decided by the rounding of a float32 log in any
 * implementation; 1e-3 of a bin away from every edge the class is exact (tests/test_gpu_teacher.py). */
WN_API int wn_teacher_log_prob(wn_handle* h, const float* out_params, const float* wav, int B, int64_t T, float* log_prob,
                        void* stream);

/* 1 when wn_iaf_generate(B, F) evaluates the per-layer conditioning 1x1s in one hoisted GEMM per
 * deconv stack (the default of the split-fp16 path: the layer kernels then stream 768 B/sample
 * instead of 1536 and the small-dilation layers run two per launch; since round 6 also the default of the fp32 form:
 * one fp32 GEMM per deconv stack, the layer kernels on the dilated conv alone), 0 for cond_mode 1 (fused), for fp32
 * calls whose length is not a multiple of 128 samples and for calls whose projected term would not fit (cond_mode 0). */
WN_API int wn_iaf_cond_hoisted(const wn_handle* h, int B, int F);

/* 1 when wn_iaf_generate(B, F) runs the residual layers in layer groups (up to five layers per launch with the residual
 * stream in LDS: one natural and one decimated group per ten-layer dilation cycle) -- the default of the hoisted form
 * whenever the flows split into alternating natural / decimated groups and the length is a multiple of 512 samples (every
 * shipped configuration), except for five to seven 4.8 s utterances per call, where the per-layer launches measured 1 %
 * ahead; 0 when every layer (or layer pair) is a launch of its own. */
WN_API int wn_iaf_layer_groups(const wn_handle* h, int B, int F);

/* Launch structure of the hoisted form for this handle: mode 1 = layer groups at every call size they support, -1 = never
 * (one launch per layer or layer pair), 0 = back to what the handle was created with (the size policy above, or the form
 * the environment variables WN_GROUPS=1 / WN_NO_GROUPS=1 named when wn_create read them -- ONCE).  A switch (see
 * "Concurrency"): WN_ESTATE while another thread's work call is inside the library; cross-form tests and A/B runs use it
 * between calls.  Returns WN_EINVAL for any other mode. */
WN_API int wn_iaf_set_groups(wn_handle* h, int mode);

/* MEASUREMENT ONLY: aid used by bench.py (not part of the reference's surface; a switch in the sense of
 * "Concurrency" above).  Between begin and end, wn_iaf_generate records a hipEvent pair
 * on the caller's stream around every flow's run of residual-stack launches --
 * whatever form the call takes: iaf_group_kernel (layer groups, the default of small
 * calls), iaf_layer_c_kernel / iaf_pair_c_kernel (one / two layers per launch),
 * iaf_layer_h_kernel (fused form), iaf_layer_kernel (fp32 form).  wn_profile_end
 * synchronises those events and returns the summed elapsed milliseconds and the
 * number of launches they bracket: average launch duration = layer_ms / layer_launches.
 * Every recorded event costs the stream a bubble of a few microseconds, so
 * wn_profile_pause(h, 1) suspends the recording for the following calls (0 resumes):
 * bench.py samples every few steps of its timed region instead of all of them. */
WN_API int wn_profile_begin(wn_handle* h);
WN_API int wn_profile_pause(wn_handle* h, int paused);
WN_API int wn_profile_end(wn_handle* h, double* layer_ms, int64_t* layer_launches);

/* MEASUREMENT ONLY: second mode, same caveats: between parts_begin and parts_end every wn_iaf_generate records one hipEvent on
 * the caller's stream where a PART of the call begins -- 0: prologue + epilogue (pads, noise draw, final clip/quantise),
 * 1: mel upsampler, 2: conditioning GEMM (hoisted form), 3: residual stack (start convs, layers / layer groups, flow
 * heads).  parts_end synchronises and returns the summed milliseconds per part (part_ms[WN_PROFILE_PARTS]) and the
 * number of calls recorded: bench.py's in-process replacement for a replayed rocprofv3 kernel trace.  Each event costs
 * the stream a bubble of a few microseconds, so the sum of the parts is slightly above the unprofiled call time. */
#define WN_PROFILE_PARTS 4
WN_API int wn_profile_parts_begin(wn_handle* h);
WN_API int wn_profile_parts_end(wn_handle* h, double* part_ms, int64_t* calls);

/* MEASUREMENT ONLY (package power of one part of the call, scripts/dev_power.py) -- a restricted call SKIPS WORK and its
 * results are meaningless, so the switch cannot be left armed: it is accepted only between wn_profile_parts_begin and
 * wn_profile_parts_end (WN_ESTATE otherwise), and parts_end (like parts_begin and wn_destroy) puts the handle back to "every
 * part".  While armed, the following wn_iaf_generate calls of the split-fp16 / fp32 student paths run only the parts whose
 * bits are set in mask (bit k = part k above; 15 = everything) -- e.g. a loop of conditioning GEMMs alone; the skipped parts
 * leave their buffers as the last full call wrote them.  At most 4096 part events are kept per session (a long power loop
 * stops recording, the restriction stays).  WN_EINVAL for a mask outside 1..15. */
WN_API int wn_profile_parts_only(wn_handle* h, int mask);

/* Message of the calling thread's last failed call (any handle, or wn_create).  `h` is accepted for source compatibility
 * and ignored: the text is thread-local, so concurrent callers never see -- or tear -- each other's messages. */
WN_API const char* wn_last_error(const wn_handle* h);

WN_API void wn_destroy(wn_handle* h);

/* CRC32C (Castagnoli) of n host bytes, continuing from `crc` (0 to start): the checksum of the TensorFlow V2 checkpoint
 * bundle (nsynth_wavenet_amd/tf_bundle.py reads and writes `model.ckpt-N.index` / `.data-*` with it; the reference leaves
 * this to tf.train.Saver, wavenet/fastgen.py:80-84).  Host-only helper, needs no handle and no device. */
WN_API uint32_t wn_crc32c(const void* data_host, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* WNHIP_H_ */
