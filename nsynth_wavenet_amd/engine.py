"""Engine: one model on one GPU behind the C ABI (include/wnhip.h).

PyTorch-ROCm is used only for device memory, streams and (in dist.py)
torch.distributed; every arithmetic operation of the generation path runs in the
hand-written HIP kernels of libwnhip.so.  There is no CPU or eager fallback: a
missing library or a missing GPU raises.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import config as cfg
from . import weights as wts


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Engine(object):
    """Replaces one TF graph + Session of the reference (parallelgen.py:24-41,
    fastgen.py:139-150): built from hparams, filled with named weights."""

    def __init__(self, hparams, kind=None, device=None, n_mel=80, precision=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('nsynth_wavenet_amd needs a ROCm GPU: the generation path has no CPU fallback')
        self.hp = cfg.load_hparams(hparams)
        self.kind = kind or cfg.model_kind(self.hp)
        self.device = torch.device(device if device is not None else 'cuda:{}'.format(torch.cuda.current_device()))
        self.n_mel = n_mel
        self._hbox = [ctypes.c_void_p(0)]   # the C handle, shared with the engine's forks (fork())
        self._shared = {'ar_graph': None}   # handle-wide switch states the forks must agree on
        self._is_fork = False
        self._ws = None
        self._ar_stream = None
        self._ar_groups = None              # per-group streams / queue states of ar_generate(streams=G)
        self._last_ws = None
        self.range_fallbacks = 0          # calls re-run on the fp32 form because they left the fp16 range
        self._finalized = False
        self.precision = precision or cfg.default_precision()
        c = cfg.to_wn_config(self.hp, self.kind, n_mel, self.precision)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.wn_create(ctypes.byref(c), ctypes.byref(self._hbox[0])))
        self.quant_chann = cfg.quant_chann(self.hp)
        self.frame_shift = cfg.frame_shift(self.hp)

    @property
    def _h(self):
        return self._hbox[0]

    # ---- lifecycle ----
    def close(self):
        """Destroy the C handle (the owner) / drop this caller's workspace (a fork: the handle belongs to its parent)."""
        self._ws = self._last_ws = None
        if self._is_fork:
            return
        if self._h:
            torch.cuda.synchronize(self.device)
            self.lib.wn_destroy(self._h)
            self._hbox[0] = ctypes.c_void_p(0)

    def fork(self):
        """A second CALLER of the same C handle (SURVEY 8(b) "Threading / streams": a finalized handle is immutable and
        may be shared by concurrent callers using distinct workspaces and streams): shares the packed weights on the
        device, owns its workspace, and issues its work on the torch stream that is current in the calling thread
        (`with torch.cuda.stream(s): fork.iaf_generate(...)`).  Forks are what host threads use -- one fork per thread;
        an Engine object itself (one workspace) is not to be shared between threads.  The parent must outlive its forks."""
        import copy
        if not self._finalized:
            raise RuntimeError('fork() needs a finalized engine (load_weights first)')
        f = copy.copy(self)                 # same _hbox / _shared objects, same hparams
        f._is_fork = True
        f._ws = f._last_ws = f._ar_stream = None
        f._ar_groups = None
        f.range_fallbacks = 0
        return f

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        _lib.check(rc, self._h)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- weights ----
    def set_weight(self, name, array):
        a = np.ascontiguousarray(np.asarray(array, np.float32))
        shape = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
        self._check(self.lib.wn_set_weight(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                           shape, a.ndim))

    def load_weights(self, weights):
        """weights: dict TF-variable-name -> ndarray (see weights.expected_variables)."""
        for name, _ in wts.expected_variables(self.hp, self.kind, self.n_mel):
            if name not in weights:
                raise KeyError('missing variable {}'.format(name))
            self.set_weight(name, weights[name])
        with torch.cuda.device(self.device):
            self._check(self.lib.wn_finalize(self._h))
        self._finalized = True
        return self

    def load_checkpoint(self, path):
        return self.load_weights(wts.load_checkpoint(path, self.hp, self.kind))

    # ---- helpers ----
    def iaf_length(self, F):
        return int(self.lib.wn_iaf_length(self._h, int(F)))

    def ar_length(self, F):
        return int(self.lib.wn_ar_length(self._h, int(F)))

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            if self._ws is not None:
                self.check_range()        # asynchronous calls not yet asked about: their status words live in the old buffer
            self._ws = self._last_ws = None
            try:
                self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
                if self._ws.numel() >= 64:                 # the range-guard words at its head start from zero
                    with torch.cuda.device(self.device):
                        self._check(self.lib.wn_iaf_range_reset(self._h, _ptr(self._ws), self._stream()))
            except torch.cuda.OutOfMemoryError as e:
                raise MemoryError('workspace of {:.1f} GB does not fit on {} ({}); for the IAF path '
                                  "precision='f16x3-fused' needs no conditioning workspace, or split the batch"
                                  .format(nbytes / 1e9, self.device, str(e).splitlines()[0]))
        return self._ws

    def _dev(self, x, dtype=torch.float32):
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(self.device)

    # ---- IAF path ----
    def iaf_generate(self, mel, noise=None, seed=0, want=('wav',), check_range=True):
        """ParallelWavenet.feed_forward + _clip_quant_scale.  mel [B,F,n_mel].
        Returns a dict of device tensors for the names in `want` out of
        wav, idx, x, mean_tot, scale_tot, rand_input.

        check_range (default): the split-fp16 arithmetic carries activations with the fp16 exponent range; when a
        call leaves it (|activation| >= 65504, possible after flows with scales near e^7) the library NaN-poisons the
        outputs and raises a status word.  The engine then reads that word (one 4-byte read-back, which synchronises
        the stream) and transparently re-runs the call on the fp32-MFMA form, so the caller always gets the
        reference's fp32 behaviour; `self.range_fallbacks` counts those re-runs.  check_range=False keeps the call
        asynchronous (serving / timing loops): call check_range() afterwards -- it raises if ANY call since the last
        check overflowed (the library accumulates the status words in the workspace; each such call NaN-poisoned its
        own outputs)."""
        mel = self._dev(mel)
        if mel.dim() != 3 or mel.shape[2] != self.n_mel:
            raise ValueError('mel must be [batch, frames, {}], got {}'.format(self.n_mel, tuple(mel.shape)))
        B, F = int(mel.shape[0]), int(mel.shape[1])
        T = self.iaf_length(F)
        noise = self._dev(noise)
        if noise is not None and tuple(noise.shape) != (B, T):
            raise ValueError('noise must be [{}, {}], got {}'.format(B, T, tuple(noise.shape)))
        out = {}
        new = lambda dt=torch.float32: torch.empty((B, T), dtype=dt, device=self.device)
        wav = new()
        idx = new(torch.int32) if 'idx' in want else None
        xr = new() if 'x' in want else None
        mt = new() if 'mean_tot' in want else None
        st = new() if 'scale_tot' in want else None
        ro = new() if 'rand_input' in want else None
        with torch.cuda.device(self.device):
            form = _lib.FORM_DEFAULT
            try:
                ws = self._workspace(self.lib.wn_workspace_bytes(self._h, B, F))
            except MemoryError:
                # the hoisted-conditioning workspace (17.5 KB per generated sample) does not fit beside what else lives
                # on this GPU: the fused form needs none of it
                if self.precision not in ('f16x3', 'f16x3-hoisted'):
                    raise
                form = _lib.FORM_F16X3_FUSED
                ws = self._workspace(self.lib.wn_iaf_workspace_bytes_form(self._h, form, B, F))

            def run(f):
                self._check(self.lib.wn_iaf_generate_form(
                    self._h, f, _ptr(mel), B, F, _ptr(noise), ctypes.c_uint64(int(seed)), _ptr(wav), _ptr(idx),
                    _ptr(xr), _ptr(mt), _ptr(st), _ptr(ro), _ptr(ws), ws.numel(), self._stream()))
            run(form)
            self._last_ws = ws
            if check_range and not self.precision.startswith('f32') and T > 0:
                rc = self.lib.wn_iaf_range_status(self._h, _ptr(ws), self._stream())
                if rc == _lib.WN_ERANGE:
                    self.range_fallbacks += 1
                    run(_lib.FORM_F32)      # same seed -> the same Philox draws when the noise is device-drawn
                    # handled here: not to be reported again by a later check_range()
                    self._check(self.lib.wn_iaf_range_reset(self._h, _ptr(ws), self._stream()))
                else:
                    self._check(rc)
        for k, v in (('wav', wav), ('idx', idx), ('x', xr), ('mean_tot', mt), ('scale_tot', st),
                     ('rand_input', ro)):
            if k in want:
                out[k] = v
        return out

    def check_range(self):
        """Raise if any iaf_generate(check_range=False) call since the last check left the fp16 range (the outputs of
        such a call are NaN); synchronises the stream once."""
        if self._last_ws is None or self.precision.startswith('f32'):
            return
        with torch.cuda.device(self.device):
            self._check(self.lib.wn_iaf_range_status_since_reset(self._h, _ptr(self._last_ws), self._stream()))

    def clip_quant(self, x):
        x = self._dev(x)
        wav = torch.empty_like(x)
        idx = torch.empty(x.shape, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.wn_clip_quant(self._h, _ptr(x), x.numel(), _ptr(wav), _ptr(idx), self._stream()))
        return wav, idx

    def deconv(self, mel, scope=None):
        """Wavenet.deconv_stack: mel [B,F,n_mel] -> [B, F*frame_shift, deconv_width]."""
        mel = self._dev(mel)
        B, F = int(mel.shape[0]), int(mel.shape[1])
        if scope is None:
            scope = '' if self.kind == 'teacher' else \
                ('iaf_share' if (getattr(self.hp, 'use_share_deconv', False) or
                                 getattr(self.hp, 'use_teacher_deconv', False)) else 'iaf_1')
        enc = torch.empty((B, F * self.frame_shift, self.hp.deconv_width), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            ws = self._workspace(self.lib.wn_workspace_bytes(self._h, B, F))
            self._check(self.lib.wn_deconv(self._h, scope.encode(), _ptr(mel), B, F, _ptr(enc), _ptr(ws),
                                           ws.numel(), self._stream()))
        return enc

    # ---- teacher, whole sequence at once ----
    def teacher_forward(self, wav, mel):
        """Wavenet.feed_forward (wavenet.py:180-291): wav [B,T] raw audio, mel [B,F,n_mel] ->
        out_params [B,T,out_width].  T <= F*frame_shift, multiple of the largest dilation."""
        wav, mel = self._dev(wav), self._dev(mel)
        if wav.dim() != 2 or mel.dim() != 3 or wav.shape[0] != mel.shape[0]:
            raise ValueError('teacher_forward: wav must be [B,T] and mel [B,F,n_mel] with equal B')
        if int(mel.shape[2]) != self.n_mel:
            raise ValueError('teacher_forward: mel has {} channels, the model expects {}'.format(
                int(mel.shape[2]), self.n_mel))
        B, T, F = int(wav.shape[0]), int(wav.shape[1]), int(mel.shape[1])
        out = torch.empty((B, T, cfg.teacher_out_width(self.hp)), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            nb = self.lib.wn_teacher_workspace_bytes(self._h, B, F, T)
            ws = self._workspace(max(int(nb), 256))
            self._check(self.lib.wn_teacher_forward(self._h, _ptr(wav), _ptr(mel), B, F, T, _ptr(out), _ptr(ws),
                                                    ws.numel(), self._stream()))
        return out

    def teacher_log_prob(self, out_params, wav):
        """Per-sample log-likelihood of wav [B,T] under out_params [B,T,out_width] (loss_func.py:22-63,104-119,128-133 on the
        targets of Wavenet.encode_signal, wavenet.py:157-178) -> [B,T]; Wavenet.calculate_loss's 'loss' is minus its mean."""
        out_params, wav = self._dev(out_params), self._dev(wav)
        if wav.dim() != 2 or out_params.dim() != 3 or tuple(out_params.shape[:2]) != tuple(wav.shape) or \
                int(out_params.shape[2]) != cfg.teacher_out_width(self.hp):
            raise ValueError('teacher_log_prob: out_params must be [B,T,{}] and wav [B,T]'.format(cfg.teacher_out_width(self.hp)))
        lp = torch.empty(tuple(wav.shape), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.wn_teacher_log_prob(self._h, _ptr(out_params), _ptr(wav), int(wav.shape[0]), int(wav.shape[1]),
                                                     _ptr(lp), self._stream()))
        return lp

    def iaf_cond_hoisted(self, batch, num_frames):
        """True when iaf_generate(batch, num_frames) runs the hoisted-conditioning kernels."""
        return bool(self.lib.wn_iaf_cond_hoisted(self._h, int(batch), int(num_frames)))

    def iaf_layer_groups(self, batch, num_frames):
        """True when iaf_generate(batch, num_frames) runs the residual layers in LDS-resident layer groups."""
        return bool(self.lib.wn_iaf_layer_groups(self._h, int(batch), int(num_frames)))

    def set_layer_groups(self, mode):
        """Launch structure of the hoisted form: True / 1 = layer groups wherever they apply, False / -1 = one launch per
        layer (pair), None / 0 = what the engine was created with: the library's size policy, or the form WN_GROUPS /
        WN_NO_GROUPS named then (wn_iaf_set_groups).  A switch: raises RuntimeError (WN_ESTATE) while another thread's
        generate call is inside the library."""
        m = 0 if mode is None else (1 if mode is True else -1 if mode is False else int(mode))
        self._check(self.lib.wn_iaf_set_groups(self._h, m))
        return self

    # ---- measurement aid (bench.py) ----
    def profile_begin(self):
        self._check(self.lib.wn_profile_begin(self._h))

    def profile_pause(self, paused):
        self._check(self.lib.wn_profile_pause(self._h, 1 if paused else 0))

    def profile_end(self):
        """-> (summed ms of the bracketed residual-layer kernel runs, number of launches)."""
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        self._check(self.lib.wn_profile_end(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    PROFILE_PARTS = ('prologue_epilogue', 'upsampler', 'cond_gemm', 'residual_stack')

    def profile_parts_begin(self):
        self._check(self.lib.wn_profile_parts_begin(self._h))

    def profile_parts_only(self, mask):
        """Power measurements only, and only between profile_parts_begin() and profile_parts_end() (which disarms it): run
        just the parts whose bits are set (bit k = PROFILE_PARTS[k]; 15 = everything)."""
        self._check(self.lib.wn_profile_parts_only(self._h, int(mask)))

    def profile_parts_end(self):
        """-> ({part: summed ms}, calls): HIP events at the part boundaries of every iaf_generate since parts_begin."""
        ms = (ctypes.c_double * len(self.PROFILE_PARTS))()
        n = ctypes.c_int64(0)
        self._check(self.lib.wn_profile_parts_end(self._h, ms, ctypes.byref(n)))
        return {k: ms[i] for i, k in enumerate(self.PROFILE_PARTS)}, n.value

    # ---- AR path ----
    def ar_n_rand(self):
        return int(self.lib.wn_ar_n_rand(self._h))

    def ar_new_state(self, B):
        n = self.lib.wn_ar_state_bytes(self._h, int(B))
        if n == 0:
            raise ValueError('not a finalized teacher engine')
        st = torch.empty(int(n), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.wn_ar_reset(self._h, _ptr(st), int(B), self._stream()))
        return st

    def _check_rnd(self, rnd, shape, what):
        if rnd is not None and tuple(rnd.shape) != shape:
            raise ValueError('{}: rnd must be {}, got {}'.format(what, shape, tuple(rnd.shape)))

    def ar_step(self, state, wav_in, enc_t, rnd=None, seed=0, want_out=False):
        """One Fastgen.sample step.  wav_in [B] or [B,1]; enc_t [B,deconv_width]; rnd [B, ar_n_rand()]."""
        wav_in = self._dev(wav_in).reshape(-1)
        enc_t = self._dev(enc_t)
        B = int(wav_in.shape[0])
        if enc_t.dim() != 2 or tuple(enc_t.shape) != (B, int(self.hp.deconv_width)):
            raise ValueError('ar_step: encoding must be [{}, {}], got {}'.format(B, self.hp.deconv_width, tuple(enc_t.shape)))
        if not isinstance(state, torch.Tensor) or state.numel() != int(self.lib.wn_ar_state_bytes(self._h, B)):
            raise ValueError('ar_step: state was not created by ar_new_state({})'.format(B))
        rnd = self._dev(rnd)
        if rnd is not None and rnd.dim() == 1:
            rnd = rnd.reshape(B, -1)
        self._check_rnd(rnd, (B, self.ar_n_rand()), 'ar_step')
        sample = torch.empty((B,), dtype=torch.int32, device=self.device)
        outp = torch.empty((B, cfg.teacher_out_width(self.hp)), dtype=torch.float32, device=self.device) \
            if want_out else None
        with torch.cuda.device(self.device):
            self._check(self.lib.wn_ar_step(self._h, _ptr(state), B, _ptr(wav_in), _ptr(enc_t), _ptr(rnd),
                                            ctypes.c_uint64(int(seed)), _ptr(sample), _ptr(outp), self._stream()))
        return (sample, outp) if want_out else sample

    def ar_cond_vars(self, enc):
        """Fastgen.cond_vars (wavenet.py:353-377): enc [B,Tn,deconv_width] -> {'mel_cond_1': [B,Tn,gate_width], ...,
        'mel_cond_<num_layers>': ..., 'mel_cond_out1': [B,Tn,skip_width]}, every layer's conditioning projection (bias
        included) in bulk over time.  The tensors are views of one device buffer."""
        enc = self._dev(enc)
        if enc.dim() != 3 or int(enc.shape[2]) != int(self.hp.deconv_width):
            raise ValueError('ar_cond_vars: enc must be [batch, steps, {}], got {}'.format(self.hp.deconv_width, tuple(enc.shape)))
        B, Tn = int(enc.shape[0]), int(enc.shape[1])
        G, S, NL = cfg.teacher_gate_width(self.hp), int(self.hp.skip_width), int(self.hp.num_layers)
        n = int(self.lib.wn_ar_cond_vars_floats(self._h, B, Tn))
        if n != B * Tn * (NL * G + S):
            raise ValueError('ar_cond_vars: needs a teacher engine and B, Tn >= 1')
        buf = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.wn_ar_cond_vars(self._h, _ptr(enc), B, Tn, _ptr(buf), self._stream()))
        out = {}
        for i in range(NL):
            out['mel_cond_%d' % (i + 1)] = buf[i * B * Tn * G:(i + 1) * B * Tn * G].view(B, Tn, G)
        out['mel_cond_out1'] = buf[NL * B * Tn * G:].view(B, Tn, S)
        return out

    def _set_ar_graph(self, on):
        """wn_ar_set_graph is a SWITCH of the shared handle (refused while another caller's work call is in flight): flipped
        only when the wanted state differs from the one the engine and its forks last set."""
        on = bool(on)
        if self._shared['ar_graph'] is not on:
            self._check(self.lib.wn_ar_set_graph(self._h, 1 if on else 0))
            self._shared['ar_graph'] = on

    def ar_generate(self, enc, rnd=None, seed=0, forced_wav=None, want_out=False, use_graph=False, streams=1):
        """fastgen.synthesis loop.  enc [B,Tn,deconv_width], rnd [Tn,B,ar_n_rand()] or None (drawn on the
        device), forced_wav [B,Tn] or None -> dict(idx, wav[, out_params]).
        The loop is a static sequence of kernels per step (every per-step address is derived on the device).
        Default: plain launches on the caller's stream.  use_graph=True replays it from hipGraphs (16 steps per
        graph); a graph cannot be captured on the legacy null stream PyTorch uses by default, so that form runs
        on a side stream of its own, ordered after and joined back into the caller's current stream.  Measured
        on MI355X the two are equally fast (191 vs 193 us per step at one utterance: the dependent-kernel chain
        on the GPU is the bound, not launch submission) and the capture costs ~20 ms per call, hence the default.

        streams=G > 1: the batch is cut into G contiguous groups of utterances, each an independent chain with its own
        queue state on its own stream, all against the one shared handle (utterances never interact: wavenet.py:379-514
        has no cross-batch term, so the results are those of the single-stream call row for row).  A step is ~65
        dependent launches of a few microseconds of work each and leaves the GPU idle in between; chains on other streams
        fill those gaps.  Each group replays hipGraphs (cheap to enqueue from one host thread), whatever use_graph says."""
        enc = self._dev(enc)
        if enc.dim() != 3 or int(enc.shape[2]) != int(self.hp.deconv_width):
            raise ValueError('ar_generate: enc must be [batch, steps, {}], got {}'.format(self.hp.deconv_width, tuple(enc.shape)))
        B, Tn = int(enc.shape[0]), int(enc.shape[1])
        rnd = self._dev(rnd)
        self._check_rnd(rnd, (Tn, B, self.ar_n_rand()), 'ar_generate')
        forced = self._dev(forced_wav)
        if forced is not None and tuple(forced.shape) != (B, Tn):
            raise ValueError('ar_generate: forced_wav must be [{}, {}], got {}'.format(B, Tn, tuple(forced.shape)))
        G = int(streams)
        if G < 1 or G > B:
            raise ValueError('ar_generate: streams must be in 1..batch ({}), got {}'.format(B, streams))
        if G > 1:
            return self._ar_generate_streams(enc, rnd, seed, forced, want_out, G)
        idx = torch.empty((B, Tn), dtype=torch.int32, device=self.device)
        wav = torch.empty((B, Tn), dtype=torch.float32, device=self.device)
        outp = torch.empty((B, Tn, cfg.teacher_out_width(self.hp)), dtype=torch.float32, device=self.device) \
            if want_out else None
        with torch.cuda.device(self.device):
            nb = self.lib.wn_ar_state_bytes(self._h, B)
            ws = self._workspace(nb)
            cur = torch.cuda.current_stream(self.device)
            self._set_ar_graph(use_graph)
            if use_graph:
                if self._ar_stream is None:
                    self._ar_stream = torch.cuda.Stream(self.device)
                run = self._ar_stream
                run.wait_stream(cur)
            else:
                run = cur
            self._check(self.lib.wn_ar_generate(
                self._h, _ptr(enc), B, Tn, _ptr(rnd), ctypes.c_uint64(int(seed)), _ptr(idx), _ptr(wav),
                _ptr(forced), _ptr(outp), _ptr(ws), ws.numel(), ctypes.c_void_p(run.cuda_stream)))
            if use_graph:
                cur.wait_stream(run)
        out = {'idx': idx, 'wav': wav}
        if want_out:
            out['out_params'] = outp
        return out

    def _ar_generate_streams(self, enc, rnd, seed, forced, want_out, G):
        B, Tn = int(enc.shape[0]), int(enc.shape[1])
        ow = cfg.teacher_out_width(self.hp)
        if self._ar_groups is None or len(self._ar_groups) < G:
            self._ar_groups = [{'stream': torch.cuda.Stream(self.device), 'ws': None} for _ in range(G)]
        bounds = [B * g // G for g in range(G + 1)]
        parts = []
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            self._set_ar_graph(True)
            for g in range(G):
                lo, hi = bounds[g], bounds[g + 1]
                nb_ = hi - lo
                grp = self._ar_groups[g]
                need = int(self.lib.wn_ar_state_bytes(self._h, nb_))
                if grp['ws'] is None or grp['ws'].numel() < need:
                    grp['ws'] = torch.empty(need, dtype=torch.uint8, device=self.device)
                # per-group inputs / outputs are contiguous tensors of their own (the C ABI takes dense [b, Tn, ...] arrays)
                e = enc[lo:hi].contiguous()
                r = rnd[:, lo:hi].contiguous() if rnd is not None else None
                f = forced[lo:hi].contiguous() if forced is not None else None
                idx = torch.empty((nb_, Tn), dtype=torch.int32, device=self.device)
                wav = torch.empty((nb_, Tn), dtype=torch.float32, device=self.device)
                outp = torch.empty((nb_, Tn, ow), dtype=torch.float32, device=self.device) if want_out else None
                parts.append((e, r, f, idx, wav, outp))
            for g in range(G):
                self._ar_groups[g]['stream'].wait_stream(cur)
            for g in range(G):
                e, r, f, idx, wav, outp = parts[g]
                grp = self._ar_groups[g]
                # device-drawn randoms: every group draws from its own Philox key (seed + group); injected randoms are
                # the caller's columns lo..hi, so the result is the single-stream call's row for row
                self._check(self.lib.wn_ar_generate(
                    self._h, _ptr(e), int(e.shape[0]), Tn, _ptr(r), ctypes.c_uint64(int(seed) + g), _ptr(idx), _ptr(wav),
                    _ptr(f), _ptr(outp), _ptr(grp['ws']), grp['ws'].numel(), ctypes.c_void_p(grp['stream'].cuda_stream)))
            for g in range(G):
                cur.wait_stream(self._ar_groups[g]['stream'])
            # (every tensor above was allocated on `cur`, the side streams start behind it and are joined back into it:
            # the caching allocator can only hand their memory to work that `cur` orders behind the joins)
        out = {'idx': torch.cat([p[3] for p in parts], 0), 'wav': torch.cat([p[4] for p in parts], 0)}
        if want_out:
            out['out_params'] = torch.cat([p[5] for p in parts], 0)
        return out
