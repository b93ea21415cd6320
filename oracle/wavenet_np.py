"""numpy restatement of the nsynth_wavenet generation path (TEST INFRASTRUCTURE).

PARITY (see oracle/__init__.py): restated from the closed forms of the reference's TensorFlow-op compositions;
held to the reference's OWN code executed over a numpy evaluator of the TensorFlow primitives (tests/test_ref_float.py:
float64 rounding, identical index streams); not validated against a TensorFlow run (TensorFlow kernel semantics unpinned).

Every function cites the reference file:line it follows (paths relative to the
reference tree).  All arrays are [B, T, C] channels-last like the reference.
`dtype` selects the arithmetic type: np.float64 = master, np.float32 = mirror of
what the reference computes in.

Weights are a dict  TF-variable-name -> ndarray  with the reference's shapes:
conv kernels HWIO [1,K,Cin,Cout] (`<scope>/W`, `<scope>/biases`), transposed-conv
kernels [1,K,Cout,Cin] (`<scope>/kernel`, `<scope>/bias`).
"""
import numpy as np

EXP_M9 = float(np.exp(-9.0))
EXP_7 = float(np.exp(7.0))


# --------------------------------------------------------------------------
# hparams (JSON dict -> attribute access with the reference's per-class defaults)
# --------------------------------------------------------------------------
class HP(object):
    """argparse.Namespace-alike over the config JSON (eval_wavenet.py:28-30)."""

    def __init__(self, d):
        self.__dict__.update(d)

    def get(self, k, default):
        return getattr(self, k, default)


def quant_chann_of(hp):
    # wavenet.py:117-120, parallel_wavenet.py:137-140
    return 2 ** 8 if hp.use_mu_law else 2 ** 16


def teacher_out_width(hp):
    # wavenet.py:121-129
    if hp.loss_type == 'ce':
        return quant_chann_of(hp)
    if hp.loss_type == 'mol':
        return hp.mol_mix * 3
    if hp.loss_type == 'gauss':
        return 2
    raise ValueError('[{}] loss is not supported'.format(hp.loss_type))


def teacher_gate_width(hp):
    # wavenet.py:106,203 : double_gate_width defaults to True when absent
    return 2 * hp.width if hp.get('double_gate_width', True) else hp.width


# --------------------------------------------------------------------------
# codecs  (auxilaries/utils.py:72-169)
# --------------------------------------------------------------------------
def mu_law(x, mu=255, dtype=np.float32):
    """utils.py:72-105: floor(sign(x)*log(1+mu|x|)/log(1+mu)*128)."""
    x = np.asarray(x, dtype)
    out = np.sign(x) * np.log(dtype(1) + dtype(mu) * np.abs(x)) / dtype(np.log(1 + mu))
    return np.floor(out * dtype(128))


def inv_mu_law(q, mu=255, dtype=np.float32):
    """utils.py:108-139: s=(q+.5)*2/(mu+1); sign(s)/mu*((1+mu)^|s|-1); 0 where q==0."""
    x = np.asarray(q).astype(dtype)
    out = (x + dtype(0.5)) * dtype(2.0) / dtype(mu + 1)
    out = np.sign(out) / dtype(mu) * (dtype(1 + mu) ** np.abs(out) - dtype(1))
    return np.where(x == 0, x, out).astype(dtype)


def cast_quantize(x, quant_chann, dtype=np.float32):
    """utils.py:142-154: int32(floor(x*Q/2))."""
    x = np.asarray(x, dtype)
    return np.floor(x * dtype(quant_chann) / dtype(2)).astype(np.int32)


def inv_cast_quantize(q, quant_chann, dtype=np.float32):
    """utils.py:157-159,167-169: q/(Q/2)."""
    return (np.asarray(q).astype(dtype) / dtype(quant_chann / 2)).astype(dtype)


def clip_quant_scale(x, quant_chann, use_mu_law, dtype=np.float32):
    """parallel_wavenet.py:347-359.  Returns (wav, int32 index)."""
    x = np.asarray(x, dtype)
    x = np.clip(x, dtype(-1.0), dtype(1.0 - 2.0 / quant_chann))
    q = cast_quantize(x, quant_chann, dtype)
    if use_mu_law:
        return inv_mu_law(q, dtype=dtype), q
    return inv_cast_quantize(q, quant_chann, dtype), q


# --------------------------------------------------------------------------
# elementwise helpers with TF semantics
# --------------------------------------------------------------------------
def softplus(x):
    """tf.nn.softplus (Eigen functor): x above -threshold, exp(x) below threshold,
    log1p(exp(x)) between; threshold = log(eps)+2."""
    x = np.asarray(x)
    thr = np.log(np.finfo(x.dtype).eps) + 2.0
    with np.errstate(over='ignore'):
        ex = np.exp(x)
        mid = np.log1p(ex)
    return np.where(x > -thr, x, np.where(x < thr, ex, mid)).astype(x.dtype)


def sigmoid(x):
    with np.errstate(over='ignore'):
        return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def upsample_act(name):
    """masked.py:28-36."""
    if name == 'tanh':
        return np.tanh
    if name == 'relu':
        return lambda v: np.maximum(v, 0)
    if name == 'leaky_relu':
        return lambda v: np.maximum(v, v.dtype.type(0.4) * v)
    raise ValueError('Unsupported activation function for upsample layer')


# --------------------------------------------------------------------------
# kernels / weight-norm   (masked.py:131-157)
# --------------------------------------------------------------------------
def get_kernel(weights, scope, name, use_weight_norm=False, deconv=False, dtype=np.float32):
    if not use_weight_norm:
        return np.asarray(weights['{}/{}'.format(scope, name)], dtype)
    V = np.asarray(weights['{}/{}_V'.format(scope, name)], dtype)
    g = np.asarray(weights['{}/{}_g'.format(scope, name)], dtype)
    if deconv:
        axes, gshape = (0, 1, 3), (1, 1, -1, 1)
    else:
        axes, gshape = (0, 1, 2), (1, 1, 1, -1)
    # tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))
    ss = np.maximum(np.sum(V * V, axis=axes, keepdims=True), dtype(1e-12))
    return (V / np.sqrt(ss) * g.reshape(gshape)).astype(dtype)


# --------------------------------------------------------------------------
# conv ops   (masked.py:39-52, 160-291)
# --------------------------------------------------------------------------
def shift_right(x):
    """masked.py:39-52: y[t]=x[t-1], y[0]=0."""
    y = np.zeros_like(x)
    y[:, 1:] = x[:, :-1]
    return y


def _delay(x, n):
    if n == 0:
        return x
    y = np.zeros_like(x)
    if n < x.shape[1]:
        y[:, n:] = x[:, :-n]
    return y


def conv1d(x, W, b, dilation=1):
    """Causal dilated conv, closed form of masked.py:160-232 (time_to_batch +
    VALID conv2d + batch_to_time):  y[t] = b + sum_k x[t-(K-1-k)d] @ W[0,k]."""
    assert x.shape[1] % dilation == 0  # masked.py:188
    K = W.shape[1]
    y = np.zeros(x.shape[:2] + (W.shape[3],), x.dtype) + b
    for k in range(K):
        y = y + _delay(x, (K - 1 - k) * dilation) @ W[0, k]
    return y


def trans_conv1d(x, W, b, stride, act):
    """masked.py:235-291: conv2d_transpose SAME, stride s, filter [1,K,Cout,Cin];
    y[n] = b + sum_i x[i] @ W[0, n+pL-i*s].T, pL=(K-s)//2, then activation."""
    B, L, _ = x.shape
    K, Cout = W.shape[1], W.shape[2]
    s = stride
    pL = max(K - s, 0) // 2
    full = np.zeros((B, s * L + K, Cout), x.dtype)
    for k in range(K):
        full[:, k:k + s * L:s] += x @ W[0, k].T
    y = full[:, pL:pL + s * L] + b
    return act(y) if act is not None else y


def resize_conv1d(x, W, b, stride, act):
    """masked.py:294-322: tf.image.resize_nearest_neighbor along time (src = floor(t / stride)),
    then the NON-causal branch of conv1d (masked.py:191,209: conv2d SAME, stride 1: zero padding
    (K-1)//2 on the left, the rest on the right), then activation.  W is HWIO [1,K,Cin,Cout]."""
    K = W.shape[1]
    u = np.repeat(x, stride, axis=1)
    pl = (K - 1) // 2
    up = np.pad(u, [(0, 0), (pl, K - 1 - pl), (0, 0)])
    T = u.shape[1]
    y = np.zeros((x.shape[0], T, W.shape[3]), x.dtype) + b
    for k in range(K):
        y = y + up[:, k:k + T] @ W[0, k]
    return act(y) if act is not None else y


def condition(x, cond):
    """wavenet.py:76-85: centre-crop cond to x's length and add."""
    tx, tc = x.shape[1], cond.shape[1]
    assert tc >= tx
    left = (tc - tx) // 2
    return x + cond[:, left:left + tx]


def deconv_stack(mel, weights, hp, prefix='', dtype=np.float32):
    """wavenet.py:23-73,142-155 / parallel_wavenet.py:186-198 (trans_conv or resize_conv branch)."""
    act = upsample_act(hp.get('upsample_act', 'tanh'))
    wn = hp.get('use_weight_norm', False)
    h = np.asarray(mel, dtype)
    for i, (fl, s) in enumerate(hp.deconv_config):
        if hp.get('use_resize_conv', False):
            scope = '{}resize_conv_{:d}'.format(prefix + '/' if prefix else '', i + 1)
            W = get_kernel(weights, scope, 'W', wn, dtype=dtype)
            assert W.shape[1] == fl
            h = resize_conv1d(h, W, np.asarray(weights[scope + '/biases'], dtype), s, act)
            continue
        scope = '{}trans_conv_{:d}'.format(prefix + '/' if prefix else '', i + 1)
        W = get_kernel(weights, scope, 'kernel', wn, deconv=True, dtype=dtype)
        assert W.shape[1] == fl
        b = np.asarray(weights[scope + '/bias'], dtype)
        h = trans_conv1d(h, W, b, s, act)
    return h


def _conv(x, weights, scope, hp, dilation=1, dtype=np.float32):
    W = get_kernel(weights, scope, 'W', hp.get('use_weight_norm', False), dtype=dtype)
    b = np.asarray(weights[scope + '/biases'], dtype)
    return conv1d(x, W, b, dilation)


# --------------------------------------------------------------------------
# IAF student   (parallel_wavenet.py:105-114, 200-345)
# --------------------------------------------------------------------------
def scale_log_scale(p):
    """parallel_wavenet.py:105-114 with USE_LOG_SCALE=False (:14)."""
    s = np.clip(softplus(p), p.dtype.type(EXP_M9), p.dtype.type(EXP_7))
    return s, np.log(s)


def iaf_flow(x, mel_en, weights, hp, iaf_idx, dtype=np.float32, trace=None):
    """parallel_wavenet.py:200-287.  x [B,T,1]; mel_en [B,Tc,Cd] (already chosen)."""
    name = 'iaf_{:d}'.format(iaf_idx + 1)
    L = hp.num_iaf_layers[iaf_idx]
    l = _conv(shift_right(x), weights, name + '/start_conv', hp, dtype=dtype)
    if trace is not None:
        trace.append(('{}/start'.format(name), l.copy()))
    for i in range(L):
        d = _conv(l, weights, '{}/dilated_conv_{:d}'.format(name, i + 1), hp,
                  dilation=2 ** (i % hp.num_stages), dtype=dtype)
        c = _conv(mel_en, weights, '{}/mel_cond_{:d}'.format(name, i + 1), hp, dtype=dtype)
        d = condition(d, c)
        m = d.shape[2] // 2
        g = sigmoid(d[:, :, :m]) * np.tanh(d[:, :, m:])
        l = l + _conv(g, weights, '{}/res_{:d}'.format(name, i + 1), hp, dtype=dtype)
        if trace is not None:
            trace.append(('{}/layer_{:d}'.format(name, i + 1), l.copy()))
    l = np.maximum(l, 0)
    l = _conv(l, weights, name + '/out1', hp, dtype=dtype)
    c = _conv(mel_en, weights, name + '/mel_cond_out1', hp, dtype=dtype)
    l = np.maximum(condition(l, c), 0)
    mean = _conv(l, weights, name + '/out2_mean', hp, dtype=dtype)
    p = _conv(l, weights, name + '/out2_scale', hp, dtype=dtype)
    scale, log_scale = scale_log_scale(p)
    return {'x': x * scale + mean, 'mean': mean, 'scale': scale, 'log_scale': log_scale}


def iaf_length(num_frames, hp):
    """parallel_wavenet.py:293-302."""
    frame_shift = int(np.prod([dc[1] for dc in hp.deconv_config]))
    max_dil = 2 ** (hp.num_stages - 1)
    return (num_frames * frame_shift // max_dil) * max_dil


def iaf_feed_forward(mel, noise, weights, hp, dtype=np.float32, trace=None):
    """parallel_wavenet.py:289-345 with the noise injected ([B,T], logistic or normal)."""
    mel = np.asarray(mel, dtype)
    B, F, _ = mel.shape
    T = iaf_length(F, hp)
    x0 = np.asarray(noise, dtype)
    assert x0.shape == (B, T), (x0.shape, (B, T))
    share = hp.get('use_share_deconv', False) or hp.get('use_teacher_deconv', False)
    assert not (hp.get('use_share_deconv', False) and hp.get('use_teacher_deconv', False))
    mel_en = deconv_stack(mel, weights, hp, 'iaf_share', dtype) if share else None
    x = x0[:, :, None]
    mean_tot = np.zeros_like(x)
    scale_tot = np.ones_like(x)
    log_scale_tot = np.zeros_like(x)
    for k in range(len(hp.num_iaf_layers)):
        en = mel_en if share else deconv_stack(mel, weights, hp, 'iaf_{:d}'.format(k + 1), dtype)
        o = iaf_flow(x, en, weights, hp, k, dtype, trace)
        x = o['x']
        mean_tot = o['mean'] + mean_tot * o['scale']
        scale_tot = scale_tot * o['scale']
        log_scale_tot = log_scale_tot + o['log_scale']
    mean_tot = mean_tot[:, :, 0]
    scale_tot = np.minimum(scale_tot, dtype(EXP_7))[:, :, 0]
    log_scale_tot = np.minimum(log_scale_tot, dtype(7.0))[:, :, 0]
    new_x = x0 * scale_tot + mean_tot
    return {'x': new_x, 'mean_tot': mean_tot, 'scale_tot': scale_tot,
            'log_scale_tot': log_scale_tot, 'rand_input': x0, 'iaf_x': x[:, :, 0]}


def parallelgen(mel, noise, weights, hp, dtype=np.float32):
    """parallelgen.py:11-19: feed_forward + _clip_quant_scale.  Returns (wav, idx, ff)."""
    ff = iaf_feed_forward(mel, noise, weights, hp, dtype)
    wav, idx = clip_quant_scale(ff['x'], quant_chann_of(hp), hp.use_mu_law, dtype)
    return wav, idx, ff


def logistic_from_uniform(u, dtype=np.float32):
    """parallel_wavenet.py:172-178."""
    u = np.asarray(u, dtype)
    return np.log(u) - np.log(dtype(1) - u)


# --------------------------------------------------------------------------
# teacher, full sequence   (wavenet.py:180-291)  -- cross-check of the AR step
# --------------------------------------------------------------------------
def encode_signal(wav, hp, dtype=np.float32):
    """wavenet.py:157-178 ('wav_scaled' only)."""
    wav = np.asarray(wav, dtype)
    if hp.use_mu_law:
        return mu_law(wav, dtype=dtype) / dtype(quant_chann_of(hp) / 2.)
    return wav


def teacher_feed_forward(wav_scaled, mel_en, weights, hp, dtype=np.float32):
    """wavenet.py:219-291 given the deconv output; returns out_params [B,T,out_width]."""
    x = np.asarray(wav_scaled, dtype)[:, :, None]
    mel_en = np.asarray(mel_en, dtype)
    l = _conv(shift_right(x), weights, 'conv_start', hp, dtype=dtype)
    s = _conv(l, weights, 'skip_start', hp, dtype=dtype)
    for i in range(hp.num_layers):
        d = _conv(l, weights, 'dilated_conv_%d' % (i + 1), hp,
                  dilation=2 ** (i % hp.num_stages), dtype=dtype)
        c = _conv(mel_en, weights, 'mel_cond_%d' % (i + 1), hp, dtype=dtype)
        d = condition(d, c)
        m = d.shape[2] // 2
        g = sigmoid(d[:, :, :m]) * np.tanh(d[:, :, m:])
        l = l + _conv(g, weights, 'res_%d' % (i + 1), hp, dtype=dtype)
        s = s + _conv(g, weights, 'skip_%d' % (i + 1), hp, dtype=dtype)
    s = np.maximum(s, 0)
    s = _conv(s, weights, 'out1', hp, dtype=dtype)
    c = _conv(mel_en, weights, 'mel_cond_out1', hp, dtype=dtype)
    s = np.maximum(condition(s, c), 0)
    return _conv(s, weights, 'out2', hp, dtype=dtype)


# --------------------------------------------------------------------------
# teacher scoring: the per-sample terms of Wavenet.calculate_loss   (wavenet.py:157-178,293-316; loss_func.py:8-63,66-75,104-133)
# --------------------------------------------------------------------------
def encode_targets(wav, hp, dtype=np.float32):
    """wavenet.py:157-178: (real_targets, cate_targets) of Wavenet.encode_signal."""
    wav = np.asarray(wav, dtype)
    qc = quant_chann_of(hp)
    if hp.use_mu_law:
        xq = mu_law(wav, dtype=dtype)
        return xq / dtype(qc / 2.), xq.astype(np.int32) + qc // 2
    return wav, cast_quantize(wav, qc, dtype) + qc // 2


def _log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.sum(np.exp(x - m), axis=-1, keepdims=True))      # loss_func.py:8-12


def mol_log_probs(out, targets, quant_chann, dtype=np.float32):
    """loss_func.py:22-63 as written (cdf difference and all), log-scale branch.  out [B,T,3M]; targets [B,T]."""
    out = np.asarray(out, dtype)
    M = out.shape[-1] // 3
    logit, means, ls = out[..., :M], out[..., M:2 * M], np.maximum(out[..., 2 * M:], dtype(-7.0))
    inv = np.exp(-ls)
    t = np.asarray(targets, dtype)[..., None] + np.zeros([1, 1, M], dtype)
    c = t - means
    plus, mn = inv * (c + dtype(1. / quant_chann)), inv * (c - dtype(1. / quant_chann))
    log_cdf_plus = plus - softplus(plus)
    log_one_minus_cdf_min = -softplus(mn)
    delta = sigmoid(plus) - sigmoid(mn)
    max_thres = (float(quant_chann - 1) - 0.5) / (quant_chann / 2.) - 1.0
    min_thres = 0.5 / (quant_chann / 2.) - 1.0
    with np.errstate(divide='ignore'):
        lp = np.where(t < min_thres, log_cdf_plus,
                      np.where(t > max_thres, log_one_minus_cdf_min, np.log(np.maximum(delta, dtype(1e-12)))))
    lp = lp + _log_softmax(logit)
    m = lp.max(axis=-1)
    return m + np.log(np.sum(np.exp(lp - m[..., None]), axis=-1))              # loss_func.py:15-19


def gauss_log_prob(out, targets, dtype=np.float32):
    """loss_func.py:66-75,104-119: Normal(mean, exp(max(p, -7))).log_prob(targets)."""
    out = np.asarray(out, dtype)
    ls = np.maximum(out[..., 1], dtype(-7.0))
    z = (np.asarray(targets, dtype) - out[..., 0]) * np.exp(-ls)
    return -0.5 * z * z - ls - dtype(0.5 * np.log(2.0 * np.pi))


def ce_log_prob(out, cate_targets):
    """loss_func.py:128-133: minus the sparse softmax cross entropy per sample."""
    ls = _log_softmax(np.asarray(out))
    return np.take_along_axis(ls, np.asarray(cate_targets)[..., None].astype(np.int64), axis=-1)[..., 0]


def teacher_log_prob(out, wav, hp, dtype=np.float32):
    """Per-sample log-likelihood of the raw audio `wav` [B,T] under out_params [B,T,ow]: what calculate_loss averages."""
    real, cate = encode_targets(wav, hp, dtype)
    if hp.loss_type == 'mol':
        return mol_log_probs(out, real, quant_chann_of(hp), dtype)
    if hp.loss_type == 'gauss':
        return gauss_log_prob(out, real, dtype)
    return ce_log_prob(np.asarray(out, dtype), cate)


# --------------------------------------------------------------------------
# sampling heads with INJECTED randoms   (loss_func.py:66-75,140-206)
# --------------------------------------------------------------------------
def mol_sample(out, quant_chann, u_sel, u_x, dtype=np.float32):
    """loss_func.py:154-186.  out [B,3*M]; u_sel [B,M], u_x [B] in (1e-5,1-1e-5)."""
    out = np.asarray(out, dtype)
    M = out.shape[1] // 3
    logit, means, log_s = out[:, :M], out[:, M:2 * M], out[:, 2 * M:]
    u_sel = np.asarray(u_sel, dtype)
    k = np.argmax(logit - np.log(-np.log(u_sel)), axis=1)
    r = np.arange(out.shape[0])
    mean = means[r, k]
    ls = np.clip(log_s[r, k], dtype(-7.0), dtype(7.0))
    u_x = np.asarray(u_x, dtype)
    x = mean + np.exp(ls) * (np.log(u_x) - np.log(dtype(1) - u_x))
    x = np.clip(x, dtype(-1.0), dtype(1.0 - 2.0 / quant_chann))
    return cast_quantize(x, quant_chann, dtype)


def gauss_sample(out, quant_chann, z, dtype=np.float32):
    """loss_func.py:66-75,200-206.  out [B,2]; z [B] ~ N(0,1)."""
    out = np.asarray(out, dtype)
    mean = out[:, 0]
    std = np.exp(np.maximum(out[:, 1], dtype(-7.0)))
    x = mean + std * np.asarray(z, dtype)
    x = np.clip(x, dtype(-1.0), dtype(1.0 - 2.0 / quant_chann))
    return cast_quantize(x, quant_chann, dtype)


def ce_sample(out, quant_chann, u, dtype=np.float32):
    """loss_func.py:140-151 (tf Categorical).  TF's RNG cannot be injected; this
    restatement draws by inverse CDF from ONE uniform u [B] in [0,1): the first
    index whose running softmax mass exceeds u*total.  Same distribution."""
    out = np.asarray(out, dtype)
    e = np.exp(out - out.max(axis=1, keepdims=True))
    cdf = np.cumsum(e, axis=1)
    thr = np.asarray(u, dtype) * cdf[:, -1]
    k = np.minimum((cdf <= thr[:, None]).sum(axis=1), out.shape[1] - 1)
    return (k - quant_chann // 2).astype(np.int32)


def sample_margin(out, rnd, hp):
    """Test aid for the integer-parity checks: the sampler of loss_func.py:140-206 evaluated in float64 on
    given network outputs `out` [B, ow] and randoms `rnd` [B, n_rand], returning
      idx     the index float64 arithmetic selects,
      margin  how far the pre-floor() quantity is from the nearest decision boundary -- in index units
              (x*Q/2 against the integers) for mol / gauss, and as a fraction of the total softmax mass
              (u*total against the running sums) for ce,
      gap     mol only: distance between the two largest selection scores (the argmax margin).
    A float32 implementation may differ from `idx` only by one step and only where `margin` is of the
    size of its own rounding error."""
    out = np.asarray(out, np.float64)
    rnd = np.asarray(rnd, np.float64).reshape(out.shape[0], -1)
    qc = quant_chann_of(hp)
    r = np.arange(out.shape[0])
    if hp.loss_type == 'ce':
        e = np.exp(out - out.max(axis=1, keepdims=True))
        cdf = np.cumsum(e, axis=1)
        thr = rnd[:, 0] * cdf[:, -1]
        k = np.minimum((cdf <= thr[:, None]).sum(axis=1), out.shape[1] - 1)
        margin = np.abs(cdf - thr[:, None]).min(axis=1) / cdf[:, -1]
        return (k - qc // 2).astype(np.int64), margin, None
    if hp.loss_type == 'mol':
        M = out.shape[1] // 3
        sel = out[:, :M] - np.log(-np.log(rnd[:, :M]))
        k = np.argmax(sel, axis=1)
        srt = np.sort(sel, axis=1)
        gap = srt[:, -1] - srt[:, -2] if M > 1 else np.full(out.shape[0], np.inf)
        ls = np.clip(out[r, 2 * M + k], -7.0, 7.0)
        u = rnd[:, M]
        x = out[r, M + k] + np.exp(ls) * (np.log(u) - np.log(1.0 - u))
    else:
        gap = None
        x = out[:, 0] + np.exp(np.maximum(out[:, 1], -7.0)) * rnd[:, 0]
    x = np.clip(x, -1.0, 1.0 - 2.0 / qc)
    y = x * (qc / 2.0)
    margin = np.abs(y - np.round(y))
    # a clipped value sits exactly on an integer by construction: it cannot flip
    margin = np.where((x <= -1.0) | (x >= 1.0 - 2.0 / qc), np.inf, margin)
    return np.floor(y).astype(np.int64), margin, gap


# --------------------------------------------------------------------------
# autoregressive step   (wavenet.py:379-514, masked.py:328-405)
# --------------------------------------------------------------------------
class Fastgen(object):
    """Incremental teacher.  Ring r of a causal layer with rate d holds that layer's
    past INPUTS: q1 = x[t-d], q2 = x[t-2d] (masked.py:352-359), zeros initially."""

    def __init__(self, weights, hp, batch_size, dtype=np.float32):
        self.w, self.hp, self.B, self.dtype = weights, hp, batch_size, dtype
        self.width = hp.width
        self.gate = teacher_gate_width(hp)
        self.qc = quant_chann_of(hp)
        self.out_width = teacher_out_width(hp)
        wn = hp.get('use_weight_norm', False)
        self.K = {}
        for scope in (['conv_start', 'skip_start', 'out1', 'mel_cond_out1', 'out2'] +
                      ['%s_%d' % (n, i + 1) for i in range(hp.num_layers)
                       for n in ('dilated_conv', 'mel_cond', 'res', 'skip')]):
            self.K[scope] = (get_kernel(weights, scope, 'W', wn, dtype=dtype),
                             np.asarray(weights[scope + '/biases'], dtype))
        self.reset()

    def reset(self):
        hp, B, dt = self.hp, self.B, self.dtype
        self.t = 0
        self.rates = [1] + [2 ** (i % hp.num_stages) for i in range(hp.num_layers)]
        chans = [1] + [self.width] * hp.num_layers
        # ring[j][slot] with 2*rate slots: value written at step t sits in slot t % (2*rate)
        self.rings = [np.zeros((2 * r, B, c), dt) for r, c in zip(self.rates, chans)]

    def _causal(self, j, scope, x):
        W, b = self.K[scope]
        r = self.rates[j]
        ring = self.rings[j]
        s2 = ring[self.t % (2 * r)].copy()          # x[t-2r]
        s1 = ring[(self.t + r) % (2 * r)].copy()    # x[t-r]
        ring[self.t % (2 * r)] = x                  # push (masked.py:357-359)
        return s2 @ W[0, 0] + s1 @ W[0, 1] + x @ W[0, 2] + b   # masked.py:369-376

    def _lin(self, scope, x):
        W, b = self.K[scope]
        return x @ W[0, 0] + b

    def out_params(self, wav, encoding):
        """One step up to `out` (wavenet.py:408-501).  wav [B,1], encoding [B,Cd]."""
        dt = self.dtype
        x = np.asarray(wav, dt).reshape(self.B, 1)
        en = np.asarray(encoding, dt)
        if self.hp.use_mu_law:
            x = mu_law(x, dtype=dt) / dt(self.qc / 2)
        l = self._causal(0, 'conv_start', x)
        s = self._lin('skip_start', l)
        for i in range(self.hp.num_layers):
            d = self._causal(i + 1, 'dilated_conv_%d' % (i + 1), l)
            d = d + self._lin('mel_cond_%d' % (i + 1), en)
            m = d.shape[1] // 2
            g = sigmoid(d[:, :m]) * np.tanh(d[:, m:])
            l = l + self._lin('res_%d' % (i + 1), g)
            s = s + self._lin('skip_%d' % (i + 1), g)
        s = np.maximum(s, 0)
        s = np.maximum(self._lin('out1', s) + self._lin('mel_cond_out1', en), 0)
        out = self._lin('out2', s)
        self.t += 1
        return out

    def n_rand(self):
        lt = self.hp.loss_type
        return {'mol': self.hp.get('mol_mix', 10) + 1, 'gauss': 1, 'ce': 1}[lt]

    def sample_from(self, out, rnd):
        """rnd [B, n_rand]: mol -> [u_sel(M), u_x]; gauss -> [z]; ce -> [u]."""
        lt = self.hp.loss_type
        rnd = np.asarray(rnd, self.dtype).reshape(self.B, -1)
        if lt == 'mol':
            return mol_sample(out, self.qc, rnd[:, :-1], rnd[:, -1], self.dtype)
        if lt == 'gauss':
            return gauss_sample(out, self.qc, rnd[:, 0], self.dtype)
        return ce_sample(out, self.qc, rnd[:, 0], self.dtype)

    def dequant(self, q):
        """fastgen.py:163-167."""
        if self.hp.use_mu_law:
            return inv_mu_law(q, dtype=self.dtype)
        return inv_cast_quantize(q, self.qc, self.dtype)

    def sample(self, wav, encoding, rnd):
        return self.sample_from(self.out_params(wav, encoding), rnd)


def fastgen_synthesis(encoding, rnd, weights, hp, dtype=np.float32, return_out=False):
    """fastgen.py:128-169 loop.  encoding [B,Tn,Cd]; rnd [Tn,B,n_rand].
    Returns (wav [B,Tn] float, idx [B,Tn] int32[, out_params [B,Tn,ow]])."""
    encoding = np.asarray(encoding, dtype)
    B, Tn, _ = encoding.shape
    fg = Fastgen(weights, hp, B, dtype)
    audio = np.zeros((B, 1), dtype)
    wav = np.zeros((B, Tn), dtype)
    idx = np.zeros((B, Tn), np.int32)
    outs = []
    for t in range(Tn):
        out = fg.out_params(audio, encoding[:, t])
        q = fg.sample_from(out, rnd[t])
        audio = fg.dequant(q).reshape(B, 1)
        wav[:, t] = audio[:, 0]
        idx[:, t] = q
        if return_out:
            outs.append(out)
    if return_out:
        return wav, idx, np.stack(outs, axis=1)
    return wav, idx


def scale_probe_weights(hp):
    """Test aid: student weights (one flow, one layer) that make the flow-head scale parameter of sample t
    equal the flow input of sample t-1 -- so that feeding N(0,1) noise turns `scale_tot` into
    clip(softplus(N(0,1)), e^-9, e^7), the quantity the reference's tests/test_scale.py:94-107 draws
    (get_scale, use_log_scale=False).  start_conv copies +x[t-1] / -x[t-1] into channels 0 / 1; the
    residual layer is zero; out1 is the identity on those channels; out2_scale = ch0 - ch1 = relu(x) - relu(-x)."""
    assert list(hp.num_iaf_layers) == [1]
    w = {k: np.zeros_like(v) for k, v in synth_weights(hp, 'student', seed=0).items()}
    w['iaf_1/start_conv/W'][0, 2, 0, 0] = 1.0       # tap k=2 of shift_right(x) is x[t-1]
    w['iaf_1/start_conv/W'][0, 2, 0, 1] = -1.0
    w['iaf_1/out1/W'][0, 0, 0, 0] = 1.0
    w['iaf_1/out1/W'][0, 0, 1, 1] = 1.0
    w['iaf_1/out2_scale/W'][0, 0, 0, 0] = 1.0
    w['iaf_1/out2_scale/W'][0, 0, 1, 0] = -1.0
    return w


# analytic moments of s = softplus(Z), Z ~ N(0,1) (the e^-9 / e^7 clip is beyond 9 sigma): E s, E s^2
SOFTPLUS_N01_M1 = 0.8060591833474399
SOFTPLUS_N01_M2 = 0.9212459088593004


# --------------------------------------------------------------------------
# synthetic weights (shapes: SURVEY Appendix A; init: masked.py:166-167,
# parallel_wavenet.py:92,274)
# --------------------------------------------------------------------------
def _mk(rng, shape, std):
    return (rng.standard_normal(shape) * std).astype(np.float32)


def synth_weights(hp, kind, seed=1234, init='tf'):
    """kind: 'student' | 'teacher'.  init: 'tf' = N(0,0.05) kernels, zero biases,
    out2_scale bias -0.3; 'unit' = N(0,1/sqrt(fan_in)) kernels, N(0,0.1) biases."""
    rng = np.random.RandomState(seed)
    w = {}
    n_mel = 80
    dw = hp.deconv_width

    def kstd(fan_in):
        return 0.05 if init == 'tf' else 1.0 / np.sqrt(fan_in)

    def bias(n, const=0.0):
        if init == 'tf':
            return np.full([n], const, np.float32)
        return (_mk(rng, [n], 0.1) + np.float32(const)).astype(np.float32)

    def conv(scope, K, cin, cout, bconst=0.0):
        w[scope + '/W'] = _mk(rng, [1, K, cin, cout], kstd(K * cin))
        w[scope + '/biases'] = bias(cout, bconst)

    def deconv(prefix):
        cin = n_mel
        for j, (fl, s) in enumerate(hp.deconv_config):
            if hp.get('use_resize_conv', False):
                scope = '{}resize_conv_{:d}'.format(prefix, j + 1)
                w[scope + '/W'] = _mk(rng, [1, fl, cin, dw], 0.05 if init == 'tf' else 1.0 / (fl * np.sqrt(cin)))
                w[scope + '/biases'] = bias(dw)
                cin = dw
                continue
            scope = '{}trans_conv_{:d}'.format(prefix, j + 1)
            w[scope + '/kernel'] = _mk(rng, [1, fl, dw, cin], kstd(cin * fl / s))
            w[scope + '/bias'] = bias(dw)
            cin = dw

    if kind == 'student':
        W = hp.width
        share = hp.get('use_share_deconv', False) or hp.get('use_teacher_deconv', False)
        if share:
            deconv('iaf_share/')
        for k, L in enumerate(hp.num_iaf_layers):
            p = 'iaf_{:d}'.format(k + 1)
            if not share:
                deconv(p + '/')
            conv(p + '/start_conv', hp.filter_length, 1, W)
            for i in range(L):
                conv('{}/dilated_conv_{:d}'.format(p, i + 1), hp.filter_length, W, W)
                conv('{}/mel_cond_{:d}'.format(p, i + 1), 1, dw, W)
                conv('{}/res_{:d}'.format(p, i + 1), 1, W // 2, W)
            conv(p + '/out1', 1, W, W)
            conv(p + '/mel_cond_out1', 1, dw, W)
            conv(p + '/out2_mean', 1, W, 1)
            conv(p + '/out2_scale', 1, W, 1, bconst=-0.3)
    elif kind == 'teacher':
        W, S = hp.width, hp.skip_width
        G = teacher_gate_width(hp)
        deconv('')
        conv('conv_start', hp.filter_length, 1, W)
        conv('skip_start', 1, W, S)
        for i in range(hp.num_layers):
            conv('dilated_conv_%d' % (i + 1), hp.filter_length, W, G)
            conv('mel_cond_%d' % (i + 1), 1, dw, G)
            conv('res_%d' % (i + 1), 1, G // 2, W)
            conv('skip_%d' % (i + 1), 1, G // 2, S)
        conv('out1', 1, S, S)
        conv('mel_cond_out1', 1, dw, S)
        conv('out2', 1, S, teacher_out_width(hp))
    else:
        raise ValueError(kind)
    return w
