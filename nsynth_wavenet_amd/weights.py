"""Weight container keyed by the reference's TensorFlow variable names.

The reference restores  '<var>/ExponentialMovingAverage' -> var  for every
trainable variable (wavenet/fastgen.py:12-14), except that with
`use_teacher_deconv` the teacher-owned 'iaf_share/trans_conv_*' variables are
read under their raw names (wavenet/parallelgen.py:29-41).  TensorFlow V2
checkpoint bundles are read by tf_bundle.py (validated only
against its own writer: no TF-written file is available offline); the package's
own on-disk format is an `.npz` holding exactly the same keys.
"""
import glob
import os
import re

import numpy as np

from . import config as cfg

EMA = '/ExponentialMovingAverage'


def _conv(out, hp, scope, K, cin, cout):
    if getattr(hp, 'use_weight_norm', False):      # masked.py:145-153
        out.append((scope + '/W_V', (1, K, cin, cout)))
        out.append((scope + '/W_g', (cout,)))
    else:
        out.append((scope + '/W', (1, K, cin, cout)))
    out.append((scope + '/biases', (cout,)))


def _deconv(out, hp, prefix, n_mel):
    cin = n_mel
    resize = getattr(hp, 'use_resize_conv', False)
    for j, (fl, s) in enumerate(hp.deconv_config):   # wavenet.py:37-44,52-67
        scope = '{}{}_{:d}'.format(prefix + '/' if prefix else '', 'resize_conv' if resize else 'trans_conv', j + 1)
        if resize:                                   # masked.py:294-322: an ordinary conv1d variable pair
            _conv(out, hp, scope, fl, cin, hp.deconv_width)
            cin = hp.deconv_width
            continue
        if getattr(hp, 'use_weight_norm', False):
            out.append((scope + '/kernel_V', (1, fl, hp.deconv_width, cin)))
            out.append((scope + '/kernel_g', (hp.deconv_width,)))
        else:
            out.append((scope + '/kernel', (1, fl, hp.deconv_width, cin)))
        out.append((scope + '/bias', (hp.deconv_width,)))
        cin = hp.deconv_width


def expected_variables(hp, kind=None, n_mel=80):
    """[(tf variable name, shape)] the generation graph of `hp` creates."""
    kind = kind or cfg.model_kind(hp)
    out = []
    fl = hp.filter_length
    if kind == 'student':
        W = hp.width
        share = getattr(hp, 'use_share_deconv', False) or getattr(hp, 'use_teacher_deconv', False)
        if share:
            _deconv(out, hp, 'iaf_share', n_mel)      # parallel_wavenet.py:311-314
        for k, L in enumerate(hp.num_iaf_layers):
            p = 'iaf_{:d}'.format(k + 1)
            if not share:
                _deconv(out, hp, p, n_mel)            # :217-220
            _conv(out, hp, p + '/start_conv', fl, 1, W)
            for i in range(L):
                _conv(out, hp, '{}/dilated_conv_{:d}'.format(p, i + 1), fl, W, W)
                _conv(out, hp, '{}/mel_cond_{:d}'.format(p, i + 1), 1, hp.deconv_width, W)
                _conv(out, hp, '{}/res_{:d}'.format(p, i + 1), 1, W // 2, W)
            _conv(out, hp, p + '/out1', 1, W, W)
            _conv(out, hp, p + '/mel_cond_out1', 1, hp.deconv_width, W)
            _conv(out, hp, p + '/out2_mean', 1, W, 1)
            _conv(out, hp, p + '/out2_scale', 1, W, 1)
    else:
        W, S, G = hp.width, hp.skip_width, cfg.teacher_gate_width(hp)
        _deconv(out, hp, '', n_mel)
        _conv(out, hp, 'conv_start', fl, 1, W)
        _conv(out, hp, 'skip_start', 1, W, S)
        for i in range(hp.num_layers):
            _conv(out, hp, 'dilated_conv_%d' % (i + 1), fl, W, G)
            _conv(out, hp, 'mel_cond_%d' % (i + 1), 1, hp.deconv_width, G)
            _conv(out, hp, 'res_%d' % (i + 1), 1, G // 2, W)
            _conv(out, hp, 'skip_%d' % (i + 1), 1, G // 2, S)
        _conv(out, hp, 'out1', 1, S, S)
        _conv(out, hp, 'mel_cond_out1', 1, hp.deconv_width, S)
        _conv(out, hp, 'out2', 1, S, cfg.teacher_out_width(hp))
    return out


def synthetic_weights(hp, kind=None, seed=1234, init='tf', n_mel=80):
    """Random-init weights of the architecture `hp` describes (no checkpoints can be
    downloaded here).  init='tf': the reference's initialisers -- kernels N(0,0.05)
    (masked.py:166), biases 0 (masked.py:167), out2_scale bias -0.3
    (parallel_wavenet.py:92,274).  init='unit': N(0,1/sqrt(fan_in)) kernels and
    N(0,0.1) biases, which keeps activations O(1) through the whole stack."""
    kind = kind or cfg.model_kind(hp)
    rng = np.random.RandomState(seed)
    w = {}
    for name, shape in expected_variables(hp, kind, n_mel):
        leaf = name.rsplit('/', 1)[1]
        if leaf in ('W', 'kernel', 'W_V', 'kernel_V'):
            if init == 'tf':
                std = 0.05
            elif leaf.startswith('kernel'):
                stride = dict((fl, s) for fl, s in hp.deconv_config)[shape[1]]
                std = 1.0 / np.sqrt(shape[3] * shape[1] / stride)
            elif 'resize_conv' in name:               # the resized input repeats each frame `stride` times
                std = 1.0 / (shape[1] * np.sqrt(shape[2]))
            else:
                std = 1.0 / np.sqrt(shape[1] * shape[2])
            w[name] = (rng.standard_normal(shape) * std).astype(np.float32)
        elif leaf in ('W_g', 'kernel_g'):
            w[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            const = -0.3 if name.endswith('out2_scale/biases') else 0.0
            if init == 'tf':
                w[name] = np.full(shape, const, np.float32)
            else:
                w[name] = (rng.standard_normal(shape) * 0.1 + const).astype(np.float32)
    return w


def raw_name_variables(names, hp):
    """Variables a generation checkpoint stores under their RAW name instead of '<name>/ExponentialMovingAverage':
    the teacher-owned upsampler of a student built with use_teacher_deconv (the reference's restore map looks
    those up by raw name, parallelgen.py:31-39) -- transposed-conv and resize-conv flavours alike."""
    if hp is None or not getattr(hp, 'use_teacher_deconv', False):
        return set()
    return {k for k in names if k.startswith(('iaf_share/trans_conv', 'iaf_share/resize_conv'))}


def checkpoint_keys(weights, hp=None, ema=True):
    """name -> checkpoint key, the convention the reference's Saver maps use (fastgen.py:12-14)."""
    raw = raw_name_variables(weights, hp)
    return {k: (k if (not ema or k in raw) else k + EMA) for k in weights}


def save_checkpoint(path, weights, hp=None, ema=True):
    """Write an .npz with the key convention the reference's Saver maps use."""
    keys = checkpoint_keys(weights, hp, ema)
    out = {}
    for k, v in weights.items():
        out[keys[k]] = np.asarray(v, np.float32)
    np.savez(path, **out)
    return path if path.endswith('.npz') else path + '.npz'


class _BundleView(object):
    """np.load-like view (files, []) over a TensorFlow V2 checkpoint prefix."""

    def __init__(self, prefix):
        from . import tf_bundle
        self._r = tf_bundle.BundleReader(prefix)
        self.files = list(self._r.entries)

    def __getitem__(self, key):
        return self._r.get_tensor(key)


def load_checkpoint(path, hp, kind=None):
    """name -> float32 array for every variable of `hp`; EMA shadow preferred.  `path` is either
    an .npz written by save_checkpoint or a TensorFlow V2 checkpoint prefix
    (`model.ckpt-N` with `.index` / `.data-00000-of-00001` next to it, read by tf_bundle)."""
    if os.path.exists(path + '.index'):
        blob = _BundleView(path)
    else:
        if not path.endswith('.npz') and os.path.exists(path + '.npz'):
            path = path + '.npz'
        blob = np.load(path)
    out = {}
    missing = []
    for name, shape in expected_variables(hp, kind):
        key = name + EMA if name + EMA in blob.files else name
        if key not in blob.files:
            missing.append(name)
            continue
        arr = np.asarray(blob[key], np.float32)
        if int(np.prod(arr.shape)) != int(np.prod(shape)):
            raise ValueError('checkpoint tensor {} has shape {} but the model needs {}'.format(
                key, arr.shape, shape))
        out[name] = arr.reshape(shape)          # Saver(reshape=True), parallelgen.py:40
    if missing:
        raise KeyError('checkpoint {} lacks {} variables, e.g. {}'.format(path, len(missing), missing[:4]))
    return out


def latest_checkpoint(ckpt_dir):
    """tf.train.latest_checkpoint look-alike (eval_wavenet.py:21-23): honours a
    `checkpoint` state file ('model_checkpoint_path: "..."', run_all_eval.py:44-49),
    else the highest-numbered model.ckpt-N.npz."""
    state = os.path.join(ckpt_dir, 'checkpoint')
    if os.path.exists(state):
        with open(state, 'rt') as f:
            m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', f.read())
        if m:
            p = m.group(1)
            p = p if os.path.isabs(p) else os.path.join(ckpt_dir, p)
            if os.path.exists(p) or os.path.exists(p + '.npz') or os.path.exists(p + '.index'):
                return p
    best, best_n = None, -1
    for p in glob.glob(os.path.join(ckpt_dir, 'model.ckpt-*.npz')) + glob.glob(os.path.join(ckpt_dir, 'model.ckpt-*.index')):
        m = re.search(r'model\.ckpt-(\d+)\.(npz|index)$', p)      # run_all_eval.py:36-49 uses the same pattern
        if m and int(m.group(1)) > best_n:
            best_n = int(m.group(1))
            best = p if p.endswith('.npz') else p[:-len('.index')]
    if best is None:
        cands = sorted(glob.glob(os.path.join(ckpt_dir, '*.npz')))
        best = cands[-1] if cands else None
    return best
