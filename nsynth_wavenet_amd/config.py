"""Config surface: the reference's config_jsons/*.json read into an
argparse.Namespace exactly like eval_wavenet.py:28-30 does, plus the
per-class `getattr` defaults the reference scatters through its model classes
(wavenet.py:105-129,326-345; parallel_wavenet.py:124-141), gathered in one place.
Training-only keys are accepted and ignored."""
import json
from argparse import Namespace

from . import _lib


def load_hparams(path_or_dict):
    if isinstance(path_or_dict, Namespace):
        return path_or_dict
    if isinstance(path_or_dict, dict):
        return Namespace(**path_or_dict)
    with open(path_or_dict, 'rt') as f:
        return Namespace(**json.load(f))


def model_kind(hparams):
    """'student' (ParallelWavenet) when the JSON has num_iaf_layers, else 'teacher'."""
    return 'student' if hasattr(hparams, 'num_iaf_layers') else 'teacher'


def quant_chann(hparams):
    return 2 ** 8 if hparams.use_mu_law else 2 ** 16     # wavenet.py:117-120


def teacher_gate_width(hparams):
    # double_gate_width defaults to TRUE when the key is absent (wavenet.py:106,329)
    return 2 * hparams.width if getattr(hparams, 'double_gate_width', True) else hparams.width


def teacher_out_width(hparams):
    lt = hparams.loss_type                                # wavenet.py:121-129
    if lt == 'ce':
        return quant_chann(hparams)
    if lt == 'mol':
        return hparams.mol_mix * 3
    if lt == 'gauss':
        return 2
    raise ValueError('[{}] loss is not supported'.format(lt))


def frame_shift(hparams):
    fs = 1
    for _, s in hparams.deconv_config:
        fs *= s
    return fs


def iaf_length(hparams, num_frames):
    """parallel_wavenet.py:293-302."""
    md = 2 ** (hparams.num_stages - 1)
    return (num_frames * frame_shift(hparams) // md) * md


# name -> (wn_config.precision, wn_config.cond_mode)
PRECISIONS = {'f16x3': (0, 0), 'f16x3-fused': (0, 1), 'f16x3-hoisted': (0, 2), 'f32': (1, 0), 'f32-fused': (1, 1), 'f32-hoisted': (1, 2)}


def default_precision():
    """IAF contraction arithmetic: 'f16x3' = split-fp16 operands on the fp16 MFMA (three MFMAs per
    product, ~22-bit operands, fp32 accumulate) -- the default; it hoists the per-layer
    conditioning 1x1s into one GEMM per deconv stack and runs the small-dilation layers two per
    launch ('f16x3-hoisted' names that form explicitly; 'f16x3-fused' evaluates the 1x1s inside
    every layer kernel instead and needs no conditioning workspace); 'f32' = fp32 MFMA, the reference's own
    arithmetic -- since round 6 with the same split: the conditioning 1x1s in one fp32 GEMM per deconv stack and the residual
    layers on the dilated conv alone ('f32-hoisted' names that form, 'f32-fused' the one kernel per layer that reads enc itself)."""
    import os
    return os.environ.get('WN_PRECISION', 'f16x3')


def to_wn_config(hparams, kind=None, n_mel=80, precision=None):
    kind = kind or model_kind(hparams)
    c = _lib.WnConfig()
    precision = precision or default_precision()
    if precision not in PRECISIONS:
        raise ValueError('precision must be one of {}'.format(sorted(PRECISIONS)))
    c.precision, c.cond_mode = PRECISIONS[precision]
    c.use_resize_conv = int(bool(getattr(hparams, 'use_resize_conv', False)))
    dc = hparams.deconv_config
    if len(dc) > _lib.WN_MAX_DECONV:
        raise ValueError('deconv_config has too many layers')
    c.n_mel = n_mel
    c.width = hparams.width
    c.deconv_width = hparams.deconv_width
    c.n_deconv = len(dc)
    for j, (fl, s) in enumerate(dc):
        c.deconv_filter[j] = fl
        c.deconv_stride[j] = s
    c.filter_length = hparams.filter_length
    c.num_stages = hparams.num_stages
    c.use_mu_law = int(bool(hparams.use_mu_law))
    c.use_weight_norm = int(bool(getattr(hparams, 'use_weight_norm', False)))
    c.upsample_act = _lib.ACT[getattr(hparams, 'upsample_act', 'tanh')]   # default 'tanh' (wavenet.py:108)
    if kind == 'student':
        share = bool(getattr(hparams, 'use_share_deconv', False))
        teach = bool(getattr(hparams, 'use_teacher_deconv', False))
        if share and teach:
            raise ValueError('use_share_deconv and use_teacher_deconv are mutually exclusive')  # :135
        lt = getattr(hparams, 'loss_type', 'logistic')
        if lt not in ('logistic', 'gauss'):
            raise ValueError('student loss_type must be logistic or gauss')
        if len(hparams.num_iaf_layers) > _lib.WN_MAX_FLOWS:
            raise ValueError('too many IAF flows')
        c.kind = _lib.KIND_STUDENT
        c.gate_width = hparams.width                       # parallel_wavenet.py:209
        c.n_flows = len(hparams.num_iaf_layers)
        for k, n in enumerate(hparams.num_iaf_layers):
            c.iaf_layers[k] = n
        c.loss_type = _lib.LOSS[lt]
        c.out_width = 2
        c.share_deconv = int(share or teach)
    else:
        c.kind = _lib.KIND_TEACHER
        c.skip_width = hparams.skip_width
        c.gate_width = teacher_gate_width(hparams)
        c.num_layers = hparams.num_layers
        c.loss_type = _lib.LOSS[hparams.loss_type]
        c.mol_mix = getattr(hparams, 'mol_mix', 0) if hparams.loss_type == 'mol' else 0
        c.out_width = teacher_out_width(hparams)
    return c
