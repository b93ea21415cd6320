"""Host-side mirror of the reference's `ParallelWavenet` for the generation path.

Same constructor argument (the hparams Namespace), same `feed_forward({'mel': ...})`
result keys and `_clip_quant_scale` as wavenet/parallel_wavenet.py:117-141,289-359 of
the reference -- but `feed_forward` is ONE call into the HIP engine instead of a TF
graph.  The reference creates TF variables and a Saver restores them; here
`restore(checkpoint_path)` / `load_weights(dict)` fills the engine.
"""
import numpy as np
import torch

from .. import config as cfg
from ..engine import Engine


class ParallelWavenet(object):
    def __init__(self, hparams, teacher=None, train_path=None, device=None):
        self.hparams = cfg.load_hparams(hparams)
        hp = self.hparams
        self.use_mu_law = hp.use_mu_law
        self.use_weight_norm = getattr(hp, 'use_weight_norm', False)
        self.use_resize_conv = getattr(hp, 'use_resize_conv', False)
        self.upsample_act = getattr(hp, 'upsample_act', 'tanh')
        self.loss_type = getattr(hp, 'loss_type', 'logistic')
        self.use_share_deconv = getattr(hp, 'use_share_deconv', False)
        self.use_teacher_deconv = getattr(hp, 'use_teacher_deconv', False)
        assert not (self.use_share_deconv and self.use_teacher_deconv)
        self.quant_chann = 2 ** 8 if self.use_mu_law else 2 ** 16
        self.out_width = 2
        self.engine = Engine(hp, kind='student', device=device)

    def load_weights(self, weights):
        self.engine.load_weights(weights)
        return self

    def restore(self, checkpoint_path):
        self.engine.load_checkpoint(checkpoint_path)
        return self

    def feed_forward(self, inputs, init=False, noise=None, seed=0):
        """inputs: {'mel': [B,F,80]} (numpy or torch).  Returns device tensors
        x, mean_tot, scale_tot, log_scale_tot, rand_input, each [B,T].
        `noise` injects the logistic/normal draws (the reference draws them inside
        the graph, unseeded); `seed` drives the on-device Philox generator otherwise."""
        out = self.engine.iaf_generate(inputs['mel'], noise=noise, seed=seed,
                                       want=('x', 'mean_tot', 'scale_tot', 'rand_input'))
        # log_scale_tot = min(sum_k log scale_k, 7) == log(min(prod_k scale_k, e^7)): a derived
        # diagnostic the generation path never consumes (only the training losses do).
        out['log_scale_tot'] = torch.log(out['scale_tot'])
        return {k: out[k] for k in ('x', 'mean_tot', 'scale_tot', 'log_scale_tot', 'rand_input')}

    def _clip_quant_scale(self, x, quant_chann=None, use_mu_law=None):
        wav, _ = self.engine.clip_quant(x)
        return wav
