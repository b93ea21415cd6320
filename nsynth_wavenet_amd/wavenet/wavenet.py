"""Host-side mirror of the reference's `Wavenet.deconv_stack` and `Fastgen`
(wavenet/wavenet.py:94-155,318-514) for the generation path."""
import numpy as np
import torch

from .. import config as cfg
from ..engine import Engine


class _TeacherBase(object):
    def __init__(self, hparams, device=None, engine=None):
        self.hparams = cfg.load_hparams(hparams)
        hp = self.hparams
        self.use_mu_law = hp.use_mu_law
        self.loss_type = hp.loss_type
        self.use_weight_norm = getattr(hp, 'use_weight_norm', False)
        self.double_gate_width = getattr(hp, 'double_gate_width', True)
        self.dropout_inputs = getattr(hp, 'dropout_inputs', False)
        self.dropout_all = getattr(hp, 'dropout_all', False)
        assert not (self.dropout_inputs and self.dropout_all)
        self.quant_chann = cfg.quant_chann(hp)
        self.out_width = cfg.teacher_out_width(hp)
        self.engine = engine if engine is not None else Engine(hp, kind='teacher', device=device)

    def load_weights(self, weights):
        self.engine.load_weights(weights)
        return self

    def restore(self, checkpoint_path):
        self.engine.load_checkpoint(checkpoint_path)
        return self


class Wavenet(_TeacherBase):
    """The teacher's inference graphs: the upsampler and the full-sequence forward."""

    def deconv_stack(self, mel_inputs, init=False):
        return {'encoding': self.engine.deconv(mel_inputs['mel'])}

    def encode_signal(self, inputs):
        """wavenet.py:157-178 on the host (numpy): 'wav' [B,T] -> 'wav_scaled' (the network input), 'real_targets', 'cate_targets'.
        feed_forward / calculate_loss take the raw 'wav' and derive these on the device; this is the reference's function for
        callers that want the targets themselves."""
        from ..auxilaries import utils
        x = np.asarray(inputs['wav'], np.float32)
        q = self.quant_chann
        if self.use_mu_law:
            xq = utils.mu_law_numpy(x, mu=np.float32(255))          # float32 arithmetic, like the TF twin (utils.py:72-87)
            scaled = (xq / np.float32(q / 2.)).astype(np.float32)
            return {'wav_scaled': scaled, 'real_targets': scaled, 'cate_targets': xq.astype(np.int32) + q // 2}
        return {'wav_scaled': x, 'real_targets': x, 'cate_targets': np.floor(x * np.float32(q / 2)).astype(np.int32) + q // 2}

    def feed_forward(self, inputs, init=False):
        """wavenet.py:180-291 with 'wav' [B,T] raw audio and 'mel' [B,F,80]: the reference derives
        'wav_scaled' from 'wav' (wavenet.py:157-178); here the device does.  Returns 'out_params'
        [B,T,out_width] (and 'encoding' on request through deconv_stack: it is not materialised
        in the reference layout by this call)."""
        if init:
            raise ValueError('data-dependent initialisation is a training-time feature')
        # 'wav' is passed through so that the reference's own call sequence (train_wavenet.py:104-108, tests/test_wavenet.py:38-42:
        # ff_dict = feed_forward(inputs); ff_dict.update(encode_signal(inputs)); calculate_loss(ff_dict)) runs unchanged
        return {'out_params': self.engine.teacher_forward(inputs['wav'], inputs['mel']), 'wav': inputs['wav']}

    def calculate_loss(self, ff_dict):
        """wavenet.py:293-316 for scoring audio under the teacher.  ff_dict holds 'out_params' (feed_forward) and the audio: the
        raw 'wav' (feed_forward passes it through; the device derives the targets of encode_signal from it), or -- the reference's
        own keys -- encode_signal's 'real_targets' / 'cate_targets' alone: without mu-law the real target IS the audio; with it
        the class index is handed back as the centre of its bin, inv_mu_law(i), which the device's mu_law maps to the same index
        and the same real target i / 128 (the codec round trip of SURVEY K7, tests/test_ref_codec.py).
        Returns {'loss': -mean log-likelihood, 'log_probs': [B,T]}.  The mixture-of-logistics score is the float64-accurate
        value of the bin mass, not the float32 difference of two sigmoids TensorFlow evaluates (include/wnhip.h)."""
        if 'wav' in ff_dict:
            wav = ff_dict['wav']
        elif self.use_mu_law and 'cate_targets' in ff_dict:
            from ..auxilaries import utils
            cate = ff_dict['cate_targets']
            cate = cate.cpu().numpy() if hasattr(cate, 'cpu') else np.asarray(cate)
            wav = utils.inv_mu_law_numpy(cate.astype(np.int32) - self.quant_chann // 2).astype(np.float32)
        elif not self.use_mu_law and 'real_targets' in ff_dict:
            wav = ff_dict['real_targets']
        else:
            raise KeyError("calculate_loss needs 'wav', or encode_signal's 'real_targets' / 'cate_targets', beside 'out_params'")
        lp = self.engine.teacher_log_prob(ff_dict['out_params'], wav)
        return {'loss': -lp.mean(), 'log_probs': lp}


class Fastgen(_TeacherBase):
    """Incremental teacher: `sample({'wav': [B,1], 'encoding': [B,Cd]})` is one step.
    The two FIFO queues per causal layer are a device-resident ring state created by
    `init()` (the reference's sess.run(init_ops))."""

    def __init__(self, hparams, batch_size=2, device=None, engine=None):
        super(Fastgen, self).__init__(hparams, device, engine)
        self.batch_size = batch_size
        self.state = None

    def init(self):
        self.state = self.engine.ar_new_state(self.batch_size)
        return self

    def cond_vars(self, inputs):
        """wavenet.py:353-377: {'mel_cond_1' .. 'mel_cond_<num_layers>', 'mel_cond_out1'} of inputs['encoding']
        [B, T, deconv_width], each [B, T, gate_width] ([B, T, skip_width] for the last) -- one bulk evaluation on the device."""
        return self.engine.ar_cond_vars(inputs['encoding'])

    def sample(self, inputs, rnd=None, seed=0, want_out=False):
        if self.state is None:
            self.init()
        wav = inputs['wav']
        res = self.engine.ar_step(self.state, wav, inputs['encoding'], rnd=rnd, seed=seed, want_out=want_out)
        if want_out:
            s, out = res
            return {'sample': s.reshape(self.batch_size, 1), 'out_params': out}
        return {'sample': res.reshape(self.batch_size, 1)}
