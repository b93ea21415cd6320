"""Independent torch-CPU implementation of the IAF generation path (TEST INFRASTRUCTURE).

PARITY: see oracle/__init__.py (held to the reference's own code through oracle/wavenet_np.py; no TensorFlow run).  Two purposes:
  * an implementation of the same math built from torch's own conv primitives
    (F.conv1d with dilation, F.conv_transpose1d) so that oracle/wavenet_np.py is
    checked by something that shares no code with it;
  * the `cpu_baseline` leg of bench.py: it has the reference's op granularity
    (one conv call per masked.conv1d, parallel_wavenet.py:200-345) and runs on all
    host cores through oneDNN/MKL.  It stands in for the reference's TF CPU path,
    which cannot run here (TensorFlow absent).
"""
import numpy as np
import torch
import torch.nn.functional as F

EXP_M9 = float(np.exp(-9.0))
EXP_7 = float(np.exp(7.0))


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


def conv1d(x, W, b, dilation=1):
    """x [B,C,T]; W TF HWIO [1,K,Cin,Cout] (masked.py:160-232)."""
    K = W.shape[1]
    w = W[0].permute(2, 1, 0).contiguous()           # [Cout,Cin,K]
    if K > 1:
        x = F.pad(x, ((K - 1) * dilation, 0))
    return F.conv1d(x, w, b, dilation=dilation)


def trans_conv1d(x, W, b, stride, alpha=0.4):
    """x [B,Cin,L]; W TF [1,K,Cout,Cin] (masked.py:235-291), leaky_relu(alpha)."""
    K = W.shape[1]
    w = W[0].permute(2, 1, 0).contiguous()           # [Cin,Cout,K]
    assert (K - stride) % 2 == 0
    y = F.conv_transpose1d(x, w, b, stride=stride, padding=(K - stride) // 2)
    return F.leaky_relu(y, alpha)


class StudentRef(object):
    def __init__(self, weights, hp, dtype=torch.float32):
        self.hp, self.dtype = hp, dtype
        assert not hp.get('use_weight_norm', False)
        assert hp.get('upsample_act', 'tanh') == 'leaky_relu'
        self.w = {k: _t(v, dtype) for k, v in weights.items()}
        self.share = hp.get('use_share_deconv', False) or hp.get('use_teacher_deconv', False)

    def deconv(self, mel, prefix):
        h = mel.transpose(1, 2)
        for j, (fl, s) in enumerate(self.hp.deconv_config):
            sc = '{}/trans_conv_{:d}'.format(prefix, j + 1)
            h = trans_conv1d(h, self.w[sc + '/kernel'], self.w[sc + '/bias'], s)
        return h                                       # [B,Cd,200F]

    def _c(self, x, scope, dilation=1):
        return conv1d(x, self.w[scope + '/W'], self.w[scope + '/biases'], dilation)

    @torch.no_grad()
    def feed_forward(self, mel, noise):
        hp = self.hp
        mel = _t(mel, self.dtype)
        x0 = _t(noise, self.dtype)
        B, T = x0.shape
        en = self.deconv(mel, 'iaf_share') if self.share else None
        x = x0[:, None, :]
        mean_tot = torch.zeros_like(x)
        scale_tot = torch.ones_like(x)
        for k, L in enumerate(hp.num_iaf_layers):
            p = 'iaf_{:d}'.format(k + 1)
            e = en if self.share else self.deconv(mel, p)
            left = (e.shape[2] - T) // 2
            ec = e[:, :, left:left + T]
            l = self._c(F.pad(x, (1, 0))[:, :, :-1], p + '/start_conv')
            for i in range(L):
                d = self._c(l, '{}/dilated_conv_{:d}'.format(p, i + 1), 2 ** (i % hp.num_stages))
                d = d + self._c(ec, '{}/mel_cond_{:d}'.format(p, i + 1))
                m = d.shape[1] // 2
                g = torch.sigmoid(d[:, :m]) * torch.tanh(d[:, m:])
                l = l + self._c(g, '{}/res_{:d}'.format(p, i + 1))
            l = torch.relu(l)
            l = torch.relu(self._c(l, p + '/out1') + self._c(ec, p + '/mel_cond_out1'))
            mean = self._c(l, p + '/out2_mean')
            scale = torch.clamp(F.softplus(self._c(l, p + '/out2_scale')), EXP_M9, EXP_7)
            x = x * scale + mean
            mean_tot = mean + mean_tot * scale
            scale_tot = scale_tot * scale
        scale_tot = torch.clamp(scale_tot, max=EXP_7)[:, 0]
        mean_tot = mean_tot[:, 0]
        return {'x': (x0 * scale_tot + mean_tot).numpy(), 'mean_tot': mean_tot.numpy(),
                'scale_tot': scale_tot.numpy(), 'iaf_x': x[:, 0].numpy()}

    def parallelgen(self, mel, noise):
        """feed_forward + _clip_quant_scale (non-mu-law, parallel_wavenet.py:347-359)."""
        ff = self.feed_forward(mel, noise)
        Q = 2 ** 8 if self.hp.use_mu_law else 2 ** 16
        assert not self.hp.use_mu_law
        x = np.clip(ff['x'], -1.0, 1.0 - 2.0 / Q)
        q = np.floor(x * (Q / 2)).astype(np.int32)
        return (q / (Q / 2)).astype(np.float32), q, ff
