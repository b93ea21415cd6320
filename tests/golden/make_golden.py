"""Generate the restatement-derived golden vectors (NOT TensorFlow-derived; see
oracle/__init__.py "PARITY UNPINNED").  Run from the repo root:
    python tests/golden/make_golden.py
Weights are not stored: they are regenerated from (config, init, seed) by
oracle.wavenet_np.synth_weights, which the tests call with the recorded arguments."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import wavenet_np as O  # noqa: E402


def cfg(name):
    with open(os.path.join(ROOT, 'config_jsons', name)) as f:
        return json.load(f)


SMALL_TEACHER = dict(width=128, skip_width=64, deconv_width=64, num_layers=7, num_stages=3,
                     deconv_config=[[8, 2], [12, 4]])


def iaf_case(tag, cfg_name, init, B, F, noise_kind, extra=None):
    d = cfg(cfg_name)
    d.update(extra or {})
    hp = O.HP(d)
    w = O.synth_weights(hp, 'student', seed=1234, init=init)
    mel = np.random.RandomState(12345).uniform(0, 1, [B, F, 80]).astype(np.float32)
    T = O.iaf_length(F, hp)
    rs = np.random.RandomState(12346)
    if noise_kind == 'logistic':
        noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [B, T]), np.float32)
    else:
        noise = rs.standard_normal([B, T]).astype(np.float32)
    wav, idx, ff = O.parallelgen(mel, noise, w, hp, np.float64)
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), cfg_json=json.dumps(d), init=init, seed=1234,
                        mel=mel, noise=noise, x=ff['x'], mean_tot=ff['mean_tot'], scale_tot=ff['scale_tot'],
                        wav=wav.astype(np.float32), idx=idx)
    print(tag, 'T', T, 'x absmax', np.abs(ff['x']).max())


def ar_case(tag, cfg_name, extra, B, F):
    d = cfg(cfg_name)
    d.update(SMALL_TEACHER)
    d.update(extra or {})
    hp = O.HP(d)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80]).astype(np.float32)
    enc = O.deconv_stack(mel, w, hp, '', np.float64)
    Tn = enc.shape[1]
    fg = O.Fastgen(w, hp, B, np.float32)
    rs = np.random.RandomState(5)
    rnd = (rs.standard_normal([Tn, B, fg.n_rand()]) if hp.loss_type == 'gauss'
           else rs.uniform(1e-5, 1 - 1e-5, [Tn, B, fg.n_rand()])).astype(np.float32)
    forced = np.random.RandomState(3).uniform(-1, 1, [B, Tn]).astype(np.float32)
    out_forced = O.teacher_feed_forward(O.encode_signal(forced, hp, np.float64), enc, w, hp, np.float64)
    wav, idx, outs = O.fastgen_synthesis(enc.astype(np.float32), rnd, w, hp, np.float32, return_out=True)
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), cfg_json=json.dumps(d), init='unit', seed=1234,
                        mel=mel, enc=enc.astype(np.float32), rnd=rnd, forced=forced,
                        out_forced=out_forced, free_idx=idx, free_wav=wav, free_out=outs)
    print(tag, 'Tn', Tn)


def codec_case():
    rs = np.random.RandomState(7)
    x = np.concatenate([rs.uniform(-1.5, 1.5, 4000), np.arange(-32768, 32768, 97) / 32768.0,
                        [-1.0, 1.0, 0.0, -0.0, 1 - 2 / 65536, 1 - 1 / 65536, -1 + 1e-7, 5.0, -5.0,
                         1 - 2 / 256, 0.5, 0.49999997, 2.0 ** -16, -2.0 ** -16, 3e-5, -3e-5]]).astype(np.float32)
    w16, q16 = O.clip_quant_scale(x, 65536, False, np.float32)
    w8, q8 = O.clip_quant_scale(x, 256, True, np.float32)
    np.savez_compressed(os.path.join(HERE, 'codec.npz'), x=x, wav16=w16, idx16=q16, wav8=w8, idx8=q8,
                        inv_mu_table=O.inv_mu_law(np.arange(-128, 128), dtype=np.float32))
    print('codec', x.shape)


if __name__ == '__main__':
    iaf_case('iaf_logistic_tf', 'parallel_wavenet.json', 'tf', 2, 11, 'logistic')
    iaf_case('iaf_logistic_unit', 'parallel_wavenet.json', 'unit', 1, 8, 'logistic')
    iaf_case('iaf_gauss_perflow', 'parallel_wavenet_gauss.json', 'tf', 1, 8, 'gauss')
    iaf_case('iaf_mulaw', 'parallel_wavenet.json', 'tf', 1, 6, 'logistic', {'use_mu_law': True})
    ar_case('ar_mol', 'wavenet_mol.json', None, 3, 6)
    ar_case('ar_ce_mulaw', 'wavenet_ce.json', {'double_gate_width': False}, 2, 5)
    ar_case('ar_gauss', 'wavenet_gauss.json', None, 2, 5)
    codec_case()
