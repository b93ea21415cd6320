"""GPU parity tests of the full-sequence teacher forward (Wavenet.feed_forward) through the C ABI."""
import json
import os

import numpy as np
import pytest

from conftest import load_json

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _np(t):
    return t.detach().cpu().numpy()


def _mel_for(enc_frames, B, seed):
    return np.random.RandomState(seed).uniform(0, 1, [B, enc_frames, 80]).astype(np.float32)


@pytest.mark.parametrize('tag', ['ar_mol', 'ar_ce_mulaw', 'ar_gauss'])
def test_forward_matches_oracle_and_incremental_path(tag):
    """Reduced-width teachers of the committed AR golden cases (MoL / CE+mu-law / Gauss heads):
    wn_teacher_forward == oracle feed_forward == the teacher-forced incremental path (the
    reference's incremental-vs-full identity), with centre-cropped conditioning."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    eng = Engine(cfgd).load_weights(w)
    mel, forced = g['mel'], g['forced']
    B, Tn = forced.shape
    # golden out_forced was produced with the full conditioning (T == F * frame_shift)
    out = _np(eng.teacher_forward(forced, mel))
    assert out.shape == g['out_forced'].shape
    scale = max(1.0, np.abs(g['out_forced']).max())
    assert np.abs(out - g['out_forced']).max() <= 2e-5 * scale
    inc = _np(eng.ar_generate(g['enc'], forced_wav=forced, want_out=True)['out_params'])
    assert np.abs(out - inc).max() <= 2e-5 * scale
    # shorter audio than conditioning: enc is cropped by (F*fs - T)//2 on the left (wavenet.py:76-85)
    md = 2 ** (hp.num_stages - 1)
    T2 = (Tn - 3 * md) // md * md
    enc64 = O.deconv_stack(mel, w, hp, '', np.float64)
    left = (enc64.shape[1] - T2) // 2
    ref2 = O.teacher_feed_forward(O.encode_signal(forced[:, :T2], hp, np.float64), enc64[:, left:left + T2], w, hp,
                                  np.float64)
    out2 = _np(eng.teacher_forward(forced[:, :T2], mel))
    assert np.abs(out2 - ref2).max() <= 2e-5 * max(1.0, np.abs(ref2).max())
    with pytest.raises(ValueError):
        eng.teacher_forward(forced[:, :T2 - 1], mel)                       # not a multiple of the largest dilation
    with pytest.raises(ValueError):
        eng.teacher_forward(np.tile(forced, (1, 2)), mel)                  # longer than the conditioning
    eng.close()


def test_full_width_teacher_forward():
    """wavenet_mol.json as shipped (width 512, gate 1024, skip 256, 30 layers, dilations to 512):
    one 2-frame... 6-frame utterance against the float64 oracle, rows independent, and the
    `Wavenet.feed_forward` mirror."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.wavenet.wavenet import Wavenet
    cfgd = load_json('wavenet_mol.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    net = Wavenet(cfgd).load_weights(w)
    B, F, T = 2, 6, 1024
    rs = np.random.RandomState(0)
    mel = rs.uniform(0, 1, [B, F, 80]).astype(np.float32)
    wav = rs.uniform(-1, 1, [B, T]).astype(np.float32)
    out = _np(net.feed_forward({'wav': wav, 'mel': mel})['out_params'])
    enc = O.deconv_stack(mel, w, hp, '', np.float64)
    left = (enc.shape[1] - T) // 2
    ref = O.teacher_feed_forward(O.encode_signal(wav, hp, np.float64), enc[:, left:left + T], w, hp, np.float64)
    assert out.shape == ref.shape == (B, T, 30)
    assert np.abs(out - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    one = _np(net.feed_forward({'wav': wav[1:], 'mel': mel[1:]})['out_params'])
    assert np.array_equal(one[0], out[1])
    net.engine.close()


def test_forward_with_fp32_upsampler_handle():
    """A teacher handle created with precision='f32' runs its upsampler on the fp32 MFMA; the layer
    GEMMs of the full-sequence forward stay split-fp16.  Same parity bar."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, 'ar_mol.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    w = O.synth_weights(O.HP(cfgd), 'teacher', seed=1234, init='unit')
    eng = Engine(cfgd, precision='f32').load_weights(w)
    out = _np(eng.teacher_forward(g['forced'], g['mel']))
    assert np.abs(out - g['out_forced']).max() <= 2e-5 * max(1.0, np.abs(g['out_forced']).max())
    eng.close()


def test_scoring_picks_the_right_class_on_both_sides_of_every_mu_law_bin_edge():
    """Teacher scoring with a cross-entropy head: the target class is mu_law(wav) + 128 (wavenet.py:157-178), and a one-bin
    slip changes a sample's log-probability by O(1).  Audio is placed 1e-3 of a bin below and above each of the 255 interior
    bin edges -- |x_e| = (256^(k/128) - 1) / 255 -- and at the bin centres: the device must score exactly the class the float64
    definition gives.  (Closer than ~1e-5 of a bin to an edge the float32 `log` of ANY implementation -- this one, numpy's,
    TensorFlow's -- decides the side; that band is not the reference's to pin either.)"""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, 'ar_ce_mulaw.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    assert hp.use_mu_law and hp.loss_type == 'ce'
    eng = Engine(cfgd).load_weights(O.synth_weights(hp, 'teacher', seed=1234, init='unit'))

    def x_of(v):                                     # audio whose mu-law value (before floor) is v, v in (-128, 128)
        return np.sign(v) * (np.exp(np.abs(v) / 128.0 * np.log(256.0)) - 1.0) / 255.0
    k = np.arange(-127, 128, dtype=np.float64)       # the interior edges
    k = k[k != 0]                                    # (sign(x) makes 0 an edge of its own kind: floor(+-0) = 0 both sides)
    v = np.concatenate([k - 1e-3, k + 1e-3, np.arange(-128, 128) + 0.5])
    want = np.floor(v).astype(np.int64) + 128
    wav = x_of(v).astype(np.float32)[None, :]
    assert np.array_equal(O.encode_targets(wav.astype(np.float64), hp, np.float64)[1][0], want)
    rs = np.random.RandomState(5)
    out = rs.normal(0, 3, [1, wav.shape[1], 256]).astype(np.float32)
    lp = _np(eng.teacher_log_prob(out, wav))[0]
    ref_all = out[0].astype(np.float64) - np.log(np.sum(np.exp(out[0].astype(np.float64)), axis=-1, keepdims=True))
    ref = ref_all[np.arange(len(want)), want]
    # the scored class is identifiable: the 256 logits of a sample are distinct to ~1e-2, the tolerance is 1e-4
    got_class = np.argmin(np.abs(ref_all - lp[:, None]), axis=1)
    assert np.array_equal(got_class, want), np.flatnonzero(got_class != want)[:10]
    assert np.abs(lp - ref).max() <= 1e-4
    eng.close()
