"""GPU parity tests of the parallel-WaveNet (IAF) path: the HIP engine, called through
the C ABI, against the float64 oracle on the same seeded inputs and against the
committed golden vectors; size-independent properties at BASELINE.json's full size.

Tolerances (north_star): <= 1e-3 max-abs on the float IAF path (we hold 2e-5 relative to
the signal range); the integer quantisation index is bit-exact for identical float input."""
import json
import os

import numpy as np
import pytest

from conftest import load_json

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _engine(cfgd, weights, precision=None):
    from nsynth_wavenet_amd.engine import Engine
    return Engine(cfgd, precision=precision).load_weights(weights)


def _np(t):
    return t.detach().cpu().numpy()


def test_unknown_forms_are_refused():
    """Only forms a default or a documented precision= can select exist in the library (the round-2 single-launch
    experiments are gone): unknown precision names and conditioning modes are refused, not mapped to something else."""
    from nsynth_wavenet_amd.engine import Engine
    from nsynth_wavenet_amd import config as cfg, _lib
    import ctypes
    for name in ('f16x3-pipe', 'f16x3-resident', 'f16x3-hoisted-resident'):
        with pytest.raises(Exception):
            Engine(load_json('parallel_wavenet.json'), precision=name)
    c = cfg.to_wn_config(cfg.load_hparams(load_json('parallel_wavenet.json')), 'student', 80, 'f16x3')
    c.cond_mode = 3
    h = ctypes.c_void_p(0)
    assert _lib.load().wn_create(ctypes.byref(c), ctypes.byref(h)) == -22


@pytest.mark.parametrize('precision', ['f16x3-fused', 'f16x3-hoisted', 'f32', 'f32-fused'])
@pytest.mark.parametrize('tag', ['iaf_logistic_tf', 'iaf_logistic_unit', 'iaf_gauss_perflow', 'iaf_mulaw'])
def test_golden_vectors(tag, precision):
    """HIP path vs the committed oracle vectors (shared deconv + centre crop 76; unit-gain
    stress weights; ClariNet config with four private deconv stacks; mu-law student), for both
    contraction arithmetics: split-fp16 x3 on the fp16 MFMA (default) and plain fp32 MFMA."""
    from oracle import wavenet_np as O
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    w = O.synth_weights(O.HP(cfgd), 'student', seed=int(g['seed']), init=str(g['init']))
    eng = _engine(cfgd, w, precision)
    assert eng.precision == precision
    out = eng.iaf_generate(g['mel'], g['noise'], want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot', 'rand_input'))
    scale = max(1.0, float(np.abs(g['x']).max()))
    assert np.abs(_np(out['x']) - g['x']).max() <= 2e-5 * scale
    assert np.abs(_np(out['mean_tot']) - g['mean_tot']).max() <= 2e-5 * max(1.0, np.abs(g['mean_tot']).max())
    assert np.abs(_np(out['scale_tot']) - g['scale_tot']).max() <= 2e-5 * max(1.0, np.abs(g['scale_tot']).max())
    assert np.array_equal(_np(out['rand_input']), g['noise'])
    assert np.abs(_np(out['wav']) - g['wav']).max() <= 1e-3            # north-star bound on the audio
    # Integer index vs the golden index, EVERY case: floor() is monotone, so the two may differ by no more
    # than the float difference of the pre-quantisation signal in index units, rounded up -- and where that
    # difference is below one step (TF-init weights: |dx| ~ 2e-7) by at most ONE step, at a boundary.
    Q = 256 if cfgd['use_mu_law'] else 65536
    di = np.abs(_np(out['idx']).astype(np.int64) - g['idx'])
    dx = np.abs(np.clip(_np(out['x']).astype(np.float64), -1, 1 - 2.0 / Q) - np.clip(g['x'].astype(np.float64), -1, 1 - 2.0 / Q))
    assert np.all(di <= np.ceil(dx * (Q / 2)) + (dx > 0))
    y = np.clip(g['x'].astype(np.float64), -1, 1 - 2.0 / Q) * (Q / 2)
    margin = np.abs(y - np.round(y))
    flips = di != 0
    assert np.all(margin[flips] <= dx[flips] * (Q / 2) + 1e-9)        # a flip only where the golden value is that close
    if str(g['init']) == 'tf':
        assert di.max() <= 1 and flips.mean() < 0.02
    print('{} {}: idx differs from the golden index at {} of {} samples (max {} steps, max |dx| {:.2e})'.format(
        tag, precision, int(flips.sum()), di.size, int(di.max()), float(dx.max())))
    # and it is EXACTLY the quantiser applied to the engine's own float output
    wav_o, idx_o = O.clip_quant_scale(_np(out['x']), Q, cfgd['use_mu_law'], np.float32)
    assert np.array_equal(_np(out['idx']), idx_o)
    if not cfgd['use_mu_law']:
        assert np.array_equal(_np(out['wav']), wav_o)
    else:
        assert np.abs(_np(out['wav']) - wav_o).max() <= 2.0 ** -23
    eng.close()


def test_clip_quant_bit_exact():
    """_clip_quant_scale alone: bit-exact int32 index (and exact float for the 16-bit grid)."""
    from oracle import wavenet_np as O
    g = np.load(os.path.join(GOLD, 'codec.npz'))
    w = O.synth_weights(O.HP(load_json('parallel_wavenet.json')), 'student')
    eng = _engine(load_json('parallel_wavenet.json'), w)
    wav, idx = eng.clip_quant(g['x'])
    assert np.array_equal(_np(idx), g['idx16']) and np.array_equal(_np(wav), g['wav16'])
    big = np.random.RandomState(3).uniform(-1.01, 1.01, 1 << 20).astype(np.float32)
    wav, idx = eng.clip_quant(big)
    wo, qo = O.clip_quant_scale(big, 65536, False, np.float32)
    assert np.array_equal(_np(idx), qo) and np.array_equal(_np(wav), wo)
    e0 = eng.clip_quant(np.zeros([0], np.float32))
    assert e0[0].numel() == 0
    eng.close()
    cfgd = dict(load_json('parallel_wavenet.json'), use_mu_law=True)
    eng = _engine(cfgd, w)
    wav, idx = eng.clip_quant(g['x'])
    assert np.array_equal(_np(idx), g['idx8'])
    assert np.abs(_np(wav) - g['wav8']).max() <= 2.0 ** -23
    assert np.all(_np(wav)[g['idx8'] == 0] == 0.0)
    eng.close()


def test_ragged_and_edge_shapes():
    """Batch rows are independent; F too short for one 512 block gives an empty result;
    several (B,F) through one engine; bad shapes raise like the reference's asserts."""
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', init='unit')
    eng = _engine(cfgd, w)
    rs = np.random.RandomState(0)
    mel = rs.uniform(0, 1, [3, 9, 80]).astype(np.float32)
    T = O.iaf_length(9, hp)
    assert T == 1536 and eng.iaf_length(9) == T
    noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [3, T]))
    full = _np(eng.iaf_generate(mel, noise, want=('x',))['x'])
    for b in range(3):
        one = _np(eng.iaf_generate(mel[b:b + 1], noise[b:b + 1], want=('x',))['x'])
        assert np.array_equal(one[0], full[b])                     # bitwise: no cross-utterance term
    ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)['x']
    assert np.abs(full - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    empty = eng.iaf_generate(mel[:, :2], None, want=('wav', 'idx'))
    assert empty['wav'].shape == (3, 0) and empty['idx'].shape == (3, 0)
    with pytest.raises(ValueError):
        eng.iaf_generate(mel[:, :, :40], None)
    with pytest.raises(ValueError):
        eng.iaf_generate(mel, noise[:, :-1])
    eng.close()


def test_full_size_properties_config2():
    """BASELINE configs[1] size (F=384 -> T=76800, batch 1) and the crop variant F=400:
    K2  x == rand_input*scale_tot + mean_tot, scale_tot > 0;  K5  audio on the 2^-15 grid in
    [-1, 1-2^-15];  determinism (bitwise equal double run);  prefix property of causality:
    the first 512k samples do not depend on later mel frames beyond the deconv support."""
    import torch
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    eng = _engine(cfgd, w)
    for F, T, crop in ((384, 76800, 0), (400, 79872, 64)):
        mel = np.random.RandomState(12345).uniform(0, 1, [1, F, 80]).astype(np.float32)
        assert eng.iaf_length(F) == T and (F * 200 - T) // 2 == crop
        noise = O.logistic_from_uniform(np.random.RandomState(12346).uniform(1e-5, 1 - 1e-5, [1, T]))
        a = eng.iaf_generate(mel, noise, want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot', 'rand_input'))
        b = eng.iaf_generate(mel, noise, want=('wav', 'x'))
        assert torch.equal(a['x'], b['x']) and torch.equal(a['wav'], b['wav'])
        x, m, s, r = (_np(a[k]).astype(np.float64) for k in ('x', 'mean_tot', 'scale_tot', 'rand_input'))
        assert np.all(s > 0) and np.all(np.isfinite(x))
        assert np.abs(x - (r * s + m)).max() <= 1e-6 * max(1.0, np.abs(x).max())
        wav = _np(a['wav']).astype(np.float64)
        assert np.all(wav * 32768 == np.round(wav * 32768)) and wav.min() >= -1 and wav.max() <= 1 - 2.0 ** -15
        assert np.array_equal(_np(a['idx']), (wav * 32768).astype(np.int32))
    # prefix property of the causal stack with an aligned crop (crop 0 on both sides): the first samples of
    # the long utterance equal the short utterance built from the same leading mel frames and noise
    F1, F2 = 64, 384                                                   # 64 * 200 = 12800 = 25 * 512
    assert O.iaf_length(F1, hp) == 12800
    mel = np.random.RandomState(5).uniform(0, 1, [1, F2, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(np.random.RandomState(6).uniform(1e-5, 1 - 1e-5, [1, 76800]))
    big = _np(eng.iaf_generate(mel, noise, want=('x',))['x'])
    small = _np(eng.iaf_generate(mel[:, :F1], noise[:, :12800], want=('x',))['x'])
    keep = 12800 - 200 * 3                                             # last frames see truncated mel context
    assert np.abs(big[:, :keep] - small[:, :keep]).max() <= 1e-5
    ref = O.iaf_feed_forward(mel[:, :F1], noise[:, :12800], w, hp, np.float64)['x']
    assert np.abs(small - ref).max() <= 2e-5
    eng.close()


def test_device_noise_statistics_and_seeding():
    """noise == NULL: logistic(0,1) via u ~ U(1e-5, 1-1e-5) (parallel_wavenet.py:172-178) or
    N(0,1) (:180-184), reproducible per seed."""
    import torch
    from oracle import wavenet_np as O
    w = O.synth_weights(O.HP(load_json('parallel_wavenet.json')), 'student')
    eng = _engine(load_json('parallel_wavenet.json'), w)
    mel = np.random.RandomState(1).uniform(0, 1, [2, 200, 80]).astype(np.float32)
    a = eng.iaf_generate(mel, None, seed=7, want=('rand_input', 'x'))
    b = eng.iaf_generate(mel, None, seed=7, want=('rand_input', 'x'))
    c = eng.iaf_generate(mel, None, seed=8, want=('rand_input',))
    assert torch.equal(a['rand_input'], b['rand_input']) and torch.equal(a['x'], b['x'])
    assert not torch.equal(a['rand_input'], c['rand_input'])
    r = _np(a['rand_input']).astype(np.float64)
    assert not np.array_equal(r[0], r[1])
    lim = np.log(1e-5) - np.log(1 - 1e-5)
    assert r.min() >= lim - 1e-3 and r.max() <= -lim + 1e-3
    assert abs(r.mean()) < 0.02 and abs(r.var() - np.pi ** 2 / 3) < 0.1          # logistic(0,1) variance
    eng.close()
    cg = load_json('parallel_wavenet_gauss.json')
    wg = O.synth_weights(O.HP(cg), 'student')
    eng = _engine(cg, wg)
    r = _np(eng.iaf_generate(mel, None, seed=3, want=('rand_input',))['rand_input']).astype(np.float64)
    assert abs(r.mean()) < 0.02 and abs(r.var() - 1.0) < 0.03 and abs((r ** 4).mean() - 3.0) < 0.2
    eng.close()


def test_weight_norm_folding_and_errors():
    """use_weight_norm: W = V/||V||*g folded at load (masked.py:131-157); missing / unknown /
    mis-shaped variables are reported like Saver.restore would."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    from nsynth_wavenet_amd import weights as wts, config as cfg
    cfgd = dict(load_json('parallel_wavenet.json'), use_weight_norm=True, num_iaf_layers=[3, 2])
    hp = cfg.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=9, init='unit')
    eng = Engine(cfgd).load_weights(w)
    rs = np.random.RandomState(0)
    mel = rs.uniform(0, 1, [1, 6, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [1, 1024]))
    ref = O.iaf_feed_forward(mel, noise, w, O.HP(cfgd), np.float64)['x']
    got = _np(eng.iaf_generate(mel, noise, want=('x',))['x'])
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    eng.close()
    eng = Engine(load_json('parallel_wavenet.json'))
    with pytest.raises(KeyError):
        eng.set_weight('iaf_9/not_a_variable/W', np.zeros([1, 1, 1, 1], np.float32))
    with pytest.raises(ValueError):
        eng.set_weight('iaf_1/out1/W', np.zeros([1, 1, 64, 63], np.float32))
    with pytest.raises(KeyError):
        eng.load_weights({})
    with pytest.raises(RuntimeError):
        eng.iaf_generate(mel, None)                  # not finalized
    eng.close()


def test_reference_interface_mirror(tmp_path):
    """parallelgen.synthesis / ParallelWavenet.feed_forward / the CLI write what the
    reference's drivers write: gen_<name>.wav, float32, 16 kHz, T samples on the grid."""
    import subprocess
    import sys
    from scipy.io import wavfile
    from conftest import ROOT
    from nsynth_wavenet_amd import weights as wts, config as cfg
    from nsynth_wavenet_amd.wavenet import parallelgen
    from nsynth_wavenet_amd.wavenet.parallel_wavenet import ParallelWavenet
    cfgd = dict(load_json('parallel_wavenet.json'), num_iters=1)
    hp = cfg.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=1234)
    ck = tmp_path / 'ckpt'
    ck.mkdir()
    path = wts.save_checkpoint(str(ck / 'model.ckpt-7'), w, hp)
    (ck / 'parallel_wavenet.json').write_text(json.dumps(cfgd))
    mel = np.random.RandomState(2).uniform(0, 1, [2, 12, 80]).astype(np.float32)
    names = [str(tmp_path / 'gen_a.wav'), str(tmp_path / 'gen_b.wav')]
    parallelgen.synthesis(hp, mel, names, path)
    for n in names:
        sr, a = wavfile.read(n)
        assert sr == 16000 and a.dtype == np.float32 and a.shape == (2048,)
        assert np.all(a.astype(np.float64) * 32768 == np.round(a.astype(np.float64) * 32768))
    pw = ParallelWavenet(hp).restore(path)
    ff = pw.feed_forward({'mel': mel}, seed=5)
    assert sorted(ff) == ['log_scale_tot', 'mean_tot', 'rand_input', 'scale_tot', 'x']
    x = _np(ff['x']).astype(np.float64)
    assert np.abs(x - (_np(ff['rand_input']) * _np(ff['scale_tot']).astype(np.float64) + _np(ff['mean_tot']))).max() < 1e-5
    assert np.abs(np.exp(_np(ff['log_scale_tot'])) - _np(ff['scale_tot'])).max() < 1e-5
    q = _np(pw._clip_quant_scale(ff['x']))
    assert q.min() >= -1 and q.max() <= 1 - 2.0 ** -15
    # CLI on .npy mels
    src = tmp_path / 'src'
    src.mkdir()
    np.save(str(src / 'utt1.npy'), mel[0])
    np.save(str(src / 'utt2.npy'), mel[1][:9])
    out = tmp_path / 'out'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'eval_parallel_wavenet.py'), '--ckpt_dir', str(ck),
                        '--source_path', str(src), '--save_path', str(out), '--batch_size', '2'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(str(out))) == ['gen_utt1.wav', 'gen_utt2.wav']
    sr, a = wavfile.read(str(out / 'gen_utt1.wav'))
    assert sr == 16000 and a.dtype == np.float32 and a.shape == (2048,)


def test_split_fp16_is_as_accurate_as_fp32():
    """The default arithmetic (split-fp16 operands, 3 fp16 MFMAs per product, fp32 accumulate)
    must not be a precision downgrade: against the float64 oracle its error is within 2x of the
    fp32-MFMA path's error on the same inputs, on both weight sets, at a non-trivial length."""
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    rs = np.random.RandomState(4)
    mel = rs.uniform(0, 1, [1, 21, 80]).astype(np.float32)          # T = 4096, crop 52
    noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [1, 4096]))
    for init in ('tf', 'unit'):
        w = O.synth_weights(hp, 'student', seed=1234, init=init)
        ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)['x']
        err = {}
        for prec in ('f16x3-hoisted', 'f16x3-fused', 'f32'):
            eng = _engine(cfgd, w, prec)
            err[prec] = np.abs(_np(eng.iaf_generate(mel, noise, want=('x',))['x']) - ref).max()
            eng.close()
        scale = max(1.0, np.abs(ref).max())
        assert max(err.values()) <= 2e-5 * scale, (init, err)
        assert err['f16x3-hoisted'] <= 2.0 * err['f32'] + 1e-7 * scale, (init, err)
        assert err['f16x3-fused'] <= 2.0 * err['f32'] + 1e-7 * scale, (init, err)


def test_conditioning_placement_policy():
    """The default 'f16x3' engine hoists the per-layer conditioning 1x1s into one GEMM per deconv
    stack (wn_iaf.hip wn_iaf_hoisted); 'f16x3-fused' keeps them inside the layer kernels.  Both
    forms are the same arithmetic up to fp32 summation order; the forced engine reproduces the
    default bit for bit."""
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    auto, fused, hoisted = (_engine(cfgd, w, p) for p in ('f16x3', 'f16x3-fused', 'f16x3-hoisted'))
    assert auto.iaf_cond_hoisted(1, 384) and auto.iaf_cond_hoisted(8, 384)
    # 96 GB of projected conditioning is the limit of the default form (16 KB per sample here)
    assert auto.iaf_cond_hoisted(64, 384) and not auto.iaf_cond_hoisted(128, 384) and hoisted.iaf_cond_hoisted(128, 384)
    assert not fused.iaf_cond_hoisted(8, 384) and hoisted.iaf_cond_hoisted(1, 8)
    rs = np.random.RandomState(9)
    mel = rs.uniform(0, 1, [3, 80, 80]).astype(np.float32)           # T = 15872 per row
    T = O.iaf_length(80, hp)
    noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [3, T]))
    xa = _np(auto.iaf_generate(mel, noise, want=('x',))['x'])
    xf = _np(fused.iaf_generate(mel, noise, want=('x',))['x'])
    xh = _np(hoisted.iaf_generate(mel, noise, want=('x',))['x'])
    assert np.array_equal(xa, xh)
    assert np.abs(xh - xf).max() <= 2e-6 * max(1.0, np.abs(xf).max())
    for e in (auto, fused, hoisted):
        e.close()


def test_layer_pairs_match_single_layer_launches(monkeypatch):
    """Hoisted form WITHOUT the layer-group kernel (WN_NO_GROUPS=1: the per-layer launches that serve configurations
    the group plan cannot cover): layers with dilations (1,2) and (4,8) run two per launch (wn_iaf_c_pair, layer A's
    output stays in registers, the start conv of a flow runs inside the first pair).  WN_NO_PAIR=1
    launches every layer on its own; both must agree to fp32 rounding of the start conv, on shapes
    that give one run per wave, several runs per row, ragged last runs and several rows."""
    monkeypatch.setenv('WN_NO_GROUPS', '1')
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=4321, init='tf')
    eng = _engine(cfgd, w, 'f16x3')
    rs = np.random.RandomState(77)
    for B, F in ((1, 8), (1, 35), (3, 80), (2, 391), (5, 13)):
        T = O.iaf_length(F, hp)
        mel = rs.uniform(0, 1, [B, F, 80]).astype(np.float32)
        noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [B, T]))
        monkeypatch.delenv('WN_NO_PAIR', raising=False)
        monkeypatch.delenv('WN_NO_HEADFUSE', raising=False)
        a = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'))
        a = {k: _np(v) for k, v in a.items()}
        monkeypatch.setenv('WN_NO_PAIR', '1')
        b = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'))
        b = {k: _np(v) for k, v in b.items()}
        # ... and the flow head as its own launch instead of the last layer's epilogue: same arithmetic
        monkeypatch.setenv('WN_NO_HEADFUSE', '1')
        c = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'))
        c = {k: _np(v) for k, v in c.items()}
        for k in a:
            assert np.isfinite(a[k]).all()
            assert np.abs(a[k] - b[k]).max() <= 2e-6 * max(1.0, np.abs(b[k]).max()), (B, F, k)
            assert np.array_equal(b[k], c[k]), (B, F, k)
    monkeypatch.delenv('WN_NO_PAIR', raising=False)
    monkeypatch.delenv('WN_NO_HEADFUSE', raising=False)
    eng.close()


def test_layer_groups_match_per_layer_launches(monkeypatch):
    """The default hoisted form runs every dilation cycle as TWO launches of the layer-group kernel (wn_iaf_g.hip): a
    natural group (1, 2, 4, 8, 16) with its 62-sample causal halo recomputed per segment, and a decimated group
    (32 .. 512) on the 32 residue classes of time, with the start conv in a flow's first group and the flow head in
    its last.  Held against the per-layer / layer-pair launches on the same engine (wn_iaf_set_groups; the environment
    switches are read once, in wn_create): same arithmetic,
    different summation partners only in the start conv -- on shapes with one segment, ragged last segments, one
    decimated block per residue, several utterances, the centre-crop variant, and private deconv stacks."""
    from oracle import wavenet_np as O
    rs = np.random.RandomState(78)
    for extra, shapes in (({}, ((1, 8), (1, 35), (3, 80), (2, 391), (5, 13), (1, 400), (6, 200))),
                          ({'use_share_deconv': False, 'num_iaf_layers': [10, 20]}, ((2, 30), (1, 77))),
                          ({'num_iaf_layers': [5, 12, 7]}, ((2, 21), (1, 130)))):
        cfgd = dict(load_json('parallel_wavenet.json'), **extra)
        hp = O.HP(cfgd)
        w = O.synth_weights(hp, 'student', seed=4321, init='unit' if extra else 'tf')
        eng = _engine(cfgd, w, 'f16x3')
        for B, F in shapes:
            T = O.iaf_length(F, hp)
            mel = rs.uniform(0, 1, [B, F, 80]).astype(np.float32)
            noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [B, T]))
            eng.set_layer_groups(True)
            assert eng.iaf_layer_groups(B, F)             # ... so that the comparison cannot degenerate into a form against itself
            a = {k: _np(v) for k, v in eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot')).items()}
            a2 = {k: _np(v) for k, v in eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot')).items()}
            eng.set_layer_groups(False)
            assert not eng.iaf_layer_groups(B, F)
            b = {k: _np(v) for k, v in eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot')).items()}
            for k in a:
                assert np.isfinite(a[k]).all(), (extra, B, F, k)
                assert np.array_equal(a[k], a2[k]), (extra, B, F, k)                      # deterministic
                assert np.abs(a[k] - b[k]).max() <= 4e-6 * max(1.0, np.abs(b[k]).max()), (extra, B, F, k)
        with pytest.raises(ValueError):
            eng.set_layer_groups(2)
        eng.close()


def test_phase_group_upsampler_matches_the_phase_major_form(monkeypatch):
    """The last upsampler layer of the split-fp16 path runs as ONE launch (deconv_pg_kernel: four consecutive output phases
    x 32 channels per wave tile, G4 words written from the GEMM epilogue; 2 + 2 phases where a group of four would
    straddle two input frames) instead of phase-major GEMM + interleave (masked.py:235-291, wavenet.py:46-73).  Same
    products, another summation order over the taps: held against the old form (WN_DC_NO_PG=1 at engine creation) at
    fp32-rounding level and against the float64 oracle, on one tile, ragged last tiles (L not a multiple of 640 frames),
    several utterances, the centre-crop shapes, the fused and hoisted forms, tanh activation and private stacks."""
    from oracle import wavenet_np as O
    rs = np.random.RandomState(91)
    for extra, shapes in (({}, ((1, 8), (2, 35), (1, 70), (3, 129), (1, 384))),
                          ({'upsample_act': 'tanh', 'use_share_deconv': False, 'num_iaf_layers': [10, 10]}, ((2, 21), (1, 66)))):
        cfgd = dict(load_json('parallel_wavenet.json'), **extra)
        hp = O.HP(cfgd)
        w = O.synth_weights(hp, 'student', seed=777, init='tf')
        monkeypatch.delenv('WN_DC_NO_PG', raising=False)
        new = {p: _engine(cfgd, w, p) for p in ('f16x3', 'f16x3-fused')}
        monkeypatch.setenv('WN_DC_NO_PG', '1')
        old = _engine(cfgd, w, 'f16x3')
        monkeypatch.delenv('WN_DC_NO_PG', raising=False)
        for B, F in shapes:
            T = O.iaf_length(F, hp)
            mel = rs.uniform(0, 1, [B, F, 80]).astype(np.float32)
            noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [B, T]))
            xo = _np(old.iaf_generate(mel, noise, want=('x',))['x'])
            for p, e in new.items():
                xn = _np(e.iaf_generate(mel, noise, want=('x',))['x'])
                assert np.isfinite(xn).all(), (extra, B, F, p)
                assert np.abs(xn - xo).max() <= 4e-6 * max(1.0, np.abs(xo).max()), (extra, B, F, p)
            if F <= 70:
                ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)['x']
                assert np.abs(xn - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (extra, B, F)
        for e in list(new.values()) + [old]:
            e.close()


def test_fp32_forms_hoisted_against_fused_and_the_frame_axis_upsampler_against_the_phase_major_one(monkeypatch):
    """The reference-precision (fp32-MFMA) arithmetic in its two placements of the conditioning 1x1s -- hoisted into one fp32
    GEMM per deconv stack with the layer kernels on Q4 rows and the flow head in the last layer's epilogue (the default since
    round 6) against one kernel per layer that reads enc itself -- and the upsampler's last layer as a frame-axis GEMM with the
    input tile in LDS (gemm_f32_kernel<4, true>) against the phase-major GEMM of rounds 1-5 (WN_DC_NO_PG=1 at engine
    creation): the same fp32 products summed in another order -> fp32-rounding level against each other, 2e-5 of the range
    against the float64 oracle.  Shapes: one tile, ragged frame counts (the GEMM's last tile, the crop), several utterances,
    a length that is a multiple of 64 but not of 128 samples (the hoisted form steps aside), private stacks (11- and
    31-row-block GEMMs: task ranges that are not whole octets), tanh activation."""
    from oracle import wavenet_np as O
    rs = np.random.RandomState(92)
    for extra, shapes in (({}, ((1, 8), (2, 35), (1, 71), (3, 129), (1, 384))),
                          ({'num_stages': 7, 'num_iaf_layers': [7, 14]}, ((2, 9), (1, 24))),          # T % 128 == 64 at F = 24
                          ({'upsample_act': 'tanh', 'use_share_deconv': False, 'num_iaf_layers': [10, 10, 10, 30]}, ((2, 21), (1, 66)))):
        cfgd = dict(load_json('parallel_wavenet.json'), **extra)
        hp = O.HP(cfgd)
        w = O.synth_weights(hp, 'student', seed=778, init='tf')
        monkeypatch.delenv('WN_DC_NO_PG', raising=False)
        engs = {p: _engine(cfgd, w, p) for p in ('f32', 'f32-fused')}
        monkeypatch.setenv('WN_DC_NO_PG', '1')
        engs['f32, phase-major upsampler'] = _engine(cfgd, w, 'f32')
        monkeypatch.delenv('WN_DC_NO_PG', raising=False)
        for B, F in shapes:
            T = O.iaf_length(F, hp)
            if T == 0:
                continue
            assert engs['f32'].iaf_cond_hoisted(B, F) == (T % 128 == 0) and not engs['f32-fused'].iaf_cond_hoisted(B, F)
            mel = rs.uniform(0, 1, [B, F, 80]).astype(np.float32)
            noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [B, T]))
            out = {p: {k: _np(v) for k, v in e.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot')).items()} for p, e in engs.items()}
            base = out['f32-fused']
            scale = max(1.0, np.abs(base['x']).max())
            for p in ('f32', 'f32, phase-major upsampler'):
                assert np.isfinite(out[p]['x']).all(), (extra, B, F, p)
                for k in ('x', 'mean_tot', 'scale_tot'):
                    assert np.abs(out[p][k] - base[k]).max() <= 6e-6 * scale, (extra, B, F, p, k)
            enc = {p: _np(e.deconv(mel)) for p, e in engs.items() if p != 'f32-fused'}
            assert np.abs(enc['f32'] - enc['f32, phase-major upsampler']).max() <= 2e-6 * max(1.0, np.abs(enc['f32']).max())
            if F <= 71:
                ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
                assert np.abs(out['f32']['x'] - ref['x']).max() <= 2e-5 * max(1.0, np.abs(ref['x']).max()), (extra, B, F)
                enc_ref = O.deconv_stack(mel, w, hp, 'iaf_share' if cfgd.get('use_share_deconv', False) else 'iaf_1', np.float64)
                assert np.abs(enc['f32'] - enc_ref).max() <= 2e-5 * max(1.0, np.abs(enc_ref).max())
        for e in engs.values():
            e.close()


def test_part_timing_aid_accounts_for_the_call_and_leaves_results_alone():
    """wn_profile_parts_*: HIP events where the parts of a generate call begin (bench.py's in-process kernel_us_per_call).  The
    four parts of the default form are all present, non-negative and add up to about the wall time of the calls; the
    recording does not change results; wn_profile_parts_only validates its mask, is refused outside a measurement session and
    is disarmed by wn_profile_parts_end."""
    import time
    import torch
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    w = O.synth_weights(O.HP(cfgd), 'student', seed=1234, init='tf')
    eng = _engine(cfgd, w)
    mel = np.random.RandomState(3).uniform(0, 1, [1, 384, 80]).astype(np.float32)
    ref = eng.iaf_generate(mel, None, seed=5, want=('x',))['x']
    for i in range(3):
        eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
    torch.cuda.synchronize()
    eng.profile_parts_begin()
    t0 = time.perf_counter()
    got = eng.iaf_generate(mel, None, seed=5, want=('x',))['x']
    for i in range(9):
        eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ms, n = eng.profile_parts_end()
    assert n == 10 and sorted(ms) == sorted(eng.PROFILE_PARTS)
    assert all(v >= 0 for v in ms.values()) and ms['cond_gemm'] > 0 and ms['residual_stack'] > ms['upsampler'] > 0
    assert 0.5 * wall_ms <= sum(ms.values()) <= 1.2 * wall_ms
    assert torch.equal(got, ref)
    with pytest.raises(ValueError):
        eng.profile_parts_only(0)
    with pytest.raises(ValueError):
        eng.profile_parts_only(16)
    # a restricted call SKIPS WORK, so the switch exists only inside a measurement session and cannot be left armed:
    # refused outside one, disarmed by parts_end without the caller's help
    with pytest.raises(RuntimeError, match='WN_ESTATE'):
        eng.profile_parts_only(4)
    assert torch.equal(eng.iaf_generate(mel, None, seed=5, want=('x',))['x'], ref)
    eng.profile_parts_begin()
    eng.profile_parts_only(4)                               # conditioning GEMM alone (power measurements)
    eng.iaf_generate(mel, None, seed=5, want=('wav',), check_range=False)
    eng.profile_parts_end()                                 # ... never reset to 15 by the caller
    assert torch.equal(eng.iaf_generate(mel, None, seed=5, want=('x',))['x'], ref)
    eng.close()


@pytest.mark.parametrize('precision', ['f16x3', 'f16x3-fused'])
def test_repeated_calls_are_bit_identical_at_full_size(precision):
    """Six calls on the same inputs at the headline size (one utterance, 384 frames, 76 800 samples: 4 800 blocks x 60
    layers of epilogues per call) give the same bits.  This is the property that exposed the third gfx950 hazard of
    DESIGN.md section 3.7 -- a packed-fp32 operand form that went wrong in a fraction of a percent of the block
    epilogues, differently in every call, while every result stayed within a few 1e-3 of the right one."""
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    eng = _engine(cfgd, w, precision)
    rs = np.random.RandomState(5)
    mel = rs.uniform(0, 1, [1, 384, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [1, O.iaf_length(384, hp)]))
    first = {k: _np(v) for k, v in eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot')).items()}
    assert all(np.isfinite(v).all() for v in first.values())
    for rep in range(5):
        again = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'))
        for k in first:
            assert np.array_equal(first[k], _np(again[k])), (precision, rep, k)
    assert eng.range_fallbacks == 0
    eng.close()


@pytest.mark.parametrize('precision', ['f16x3', 'f32'])
def test_resize_conv_upsampler(precision):
    """use_resize_conv=true (masked.py:294-322; disabled in the shipped JSONs but part of the config
    surface): nearest-neighbour resize + SAME conv runs as the same per-phase GEMM with a summed
    kernel.  Student end to end (shared stack, centre crop) and the private-stack variant."""
    from oracle import wavenet_np as O
    for extra in ({'use_resize_conv': True}, {'use_resize_conv': True, 'use_share_deconv': False,
                                               'num_iaf_layers': [10, 10], 'upsample_act': 'tanh'}):
        cfgd = load_json('parallel_wavenet.json')
        cfgd.update(extra)
        hp = O.HP(cfgd)
        w = O.synth_weights(hp, 'student', seed=7, init='unit')
        eng = _engine(cfgd, w, precision)
        rs = np.random.RandomState(3)
        mel = rs.uniform(0, 1, [2, 11, 80]).astype(np.float32)        # T = 2048, crop 76
        T = O.iaf_length(11, hp)
        noise = O.logistic_from_uniform(rs.uniform(1e-5, 1 - 1e-5, [2, T]))
        ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
        out = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'))
        for k in ('x', 'mean_tot', 'scale_tot'):
            assert np.abs(_np(out[k]) - ref[k]).max() <= 2e-5 * max(1.0, np.abs(ref[k]).max()), (extra, k)
        if 'use_share_deconv' not in extra:
            enc = _np(eng.deconv(mel))
            enc_ref = O.deconv_stack(mel, w, hp, 'iaf_share', np.float64)
            assert enc.shape == enc_ref.shape == (2, 2200, 256)
            assert np.abs(enc - enc_ref).max() <= 1e-5 * max(1.0, np.abs(enc_ref).max())
        eng.close()


def test_device_mel_featuriser_against_analysis():
    """auxilaries/mel_extractor.py on the GPU (wn_mel_spectrogram, the HIP kernel of csrc/wn_mel.hip, through the
    C ABI) against CLOSED FORMS -- two stationary tones
    (the Hann window's DTFT at the bin offsets) and a unit impulse (the window sample, flat over frequency) --
    and against the float64 oracle featuriser (oracle/mel_np.py, explicit DFT) on noise, incl. the reference
    fixture length (154 480 samples -> 773 frames).  The host featuriser has the same test in tests/test_mel.py."""
    import torch
    from nsynth_wavenet_amd.auxilaries import mel_extractor as M
    from oracle import mel_np as OM
    n = 8000
    t = np.arange(n) / 16000.0
    imp = np.zeros(n)
    imp[4000] = 1.0
    rs = np.random.RandomState(0)
    wavs = np.stack([0.5 * np.sin(2 * np.pi * 1000.0 * t + 0.3), 0.25 * np.sin(2 * np.pi * 3437.5 * t + 1.1), imp,
                     rs.uniform(-0.5, 0.5, n), np.zeros(n)]).astype(np.float32)
    dev = M.batch_melspectrogram_device(wavs)
    assert dev.is_cuda and tuple(dev.shape) == (5, 41, 80) and dev.dtype == torch.float32
    got = _np(dev).astype(np.float64)
    for row, (f, a) in enumerate(((1000.0, 0.5), (3437.5, 0.25))):
        want = OM.analytic_tone_mel(f, a)
        near = want >= want.max() - 60.0 / 140.0
        assert np.abs(got[row, 15:26][:, near] - want[None, near]).max() < 2e-4
    assert np.abs(got[2] - OM.analytic_impulse_mel(4000, n)).max() < 5e-5
    assert np.abs(got[3] - OM.melspectrogram(wavs[3])).max() < 2e-4
    assert np.abs(got[4] - 40.0 / 140.0).max() < 1e-6               # silence sits on the floor (1e-5 -> -100 dB)
    long = M.batch_melspectrogram_device(rs.uniform(-0.5, 0.5, [2, 154480]).astype(np.float32))
    assert tuple(long.shape) == (2, 773, 80)
    host = M.batch_melspectrogram(wavs)                              # the product's two implementations agree as well
    assert np.abs(got - host).max() <= 2e-4
    # ragged frame counts (the last workgroup of a row is partly filled), both edges reflect-padded, on the oracle
    for L in (1025, 1799, 3000, 3200):
        y = rs.uniform(-0.5, 0.5, [3, L]).astype(np.float32)
        d = M.batch_melspectrogram_device(y)
        assert tuple(d.shape) == (3, 1 + L // 200, 80)
        for r in range(3):
            assert np.abs(_np(d[r]).astype(np.float64) - OM.melspectrogram(y[r])).max() < 2e-4
    # determinism, torch input on the device, error behaviour of numpy.pad(reflect) for too short signals
    yt = torch.as_tensor(wavs, device='cuda')
    assert torch.equal(M.batch_melspectrogram_device(yt), dev)
    with pytest.raises(ValueError, match='1024'):
        M.batch_melspectrogram_device(np.zeros([1, 1024], np.float32))
    with pytest.raises(ValueError):
        M.batch_melspectrogram_device(np.zeros([800], np.float32))


def test_full_size_batch8_hoisted_conditioning():
    """BASELINE configs[2] per-GPU share (8 utterances of F=384): the call runs the hoisted
    conditioning GEMM; every row must equal the single-utterance (fused-kernel) result up to fp32
    summation order, keep K2 (x == eps*scale_tot + mean_tot) and be reproducible bit for bit."""
    import torch
    from oracle import wavenet_np as O
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    eng = _engine(cfgd, w)
    B, F, T = 8, 384, 76800
    assert eng.iaf_cond_hoisted(B, F)
    mel = np.random.RandomState(12345).uniform(0, 1, [B, F, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(np.random.RandomState(12346).uniform(1e-5, 1 - 1e-5, [B, T]))
    a = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot', 'rand_input', 'wav'))
    b = eng.iaf_generate(mel, noise, want=('x',))
    assert torch.equal(a['x'], b['x'])
    x, m, s, r = (_np(a[k]).astype(np.float64) for k in ('x', 'mean_tot', 'scale_tot', 'rand_input'))
    assert np.all(np.isfinite(x)) and np.all(s > 0)
    assert np.abs(x - (r * s + m)).max() <= 1e-6 * max(1.0, np.abs(x).max())
    for row in (0, 5):
        one = _np(eng.iaf_generate(mel[row:row + 1], noise[row:row + 1], want=('x',))['x'])
        assert np.abs(one[0] - x[row]).max() <= 5e-6 * max(1.0, np.abs(one).max())
    # the call takes the layer-group form by default (round 5: at every size); the per-layer / layer-pair launches -- the
    # fallback of flows the group plan cannot cover -- on the same eight utterances
    assert eng.iaf_layer_groups(B, F)
    eng.set_layer_groups(False)
    assert not eng.iaf_layer_groups(B, F)
    xl = _np(eng.iaf_generate(mel, noise, want=('x',))['x']).astype(np.float64)
    assert np.abs(xl - x).max() <= 5e-6 * max(1.0, np.abs(x).max())
    eng.close()


@pytest.mark.parametrize('precision', ['f16x3', 'f16x3-fused', 'f32'])
def test_flow_head_scale_path_on_test_scale_draws(precision):
    """a6 scale path (parallel_wavenet.py:105-114: softplus -> clip(e^-9, e^7)) at the reference's
    tests/test_scale.py size: 76 800 N(0,1) draws routed into out2_scale by the probe weights
    (oracle.scale_probe_weights), so scale_tot[t] must equal clip(softplus(noise[t-1])) sample by sample --
    incl. a 12x-scaled set that reaches tf.nn.softplus's pass-through / exp branches and the lower clip --
    and its moments must be the closed-form ones of the reference's draw."""
    from oracle import wavenet_np as O
    cfgd = dict(load_json('parallel_wavenet.json'), num_iaf_layers=[1])
    hp = O.HP(cfgd)
    w = O.scale_probe_weights(hp)
    eng = _engine(cfgd, w, precision)
    F = 384
    T = eng.iaf_length(F)
    assert T == 76800
    mel = np.zeros([2, F, 80], np.float32)
    z = np.random.RandomState(94107).standard_normal([2, T]).astype(np.float32)
    z[1] *= 12.0
    out = eng.iaf_generate(mel, z, want=('scale_tot', 'mean_tot', 'x'))
    s = _np(out['scale_tot']).astype(np.float64)
    p = np.concatenate([np.zeros([2, 1]), z[:, :-1].astype(np.float64)], axis=1)
    want = O.scale_log_scale(p)[0]
    assert want[1].min() == np.exp(-9.0) and (p[1] > 13.95).any() and (p[1] < -13.95).any()
    # p itself passes through two split-fp16 1x1s with unit weights: 2^-22 relative; softplus' slope is <= 1
    assert np.abs(s - want).max() <= np.abs(np.maximum(want, 1.0)).max() * 4e-7 + 1e-9
    assert np.all(np.abs(s - want) <= 2e-6 * np.maximum(np.abs(p), 1.0))
    assert np.abs(_np(out['mean_tot'])).max() == 0.0
    assert np.abs(_np(out['x']).astype(np.float64) - z * s).max() <= 1e-6 * np.abs(z * s).max()
    m1, m2 = O.SOFTPLUS_N01_M1, O.SOFTPLUS_N01_M2
    assert abs(s[0].mean() - m1) < 5 * np.sqrt((m2 - m1 * m1) / T)
    assert abs(s[0].std() - np.sqrt(m2 - m1 * m1)) < 0.01
    eng.close()


def _overflow_case(bias):
    """Three one-layer flows with TF-init weights; the scale biases of the first two flows are `bias`, so that their
    scales are clip(softplus(~bias), e^-9, e^7) (parallel_wavenet.py:105-114) and flow 3 sees x ~ noise * scale^2."""
    from oracle import wavenet_np as O
    cfgd = dict(load_json('parallel_wavenet.json'), num_iaf_layers=[1, 1, 1])
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=4321, init='tf')
    for k in (1, 2):
        w['iaf_%d/out2_scale/biases' % k] = np.full([1], bias, np.float32)
    F = 6
    T = O.iaf_length(F, hp)
    mel = np.random.RandomState(3).uniform(0, 1, [2, F, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(np.random.RandomState(4).uniform(1e-5, 1 - 1e-5, [2, T]), np.float32)
    return cfgd, hp, w, mel, noise


def test_fp16_range_is_guarded_never_silent():
    """The split-fp16 forms store activations with the fp16 EXPONENT range.  With scale = e^7 in two flows
    (parallel_wavenet.py:105-114,277) the third flow's start conv sees |x| ~ 1e6 * |noise| and its output passes
    65 504, where the reference's fp32 graph is finite.  Required: (i) the library notices -- status WN_ERANGE, every
    float output NaN, never plausible audio; (ii) the Python engine transparently re-runs the call on the fp32-MFMA
    form and matches the float64 oracle; (iii) a call that stays inside the range (|l| ~ 2e4) is NOT diverted and
    matches the oracle in the split arithmetic."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    # (iii) large but in range: scale ~ 60 per flow -> x ~ 3.6e3 * |noise| <= 4e4, l0 = 0.05 * ... well below 65 504
    cfgd, hp, w, mel, noise = _overflow_case(60.0)
    ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
    assert 1e3 < np.abs(ref['x']).max() < 1e7
    for prec in ('f16x3', 'f16x3-fused'):
        eng = Engine(cfgd, precision=prec).load_weights(w)
        out = eng.iaf_generate(mel, noise, want=('x', 'scale_tot'))
        assert eng.range_fallbacks == 0
        assert np.abs(_np(out['x']) - ref['x']).max() <= 2e-5 * np.abs(ref['x']).max()
        eng.close()
    # (i) + (ii): past the range
    cfgd, hp, w, mel, noise = _overflow_case(1200.0)
    ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
    assert np.isfinite(ref['x']).all() and np.abs(ref['x']).max() > 1e5        # the reference's fp32 graph is finite here
    assert ref['scale_tot'].max() > 0.99 * np.exp(7.0)                          # the e^7 clip is reached (:327 caps the product)
    for prec in ('f16x3', 'f16x3-fused'):
        eng = Engine(cfgd, precision=prec).load_weights(w)
        raw = eng.iaf_generate(mel, noise, want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot'), check_range=False)
        for k in ('wav', 'x', 'mean_tot', 'scale_tot'):
            assert np.isnan(_np(raw[k])).all(), (prec, k)            # poisoned, not plausible
        assert (_np(raw['idx']) == 0).all()
        with pytest.raises(RuntimeError, match='WN_ERANGE'):
            eng.check_range()
        eng.check_range()                                            # asked and reset: nothing pending now
        # a RUN of asynchronous calls (what bench.py times): the status words accumulate in the workspace, ONE question
        # behind the run still finds the overflowing call although an in-range call came after it
        eng.iaf_generate(mel, noise, want=('x',), check_range=False)
        fine = eng.iaf_generate(mel, noise * 0.0, want=('x',), check_range=False)
        assert np.isfinite(_np(fine['x'])).all()
        with pytest.raises(RuntimeError, match='WN_ERANGE'):
            eng.check_range()
        eng.iaf_generate(mel, noise * 0.0, want=('x',), check_range=False)
        eng.check_range()                                            # an in-range run stays silent
        # wn_deconv on the SAME workspace leaves the range-guard words alone (it used to write its channel-major output
        # over them): no spurious WN_ERANGE after an in-range call, and a pending real flag survives it
        eng.iaf_generate(mel, noise * 0.0, want=('x',), check_range=False)
        enc = eng.deconv(mel)
        assert np.isfinite(_np(enc)).all() and np.abs(_np(enc)).max() > 0
        eng.check_range()
        eng.iaf_generate(mel, noise, want=('x',), check_range=False)
        enc2 = eng.deconv(mel)
        assert np.array_equal(_np(enc2), _np(enc))
        with pytest.raises(RuntimeError, match='WN_ERANGE'):
            eng.check_range()
        out = eng.iaf_generate(mel, noise, want=('wav', 'x', 'mean_tot', 'scale_tot'))     # default: guarded
        assert eng.range_fallbacks == 1
        eng.check_range()                                            # handled by the re-run: not reported again
        scale = np.abs(ref['x']).max()
        assert np.isfinite(_np(out['x'])).all()
        assert np.abs(_np(out['x']) - ref['x']).max() <= 2e-5 * scale
        assert np.abs(_np(out['scale_tot']) - ref['scale_tot']).max() <= 2e-5 * ref['scale_tot'].max()
        wav_ref, _ = O.clip_quant_scale(ref['x'], 65536, False, np.float64)
        assert np.abs(_np(out['wav']) - wav_ref).max() <= 1e-3
        # device-drawn noise: the re-run must see the SAME draws (same seed -> same Philox stream)
        a = eng.iaf_generate(mel, None, seed=99, want=('x', 'rand_input'))
        b = O.iaf_feed_forward(mel, _np(a['rand_input']), w, hp, np.float64)
        assert eng.range_fallbacks == 2 and np.abs(_np(a['x']) - b['x']).max() <= 2e-5 * np.abs(b['x']).max()
        # an in-range call on the same engine afterwards is served by the split form again
        small = eng.iaf_generate(mel, noise * 0.0, want=('x',))
        assert eng.range_fallbacks == 2 and np.isfinite(_np(small['x'])).all()
        eng.close()
    # the fp32 form has no such limit and needs no guard
    eng = Engine(cfgd, precision='f32').load_weights(w)
    out = eng.iaf_generate(mel, noise, want=('x',))
    assert eng.range_fallbacks == 0 and np.abs(_np(out['x']) - ref['x']).max() <= 2e-5 * np.abs(ref['x']).max()
    eng.close()


@pytest.mark.parametrize('patch', [
    {'width': 32, 'deconv_width': 128, 'num_stages': 5, 'num_iaf_layers': [6, 7]},
    {'width': 128, 'deconv_width': 256, 'num_iaf_layers': [10, 3], 'use_share_deconv': False},
    {'width': 48, 'deconv_width': 64, 'num_stages': 4, 'num_iaf_layers': [5], 'loss_type': 'gauss',
     'deconv_config': [[8, 2], [12, 4]], 'use_mu_law': True},
])
def test_student_shapes_outside_the_mfma_kernels(patch):
    """masked.conv1d takes any num_filters (masked.py:160-232) and ParallelWavenet any width / deconv_config /
    num_stages (parallel_wavenet.py:124-141,200-287).  The MFMA kernels cover the shipped shape (width 64, deconv
    width 256, ten stages); every other student runs on the generic fp32 kernels (csrc/wn_iaf_x.hip) behind the same
    C ABI and is held to the same float64 oracle: narrow and wide residual stacks, five- and four-stage dilation
    cycles (output length a multiple of 16 / 8, centre crop), private upsamplers, a Gaussian mu-law student with its
    own deconv_config."""
    from oracle import wavenet_np as O
    cfgd = dict(load_json('parallel_wavenet.json'), **patch)
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=99, init='unit')
    eng = _engine(cfgd, w)
    shift = int(np.prod([s for _, s in cfgd['deconv_config']]))
    for B, F in ((1, 3), (3, 17)):
        T = O.iaf_length(F, hp)
        assert T > 0 and T % (2 ** (cfgd['num_stages'] - 1)) == 0 and T <= F * shift
        # the generic kernels evaluate the conditioning inside every layer: the API says so (it used to report 'hoisted')
        assert not eng.iaf_cond_hoisted(B, F) and not eng.iaf_layer_groups(B, F)
        mel = np.random.RandomState(B).uniform(0, 1, [B, F, 80]).astype(np.float32)
        if cfgd.get('loss_type') == 'gauss':
            noise = np.random.RandomState(7).standard_normal([B, T]).astype(np.float32)
        else:
            noise = O.logistic_from_uniform(np.random.RandomState(7).uniform(1e-5, 1 - 1e-5, [B, T]), np.float32)
        ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
        out = eng.iaf_generate(mel, noise, want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot'))
        scale = max(1.0, float(np.abs(ref['x']).max()))
        assert np.abs(_np(out['x']) - ref['x']).max() <= 2e-5 * scale
        assert np.abs(_np(out['mean_tot']) - ref['mean_tot']).max() <= 2e-5 * max(1.0, np.abs(ref['mean_tot']).max())
        assert np.abs(_np(out['scale_tot']) - ref['scale_tot']).max() <= 2e-5 * max(1.0, np.abs(ref['scale_tot']).max())
        Q = 256 if cfgd['use_mu_law'] else 65536
        wav_ref, idx_ref = O.clip_quant_scale(ref['x'], Q, cfgd['use_mu_law'], np.float64)
        assert np.abs(_np(out['idx']).astype(np.int64) - idx_ref).max() <= 1
        # the index is exact for the engine's own float signal
        _, idx_own = O.clip_quant_scale(_np(out['x']), Q, cfgd['use_mu_law'], np.float32)
        assert np.array_equal(_np(out['idx']), idx_own)
        # device-drawn noise works too (K2 identity)
        a = eng.iaf_generate(mel, None, seed=5, want=('x', 'rand_input', 'mean_tot', 'scale_tot'))
        k2 = _np(a['rand_input']).astype(np.float64) * _np(a['scale_tot']) + _np(a['mean_tot'])
        assert np.abs(_np(a['x']) - k2).max() <= 2e-6 * max(1.0, np.abs(k2).max())
    eng.close()
