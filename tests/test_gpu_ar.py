"""GPU parity tests of the autoregressive (fastgen) path through the C ABI."""
import json
import os

import numpy as np
import pytest

from conftest import load_json

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('tag', ['ar_mol', 'ar_ce_mulaw', 'ar_gauss'])
def test_golden_ar(tag):
    """deconv (fastgen.encode), K1 (teacher-forced step outputs == full-sequence teacher),
    free-running loop vs the oracle loop, sampler exactness on the engine's own logits,
    and the single-step API with explicit queue state (tests/test_fastgen.py shape)."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    eng = Engine(cfgd).load_weights(w)
    B, Tn = g['forced'].shape
    enc = _np(eng.deconv(g['mel']))
    assert enc.shape == g['enc'].shape and np.abs(enc - g['enc']).max() <= 1e-5
    assert eng.ar_length(g['mel'].shape[1]) == Tn
    # K1
    out = eng.ar_generate(g['enc'], g['rnd'], forced_wav=g['forced'], want_out=True)
    assert np.abs(_np(out['out_params']) - g['out_forced']).max() <= 2e-5 * max(1.0, np.abs(g['out_forced']).max())
    # free run
    out = eng.ar_generate(g['enc'], g['rnd'], want_out=True)
    gi = _np(out['idx'])
    Q = 256 if cfgd['use_mu_law'] else 65536
    assert gi.min() >= -Q // 2 and gi.max() < Q // 2
    fg = O.Fastgen(w, hp, B, np.float32)
    # Integer parity of the sampling head (loss_func.py:140-206) GIVEN the engine's own network output and
    # the injected randoms: against float64 arithmetic the index may differ by ONE step, and only where
    # the pre-floor() quantity sits within float32 rounding of a decision boundary.
    gop = _np(out['out_params'])
    mol = hp.loss_type == 'mol'
    n_bad = n_flip = 0
    for t in range(Tn):
        i64, margin, gap = O.sample_margin(gop[:, t], g['rnd'][t], hp)
        d = gi[:, t].astype(np.int64) - i64
        flip = d != 0
        n_flip += int(flip.sum())
        if hp.loss_type == 'ce':
            ok = (np.abs(d) <= 1) & (~flip | (margin <= 2e-5))     # 256 fp32 running sums: ~1.5e-5 of the mass
        else:
            # |x| <= 1 in fp32 -> x*Q/2 carries <= ~4e-3 index units of rounding at Q = 65536 (libm exp/log 1-2 ulp)
            tol = 0.02 if Q == 65536 else 1e-3
            ok = (np.abs(d) <= 1) & (~flip | (margin <= tol))
            if mol:     # a different mixture component only where the two best scores tie within fp32 noise
                ok |= gap <= 1e-5
        n_bad += int((~ok).sum())
    print('{}: sampler index vs float64: {} of {} differ by one step at a boundary, {} unexplained'.format(
        tag, n_flip, B * Tn, n_bad))
    assert n_bad == 0
    assert n_flip <= max(2, B * Tn // 20)
    # the float32 oracle sampler on the same inputs agrees except at those boundary cases as well
    mism = sum(int((fg.sample_from(gop[:, t], g['rnd'][t]) != gi[:, t]).sum()) for t in range(Tn))
    assert mism <= n_flip + max(2, B * Tn // 20)
    # ... the fed-back audio is the de-quantised index ...
    assert np.abs(_np(out['wav']) - fg.dequant(gi)).max() <= 2.0 ** -23
    # ... and the free-running index stream IS the oracle loop's until the first step at which the engine's
    # own pre-floor quantity sits at a decision boundary (there float noise may fork the two loops by one step)
    diff = gi != g['free_idx']
    first = int(np.argwhere(diff)[:, 1].min()) if diff.any() else Tn
    if first < Tn:
        assert not diff[:, :first].any()
        rows = np.where(diff[:, first])[0]
        _, margin, gap = O.sample_margin(gop[:, first], g['rnd'][first], hp)
        dd = np.abs(gi[rows, first].astype(np.int64) - g['free_idx'][rows, first])
        lim = 2e-4 if hp.loss_type == 'ce' else (0.25 if Q == 65536 else 5e-3)
        near = margin[rows] <= lim
        if mol:
            near |= gap[rows] <= 1e-4
        assert np.all(near), (first, margin[rows], dd)
        assert np.all(dd[margin[rows] <= lim] <= 1)
    else:
        assert np.abs(_np(out['wav']) - g['free_wav']).max() <= 2.0 ** -23
    print('{}: free run identical to the oracle loop for {} of {} steps'.format(tag, first, Tn))
    # single-step API, reference test shape: wav [B,1], encoding [B,Cd]
    st = eng.ar_new_state(B)
    fg2 = O.Fastgen(w, hp, B, np.float32)
    a = np.zeros([B], np.float32)
    for t in range(12):
        s, op = eng.ar_step(st, a, g['enc'][:, t], g['rnd'][t], want_out=True)
        op_o = fg2.out_params(a.reshape(B, 1), g['enc'][:, t])
        assert np.abs(_np(op) - op_o).max() <= 2e-5 * max(1.0, np.abs(op_o).max())
        a = fg2.dequant(_np(s)).astype(np.float32)
    eng.close()


def test_full_width_teacher_short_run():
    """wavenet_mol.json as shipped (width 512, 30 layers, MoL-10): K1 on a short prefix,
    Philox sampling reproducible per seed, batch rows independent."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    cfgd = load_json('wavenet_mol.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    eng = Engine(cfgd).load_weights(w)
    B, Tn = 2, 512          # the full-sequence teacher needs T % 2^(num_stages-1) == 0 (masked.py:188)
    rs = np.random.RandomState(0)
    enc = (rs.standard_normal([B, Tn, 256]) * 0.3).astype(np.float32)
    forced = rs.uniform(-1, 1, [B, Tn]).astype(np.float32)
    out = eng.ar_generate(enc, None, seed=1, forced_wav=forced, want_out=True)
    ref = O.teacher_feed_forward(forced.astype(np.float64), enc.astype(np.float64), w, hp, np.float64)
    assert np.abs(_np(out['out_params']) - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    a = eng.ar_generate(enc, None, seed=5)
    b = eng.ar_generate(enc, None, seed=5)
    c = eng.ar_generate(enc[:1], None, seed=5)
    assert torch.equal(a['idx'], b['idx']) and torch.equal(a['idx'][:1], c['idx'])
    assert eng.ar_n_rand() == 11
    eng.close()


def test_fastgen_mirror_and_cli(tmp_path):
    """fastgen.encode_mel/synthesis and eval_wavenet.py write F*200 samples per utterance."""
    import subprocess
    import sys
    from scipy.io import wavfile
    from conftest import ROOT
    from nsynth_wavenet_amd import weights as wts, config as cfg
    from nsynth_wavenet_amd.wavenet.wavenet import Fastgen
    cfgd = dict(load_json('wavenet_gauss.json'), width=128, skip_width=64, num_layers=4, num_stages=2)
    hp = cfg.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=3)
    ck = tmp_path / 'ckpt'
    ck.mkdir()
    wts.save_checkpoint(str(ck / 'model.ckpt-1'), w, hp)
    (ck / 'wavenet_gauss.json').write_text(json.dumps(cfgd))
    src = tmp_path / 'src'
    src.mkdir()
    np.save(str(src / 'u.npy'), np.random.RandomState(0).uniform(0, 1, [3, 80]).astype(np.float32))
    out = tmp_path / 'out'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'eval_wavenet.py'), '--ckpt_dir', str(ck),
                        '--source_path', str(src), '--save_path', str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sr, a = wavfile.read(str(out / 'gen_u.wav'))
    assert sr == 16000 and a.dtype == np.float32 and a.shape == (600,)
    assert np.all(a.astype(np.float64) * 32768 == np.round(a.astype(np.float64) * 32768))
    fg = Fastgen(hp, batch_size=4).load_weights(w).init()
    np.random.seed(12345)                                   # tests/test_fastgen.py:28-32
    s = fg.sample({'wav': np.random.uniform(-1, 1, [4, 1]), 'encoding': np.random.uniform(-1, 1, [4, 256])},
                  rnd=np.zeros([4, 1], np.float32))
    assert s['sample'].shape == (4, 1) and str(s['sample'].dtype) == 'torch.int32'


@pytest.mark.parametrize('B', [5, 20, 40])
def test_batched_mfma_step_matches_gemv_step_and_oracle(B, monkeypatch):
    """B >= 4 runs the batched step (batch = MFMA N dimension, K-split slabs); it must agree
    with the GEMV step (forced through WN_AR_MODE) and with the full-sequence teacher (K1)."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, 'ar_mol.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    rs = np.random.RandomState(B)
    Tn = 24
    enc = (rs.standard_normal([B, Tn, hp.deconv_width]) * 0.5).astype(np.float32)
    forced = rs.uniform(-1, 1, [B, Tn]).astype(np.float32)
    rnd = rs.uniform(1e-5, 1 - 1e-5, [Tn, B, 11]).astype(np.float32)
    ref = O.teacher_feed_forward(O.encode_signal(forced, hp, np.float64), np.pad(enc, ((0, 0), (0, 0), (0, 0))).astype(np.float64),
                                 w, hp, np.float64) if Tn % 4 == 0 else None
    outs = {}
    for mode in ('mfma', 'gemv'):
        monkeypatch.setenv('WN_AR_MODE', mode)
        eng = Engine(cfgd).load_weights(w)
        a = eng.ar_generate(enc, rnd, forced_wav=forced, want_out=True)
        b = eng.ar_generate(enc, rnd, want_out=True)
        outs[mode] = (_np(a['out_params']), _np(b['idx']), _np(b['out_params']))
        eng.close()
    assert np.abs(outs['mfma'][0] - outs['gemv'][0]).max() <= 2e-5 * max(1.0, np.abs(outs['gemv'][0]).max())
    if ref is not None:
        assert np.abs(outs['mfma'][0] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    agree = (outs['mfma'][1] == outs['gemv'][1]).mean()
    assert agree > 0.9                       # free-running streams fork only where float noise crosses floor()


def test_teacher_resize_conv_encoding():
    """fastgen.encode path with use_resize_conv=true on the teacher (variables resize_conv_i/{W,biases},
    odd filter lengths allowed): wn_deconv vs the oracle, then a short teacher-forced run."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    cfgd = load_json('wavenet_mol.json')
    cfgd.update(dict(width=128, skip_width=64, deconv_width=64, num_layers=4, num_stages=2,
                     deconv_config=[[7, 2], [12, 4]], use_resize_conv=True))
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=5, init='unit')
    assert 'resize_conv_1/W' in w
    eng = Engine(cfgd).load_weights(w)
    mel = np.random.RandomState(2).uniform(0, 1, [2, 9, 80]).astype(np.float32)
    enc = _np(eng.deconv(mel))
    enc_ref = O.deconv_stack(mel, w, hp, '', np.float64)
    assert enc.shape == enc_ref.shape == (2, 72, 64)
    assert np.abs(enc - enc_ref).max() <= 1e-5 * max(1.0, np.abs(enc_ref).max())
    forced = np.random.RandomState(3).uniform(-1, 1, [2, 72]).astype(np.float32)
    ref = O.teacher_feed_forward(O.encode_signal(forced, hp, np.float64), enc_ref, w, hp, np.float64)
    res = eng.ar_generate(enc, forced_wav=forced, want_out=True)
    assert np.abs(_np(res['out_params']) - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    eng.close()


@pytest.mark.parametrize('B', [1, 5])
def test_graph_replay_matches_plain_launches_and_inputs_are_checked(B):
    """wn_ar_generate replays the step from hipGraphs (16 steps per graph + single-step graphs for the rest) when
    the stream can be captured -- Engine.ar_generate runs it on a side stream for that reason -- and issues plain
    launches otherwise (use_graph=False -> wn_ar_set_graph(0)).  Same kernels, same order: bitwise equal results,
    teacher-forced and free-running, for the GEMV step (B < 4) and the batched MFMA step.  Mis-shaped inputs
    raise before anything reaches the device."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, 'ar_mol.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    eng = Engine(cfgd).load_weights(w)
    rs = np.random.RandomState(B)
    Tn = 37                                                   # 2 x 16-step graphs + 5 single-step replays
    enc = (rs.standard_normal([B, Tn, hp.deconv_width]) * 0.5).astype(np.float32)
    forced = rs.uniform(-1, 1, [B, Tn]).astype(np.float32)
    rnd = rs.uniform(1e-5, 1 - 1e-5, [Tn, B, eng.ar_n_rand()]).astype(np.float32)
    for kw in (dict(forced_wav=forced), dict()):
        a = eng.ar_generate(enc, rnd, want_out=True, use_graph=True, **kw)
        b = eng.ar_generate(enc, rnd, want_out=True, use_graph=False, **kw)
        c = eng.ar_generate(enc, rnd, want_out=True, use_graph=True, **kw)      # a second capture on the same handle
        for k in ('idx', 'wav', 'out_params'):
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
    # the caller's stream is usable afterwards and sees the results (the side stream was joined)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d = eng.ar_generate(enc, rnd, want_out=True)
        assert torch.equal(d['idx'], a['idx'])
    with pytest.raises(ValueError):
        eng.ar_generate(enc, np.transpose(rnd, (1, 0, 2)))                       # [B,Tn,n] instead of [Tn,B,n]
    with pytest.raises(ValueError):
        eng.ar_generate(enc, rnd, forced_wav=forced[:, :-1])
    with pytest.raises(ValueError):
        eng.ar_generate(enc[:, :, :-1], rnd)
    st = eng.ar_new_state(B)
    with pytest.raises(ValueError):
        eng.ar_step(torch.empty(4096, dtype=torch.uint8, device='cuda'), np.zeros([B], np.float32), enc[:, 0], rnd[0])
    with pytest.raises(ValueError):
        eng.ar_step(st, np.zeros([B], np.float32), enc[:, 0, :-1], rnd[0])
    with pytest.raises(ValueError):
        eng.ar_step(st, np.zeros([B], np.float32), enc[:, 0], rnd[0][:, :-1])
    eng.close()


def test_cond_vars_match_the_reference_definition():
    """Fastgen.cond_vars (wavenet.py:353-377) / fastgen.calculate_cond_vars (fastgen.py:91-115): mel_cond_i(encoding) for every
    layer and mel_cond_out1, biases included, [B, T, channels] -- against a float64 evaluation of the same 1x1 convolutions
    on the variables (masked.conv1d with filter_length 1: x @ W[0, 0] + b), through the Python mirror, on a shipped
    teacher and on one with a doubled gate and odd sizes (ragged 64 x 64 tiles)."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.wavenet.wavenet import Fastgen
    rs = np.random.RandomState(5)
    for patch, B, Tn in (({}, 2, 77), ({'double_gate_width': True, 'num_layers': 4, 'num_stages': 2, 'width': 64, 'skip_width': 64}, 3, 130)):
        cfgd = dict(load_json('wavenet_mol.json'), **patch)
        hp = O.HP(cfgd)
        w = O.synth_weights(hp, 'teacher', seed=99, init='unit')
        fg = Fastgen(cfgd, batch_size=B).load_weights(w)
        enc = (rs.standard_normal([B, Tn, cfgd['deconv_width']]) * 0.5).astype(np.float32)
        cv = fg.cond_vars({'encoding': enc})
        names = ['mel_cond_%d' % (i + 1) for i in range(cfgd['num_layers'])] + ['mel_cond_out1']
        assert sorted(cv.keys()) == sorted(names)
        for nme in names:
            Wk = np.asarray(w[nme + '/W'], np.float64)
            ref = enc.astype(np.float64) @ Wk.reshape(Wk.shape[-2], Wk.shape[-1]) + np.asarray(w[nme + '/biases'], np.float64)
            got = _np(cv[nme])
            assert got.shape == ref.shape, (nme, got.shape, ref.shape)
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), nme
        fg.engine.close()
    with pytest.raises(ValueError):
        from nsynth_wavenet_amd.engine import Engine
        e = Engine(load_json('wavenet_mol.json'))
        try:
            e.ar_cond_vars(np.zeros([1, 4, 17], np.float32))
        finally:
            e.close()


def test_teacher_configs_outside_the_kernel_limits_are_refused():
    """The AR step kernels hold one weight row in a register tile: 3*width + deconv_width <= 4096 (2048 on the tuned
    instantiation, the rest on the wide one), gate_width/2 <= 2048, 1 <= mol_mix <= 64 -- anything else must fail at
    wn_create, not truncate a dot product."""
    from nsynth_wavenet_amd.engine import Engine
    base = load_json('wavenet_mol.json')
    for bad in (dict(width=1408, skip_width=256), dict(mol_mix=65), dict(mol_mix=0)):
        with pytest.raises(ValueError):
            Engine(dict(base, **bad))
    Engine(dict(base, mol_mix=64)).close()


@pytest.mark.parametrize('patch', [
    dict(width=768, skip_width=256),                                    # 3 * 768 + 256 = 2560: past the tuned tile, gate = width
    dict(width=640, skip_width=512, double_gate_width=True),            # gate 1280: H = 640, d row 2176
    dict(width=1024, skip_width=256, deconv_width=512,                  # 3584-float rows, the widest class
         deconv_config=[[40, 10], [80, 20]]),
    dict(width=256, skip_width=1536),                                   # narrow layers, out1 / out2 rows of 1792 / 1536 floats
])
def test_wide_teachers_run_on_the_wide_instantiation(patch):
    """masked.conv1d / Fastgen.sample take any width (wavenet.py:326-345,379-514; masked.py:328-405).  Shapes whose weight
    rows do not fit the tuned register tiles (3 * width + deconv_width > 2048) used to be refused; they now run on a wide
    instantiation of the same step kernels (one utterance at a time) -- same C ABI, same oracle: K1 against the float64
    full-sequence teacher, batch rows independent, free run reproducible, both step forms where a batched pack exists."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    cfgd = dict(load_json('wavenet_mol.json'), num_layers=6, num_stages=3, **patch)
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=4321, init='unit')
    eng = Engine(cfgd).load_weights(w)
    Cd = cfgd['deconv_width']
    rs = np.random.RandomState(1)
    for B in (1, 3, 5, 17):                     # 17: the batched pack of the shapes that have one, with two column tiles
        Tn = 32
        enc = (rs.standard_normal([B, Tn, Cd]) * 0.3).astype(np.float32)
        forced = rs.uniform(-1, 1, [B, Tn]).astype(np.float32)
        out = eng.ar_generate(enc, None, seed=1, forced_wav=forced, want_out=True)
        ref = O.teacher_feed_forward(forced.astype(np.float64), enc.astype(np.float64), w, hp, np.float64)
        assert np.abs(_np(out['out_params']) - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max()), (patch, B)
        a = eng.ar_generate(enc, None, seed=5)
        b = eng.ar_generate(enc, None, seed=5)
        c = eng.ar_generate(enc[:1], None, seed=5)
        assert torch.equal(a['idx'], b['idx']) and torch.equal(a['idx'][:1], c['idx'])
    eng.close()
