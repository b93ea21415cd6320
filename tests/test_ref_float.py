"""The float path against vectors produced by the REFERENCE'S OWN Python (tests/golden/ref_float.npz).

tests/golden/make_ref_float.py imports the reference's wavenet/*.py and auxilaries/utils.py unmodified and drives them the way
eval_parallel_wavenet.py / eval_wavenet.py do (parallelgen.synthesis, fastgen.load_deconv_stack / load_fastgen / synthesis,
Wavenet.feed_forward, Fastgen.cond_vars), with `import tensorflow` resolved to a numpy evaluator of the TensorFlow primitives
those files call (tests/golden/tf_standin.py; TensorFlow itself is not installed).  What that pins: every decision the
reference's code makes (names, scopes, checkpoint keys, dilation / crop / tap / gate arithmetic, flow recursion, queue
discipline, samplers, quantiser, driver loops).  What it does not: TensorFlow's kernels (restated from their documented
semantics) and TF's random generators (injected).  `*_f64` = tf.float32 evaluated in float64, `*_f32` = in float32.

CPU tests hold the oracle (oracle/wavenet_np.py) and the older oracle-made goldens to these vectors; the `gpu` tests hold the
HIP engine, through the C ABI, to them directly.
"""
import json
import os

import numpy as np
import pytest

from conftest import load_json

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
STUDENT_GOLD = ['iaf_logistic_tf', 'iaf_logistic_unit', 'iaf_gauss_perflow', 'iaf_mulaw']
STUDENT_EXTRA = ['iaf_wn_resize', 'iaf_teacher_deconv']
TEACHER_GOLD = ['ar_mol', 'ar_ce_mulaw', 'ar_gauss']
TEACHER_EXTRA = ['ar_wn_resize']


@pytest.fixture(scope='module')
def R():
    return np.load(os.path.join(GOLD, 'ref_float.npz'))


def _case(R, tag):
    """(inputs dict, config dict, weights) of a case: make_golden's inputs for its tags, the fixture's own otherwise."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd import weights as wts, config as cfg
    if tag + '/in_cfg_json' in R.files:
        g = {k[len(tag) + 4:]: R[k] for k in R.files if k.startswith(tag + '/in_')}
        cfgd = json.loads(str(g['cfg_json']))
        w = wts.synthetic_weights(cfg.load_hparams(cfgd), seed=int(g['seed']), init=str(g['init']))
    else:
        g = np.load(os.path.join(GOLD, tag + '.npz'))
        cfgd = json.loads(str(g['cfg_json']))
        w = O.synth_weights(O.HP(cfgd), str(R[tag + '/kind']), seed=int(g['seed']), init=str(g['init']))
    return g, cfgd, w


def _np(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------------------------
# CPU: the oracle, the product's naming and the older goldens against the reference's code
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag', STUDENT_GOLD + STUDENT_EXTRA)
def test_oracle_student_equals_the_reference_code(R, tag):
    """parallel_wavenet.py:200-345 + parallelgen.py:11-19 executed, against oracle.iaf_feed_forward on the noise the
    reference graph drew: float64 to rounding (1e-12 of the range), float32 within what two float32 evaluation orders
    differ by; upsampler output (masked.py:235-322) likewise; quantised audio equal except one step at a boundary."""
    from oracle import wavenet_np as O
    g, cfgd, w = _case(R, tag)
    hp = O.HP(cfgd)
    share = hp.get('use_share_deconv', False) or hp.get('use_teacher_deconv', False)
    Q = O.quant_chann_of(hp)
    for fl, dt, tol in (('f64', np.float64, 1e-12), ('f32', np.float32, 5e-6)):
        ri = R['{}/rand_input_{}'.format(tag, fl)]
        ff = O.iaf_feed_forward(g['mel'], ri, w, hp, dt)
        rng = max(1.0, float(np.abs(R[tag + '/x_f64']).max()))
        for k in ('x', 'mean_tot', 'scale_tot') + (('log_scale_tot',) if fl == 'f64' else ()):
            ref = R['{}/{}_{}'.format(tag, k, fl)]
            assert ff[k].shape == ref.shape
            assert np.abs(ff[k] - ref).max() <= tol * rng, (tag, fl, k)
        if fl == 'f64':                 # (the float32-arithmetic twins are stored for the principal outputs only)
            enc = O.deconv_stack(g['mel'], w, hp, 'iaf_share' if share else 'iaf_1', dt)
            assert np.abs(enc[:, ::7, ::5] - R[tag + '/enc_sub_f64']).max() <= tol * max(1.0, np.abs(enc).max())
            wav, _ = O.clip_quant_scale(ff['x'], Q, hp.use_mu_law, dt)
            assert np.abs(wav.astype(np.float64) - R[tag + '/wav_f64']).max() <= 2.0 ** -23     # stored as float32
    # the reference's own identity (tests/test_parallel_wavenet.py:62-64): x == rand_input * scale_tot + mean_tot
    assert np.array_equal(R[tag + '/x_f64'], R[tag + '/rand_input_f64'] * R[tag + '/scale_tot_f64'] + R[tag + '/mean_tot_f64'])


@pytest.mark.parametrize('tag', TEACHER_GOLD + TEACHER_EXTRA)
def test_oracle_teacher_equals_the_reference_code(R, tag):
    """fastgen.py:58-88 (encoding), wavenet.py:157-291 (full-sequence teacher on a forced waveform), wavenet.py:353-377
    (cond_vars), wavenet.py:379-514 + masked.py:328-405 + loss_func.py:140-206 + fastgen.py:128-169 (the incremental
    sampler's loop with its queues), executed, against the oracle: float64 to rounding, index streams identical."""
    from oracle import wavenet_np as O
    g, cfgd, w = _case(R, tag)
    hp = O.HP(cfgd)
    for fl, dt, tol in (('f64', np.float64, 1e-12), ('f32', np.float32, 5e-6)):
        enc_r = R['{}/enc_{}'.format(tag, fl)]
        enc = O.deconv_stack(g['mel'], w, hp, '', dt)
        assert np.abs(enc - enc_r).max() <= tol * max(1.0, np.abs(enc_r).max())
        ref = R['{}/out_forced_{}'.format(tag, fl)]
        out = O.teacher_feed_forward(O.encode_signal(g['forced'], hp, dt), enc_r, w, hp, dt)
        assert np.abs(out - ref).max() <= tol * max(1.0, np.abs(ref).max())
        wav, idx, outs = O.fastgen_synthesis(enc_r.astype(np.float32), g['rnd'], w, hp, dt, return_out=True)
        assert np.array_equal(idx, R['{}/free_idx_{}'.format(tag, fl)])
        assert np.abs(wav - R['{}/free_wav_{}'.format(tag, fl)]).max() <= 2.0 ** -23
        if fl == 'f64':                 # (the float32-arithmetic twins are stored for the principal outputs only)
            fo = R[tag + '/free_out_f64']
            assert np.abs(outs - fo).max() <= tol * max(1.0, np.abs(fo).max())
            names = sorted(['mel_cond_%d' % (i + 1) for i in range(hp.num_layers)])
            cond = np.stack([O._conv(enc_r, w, n, hp, dtype=dt)[:, ::5, ::7] for n in names])
            assert np.abs(cond - R[tag + '/cond_sub_f64']).max() <= tol * max(1.0, np.abs(cond).max())
            c1 = O._conv(enc_r, w, 'mel_cond_out1', hp, dtype=dt)[:, ::5, ::7]
            assert np.abs(c1 - R[tag + '/cond_out1_sub_f64']).max() <= tol * max(1.0, np.abs(c1).max())
        # K1 on the reference itself: its incremental graph, teacher-forced, IS its full-sequence graph
        assert float(R['{}/k1_{}'.format(tag, fl)]) <= (1e-12 if fl == 'f64' else 5e-6)


@pytest.mark.parametrize('tag', STUDENT_GOLD + STUDENT_EXTRA + TEACHER_GOLD + TEACHER_EXTRA)
def test_variable_and_checkpoint_names_are_the_reference_graphs(R, tag):
    """The variables the reference's graph CREATES (name, shape) and the checkpoint keys its Saver ASKS for
    (fastgen.py:12-14, parallelgen.py:6-8,29-41) are what weights.expected_variables / checkpoint_keys say -- rows a13,
    a14.  (The checkpoint those graphs were restored from was written by weights.save_checkpoint.)"""
    from nsynth_wavenet_amd import weights as wts, config as cfg
    _, cfgd, w = _case(R, tag)
    hp = cfg.load_hparams(cfgd)
    ref_vars = sorted((n, tuple(s)) for n, s in json.loads(str(R[tag + '/vars'])))
    mine = sorted((n, tuple(int(v) for v in s)) for n, s in wts.expected_variables(hp))
    assert mine == ref_vars
    assert sorted(w.keys()) == [n for n, _ in ref_vars]
    assert sorted(wts.checkpoint_keys(w, hp).values()) == sorted(json.loads(str(R[tag + '/ckpt_keys'])))
    if tag == 'iaf_teacher_deconv':
        keys = json.loads(str(R[tag + '/ckpt_keys']))
        assert 'iaf_share/trans_conv_1/kernel' in keys and 'iaf_1/start_conv/W/ExponentialMovingAverage' in keys
    if str(R[tag + '/kind']) == 'teacher':          # the incremental graph takes the encoding: no upsampler variables
        fg = set(json.loads(str(R[tag + '/ckpt_keys_fastgen'])))
        assert fg == {k for k in wts.checkpoint_keys(w, hp).values() if 'trans_conv' not in k and 'resize_conv' not in k}


@pytest.mark.parametrize('tag', STUDENT_GOLD + TEACHER_GOLD)
def test_the_oracle_made_goldens_agree_with_the_reference_code(R, tag):
    """tests/golden/<tag>.npz (oracle-made, what the other GPU tests compare with) against the reference-made vectors of the
    same case.  Teacher: same inputs, equal to rounding.  Student: the golden's noise is the float32 rounding of the noise
    the reference graph computes in float64, so the outputs differ by that rounding carried through the flows."""
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    if tag in TEACHER_GOLD:
        ref = R[tag + '/out_forced_f64']
        assert np.abs(g['out_forced'] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
        assert np.abs(g['enc'] - R[tag + '/enc_f64']).max() <= 2e-7 * max(1.0, np.abs(g['enc']).max())
        assert np.array_equal(g['free_idx'], R[tag + '/free_idx_f32'])
        assert np.array_equal(g['free_idx'], R[tag + '/free_idx_f64'])
    else:
        rng = max(1.0, float(np.abs(g['x']).max()))
        assert np.abs(g['noise'] - R[tag + '/rand_input_f64']).max() <= 1e-6 * max(1.0, np.abs(g['noise']).max())
        for k in ('x', 'mean_tot', 'scale_tot'):
            assert np.abs(g[k] - R['{}/{}_f64'.format(tag, k)]).max() <= 2e-6 * rng


# ------------------------------------------------------------------------------------------------------------------
# GPU: the HIP engine through the C ABI against the reference-made vectors
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['f16x3', 'f16x3-fused', 'f32', 'f32-fused'])
@pytest.mark.parametrize('tag', STUDENT_GOLD + STUDENT_EXTRA)
def test_engine_student_against_the_reference_code(R, tag, precision):
    """wn_iaf_generate on the noise the reference graph drew, against what the reference's code computed from it:
    2e-5 of the range on x / mean_tot / scale_tot (north_star: 1e-3), audio within one quantiser step and only at a
    boundary of the reference's pre-quantisation value."""
    from nsynth_wavenet_amd.engine import Engine
    g, cfgd, w = _case(R, tag)
    eng = Engine(cfgd, precision=precision).load_weights(w)
    noise = R[tag + '/rand_input_f64'].astype(np.float32)
    out = eng.iaf_generate(g['mel'], noise, want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot'))
    x_ref = R[tag + '/x_f64']
    rng = max(1.0, float(np.abs(x_ref).max()))
    for k in ('x', 'mean_tot', 'scale_tot'):
        ref = R['{}/{}_f64'.format(tag, k)]
        err = float(np.abs(_np(out[k]) - ref).max())
        assert err <= 2e-5 * max(1.0, float(np.abs(ref).max()), rng if k == 'x' else 0.0), (tag, precision, k, err)
    Q = 256 if cfgd['use_mu_law'] else 65536
    y = np.clip(x_ref, -1, 1 - 2.0 / Q) * (Q / 2)
    idx_ref = np.floor(y).astype(np.int64)
    di = np.abs(_np(out['idx']).astype(np.int64) - idx_ref)
    dx = np.abs(np.clip(_np(out['x']).astype(np.float64), -1, 1 - 2.0 / Q) - np.clip(x_ref, -1, 1 - 2.0 / Q)) * (Q / 2)
    assert np.all(di <= np.ceil(dx) + (dx > 0))
    flips = di != 0
    margin = np.minimum(y - np.floor(y), np.floor(y) + 1 - y)
    assert np.all(margin[flips] <= dx[flips] + 1e-9)
    assert np.abs(_np(out['wav']) - R[tag + '/wav_f64']).max() <= 1e-3
    print('{} {}: max|x - reference code| = {:.2e} on a range of {:.2f}; {} of {} indices one step off, all at a boundary'.format(
        tag, precision, float(np.abs(_np(out['x']) - x_ref).max()), rng, int(flips.sum()), di.size))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TEACHER_GOLD + TEACHER_EXTRA)
def test_engine_teacher_against_the_reference_code(R, tag):
    """wn_deconv, the teacher-forced AR steps (== the reference's full-sequence teacher) and the free-running loop with the
    sampler's randoms injected, against what the reference's code computed."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g, cfgd, w = _case(R, tag)
    hp = O.HP(cfgd)
    eng = Engine(cfgd).load_weights(w)
    enc_r = R[tag + '/enc_f64']
    enc = _np(eng.deconv(g['mel']))
    assert enc.shape == enc_r.shape and np.abs(enc - enc_r).max() <= 1e-5 * max(1.0, np.abs(enc_r).max())
    enc32 = enc_r.astype(np.float32)
    ref = R[tag + '/out_forced_f64']
    out = eng.ar_generate(enc32, g['rnd'], forced_wav=g['forced'], want_out=True)
    assert np.abs(_np(out['out_params']) - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    out = eng.ar_generate(enc32, g['rnd'], want_out=True)
    gi, gop = _np(out['idx']), _np(out['out_params'])
    ri = R[tag + '/free_idx_f64']
    diff = gi != ri
    Tn = gi.shape[1]
    first = int(np.argwhere(diff)[:, 1].min()) if diff.any() else Tn
    Q = 256 if cfgd['use_mu_law'] else 65536
    if first < Tn:                                  # the two loops may fork only where the engine's own pre-floor value
        assert not diff[:, :first].any()            # sits at a decision boundary (then by one step)
        rows = np.where(diff[:, first])[0]
        _, margin, gap = O.sample_margin(gop[:, first], g['rnd'][first], hp)
        lim = 2e-4 if hp.loss_type == 'ce' else (0.25 if Q == 65536 else 5e-3)
        near = margin[rows] <= lim
        if hp.loss_type == 'mol':
            near |= gap[rows] <= 1e-4
        assert np.all(near), (first, margin[rows])
    else:
        assert np.abs(_np(out['wav']) - R[tag + '/free_wav_f64']).max() <= 2.0 ** -23
    fo = R[tag + '/free_out_f64'][:, :max(first, 1)]
    assert np.abs(gop[:, :max(first, 1)] - fo).max() <= 2e-5 * max(1.0, np.abs(fo).max())
    print('{}: free run identical to the reference code\'s loop for {} of {} steps'.format(tag, first, Tn))
    eng.close()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] at its full size (one utterance of 384 frames = 76 800 samples, bench.py's weights), through
# parallelgen.synthesis as written: tests/golden/ref_float_full.npz (make_ref_float.py --full)
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def RF():
    return np.load(os.path.join(GOLD, 'ref_float_full.npz'))


def _full_noise():
    """The noise the reference graph drew at full size: log(u) - log(1 - u) in float64 on the seeded float32 uniforms
    make_ref_float.py injected (the generator asserts this expression equals the graph's rand_input to 1e-14)."""
    u = np.random.RandomState(12346).uniform(1e-5, 1 - 1e-5, [1, 76800]).astype(np.float32).astype(np.float64)
    return np.log(u) - np.log(1.0 - u)


def test_oracle_equals_the_reference_code_at_full_size(RF):
    from oracle import wavenet_np as O
    g, cfgd, w = _case(RF, 'full')
    hp = O.HP(cfgd)
    ff = O.iaf_feed_forward(g['mel'], _full_noise(), w, hp, np.float64)
    assert ff['x'].shape == (1, 76800)
    assert np.abs(ff['x'] - RF['full/x_f64']).max() <= 1e-12
    assert np.abs(ff['scale_tot'] - RF['full/scale_tot_f32']).max() <= 1e-6          # stored as float32
    _, idx = O.clip_quant_scale(ff['x'], 65536, False, np.float64)
    assert np.array_equal(idx, RF['full/idx_i16'].astype(np.int32))                  # the wav file synthesis() wrote
    enc = O.deconv_stack(g['mel'], w, hp, 'iaf_share', np.float64)
    assert np.abs(enc[:, ::7, ::5][:, ::16] - RF['full/enc_sub_f32']).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['f32', 'f32-fused'])
def test_fp32_engine_equals_the_reference_code_at_full_size(RF, precision):
    """The headline configuration in the reference's own arithmetic (fp32 MFMA; `roofline_f32` of the bench line), hoisted form
    (round 6: conditioning GEMM, Q4 rows, head in the last layer's epilogue) and fused form, against the reference's code."""
    from nsynth_wavenet_amd.engine import Engine
    g, cfgd, w = _case(RF, 'full')
    eng = Engine(cfgd, precision=precision).load_weights(w)
    assert eng.iaf_cond_hoisted(1, 384) is (precision == 'f32')
    noise = _full_noise()
    out = eng.iaf_generate(g['mel'], noise.astype(np.float32), want=('idx', 'x', 'scale_tot'))
    x_ref = RF['full/x_f64']
    err = float(np.abs(_np(out['x']) - x_ref).max())
    assert err <= 2e-5 * max(1.0, float(np.abs(x_ref).max()))
    assert np.abs(_np(out['scale_tot']) - RF['full/scale_tot_f32'].astype(np.float64)).max() <= 2e-5
    di = np.abs(_np(out['idx']).astype(np.int64) - RF['full/idx_i16'].astype(np.int64))
    assert di.max() <= 1 and (di != 0).mean() < 0.02
    print('full size, {}: max|x - reference code| = {:.2e}; {} of {} indices one step off'.format(precision, err, int((di != 0).sum()), di.size))
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('groups', [1, -1])
def test_engine_equals_the_reference_code_at_full_size(RF, groups):
    """The headline configuration on the default arithmetic, layer groups in LDS (what bench.py times) and per-layer
    launches, against the reference's code: 2e-5 (north_star: 1e-3), indices one step off only at a boundary."""
    from nsynth_wavenet_amd.engine import Engine
    g, cfgd, w = _case(RF, 'full')
    eng = Engine(cfgd).load_weights(w)
    eng.set_layer_groups(groups)
    assert eng.iaf_layer_groups(1, 384) is (groups > 0)
    noise = _full_noise()
    out = eng.iaf_generate(g['mel'], noise.astype(np.float32), want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot'))
    x_ref = RF['full/x_f64']
    err = float(np.abs(_np(out['x']) - x_ref).max())
    assert err <= 2e-5 * max(1.0, float(np.abs(x_ref).max()))
    scale_ref = RF['full/scale_tot_f32'].astype(np.float64)
    assert np.abs(_np(out['scale_tot']) - scale_ref).max() <= 2e-5
    assert np.abs(_np(out['mean_tot']) - (x_ref - noise * scale_ref)).max() <= 2e-5   # parallel_wavenet.py:326
    y = np.clip(x_ref, -1, 1 - 2.0 / 65536) * 32768
    assert np.array_equal(np.floor(y).astype(np.int64), RF['full/idx_i16'].astype(np.int64))
    di = np.abs(_np(out['idx']).astype(np.int64) - RF['full/idx_i16'].astype(np.int64))
    dx = np.abs(np.clip(_np(out['x']).astype(np.float64), -1, 1 - 2.0 / 65536) * 32768 - y)
    flips = di != 0
    assert di.max() <= 1 and flips.mean() < 0.02
    assert np.all(np.minimum(y - np.floor(y), np.floor(y) + 1 - y)[flips] <= dx[flips] + 1e-9)
    assert np.abs(_np(out['wav']).astype(np.float64) * 32768 - RF['full/idx_i16']).max() <= 1.0
    print('full size, groups {:+d}: max|x - reference code| = {:.2e}; {} of {} indices one step off, all at a boundary'.format(
        groups, err, int(flips.sum()), di.size))
    eng.close()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[0] / [3]: wavenet_mol.json as shipped (width 512, 30 layers, MoL-10), 400 steps of the reference's
# incremental graph, teacher-forced and free running (make_ref_float.py: teacher_full_width_case)
# ------------------------------------------------------------------------------------------------------------------
def _teacher_full_inputs():
    rs = np.random.RandomState(41)
    B, Tn, M = 2, 400, 10
    enc32 = (rs.standard_normal([B, Tn, 256]) * 0.3).astype(np.float32)
    rnd = rs.uniform(1e-5, 1 - 1e-5, [Tn, B, M + 1]).astype(np.float32)
    forced = rs.uniform(-1, 1, [B, Tn]).astype(np.float32)
    return enc32, rnd, forced


def test_oracle_full_width_teacher_equals_the_reference_code(R):
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd import weights as wts, config as cfg
    cfgd = json.loads(str(R['ar_mol_full/in_cfg_json']))
    assert cfgd == load_json('wavenet_mol.json')
    hp = O.HP(cfgd)
    w = wts.synthetic_weights(cfg.load_hparams(cfgd), seed=1234, init='unit')
    enc32, rnd, forced = _teacher_full_inputs()
    wav, idx, outs = O.fastgen_synthesis(enc32, rnd, w, hp, np.float64, return_out=True)
    assert np.array_equal(idx, R['ar_mol_full/free_idx_f64'])
    assert np.abs(outs - R['ar_mol_full/free_out_f64']).max() <= 1e-5           # stored as float32
    fg = O.Fastgen(w, hp, 2, np.float64)
    prev = np.zeros([2, 1])
    ref = R['ar_mol_full/out_forced_f64']
    for t in range(forced.shape[1]):
        o = fg.out_params(prev, enc32[:, t])
        assert np.abs(o - ref[:, t]).max() <= 1e-11 * max(1.0, np.abs(ref).max())
        prev = forced[:, t:t + 1]


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['gemv', 'mfma'])
def test_engine_full_width_teacher_against_the_reference_code(R, mode, monkeypatch):
    """wavenet_mol.json as shipped on the tuned AR step kernels (GEMV step, and the batched MFMA step forced onto the same
    two utterances): every teacher-forced network output of 400 steps against the reference's incremental graph, and the
    free-running loop identical to the reference's until a step whose pre-floor value sits at a decision boundary."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    from nsynth_wavenet_amd import weights as wts, config as cfg
    monkeypatch.setenv('WN_AR_MODE', mode)
    cfgd = load_json('wavenet_mol.json')
    hp = O.HP(cfgd)
    w = wts.synthetic_weights(cfg.load_hparams(cfgd), seed=1234, init='unit')
    eng = Engine(cfgd).load_weights(w)
    enc32, rnd, forced = _teacher_full_inputs()
    ref = R['ar_mol_full/out_forced_f64']
    out = eng.ar_generate(enc32, rnd, forced_wav=forced, want_out=True)
    err = float(np.abs(_np(out['out_params']) - ref).max())
    assert err <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    out = eng.ar_generate(enc32, rnd, want_out=True)
    gi, gop = _np(out['idx']), _np(out['out_params'])
    ri = R['ar_mol_full/free_idx_f64']
    diff = gi != ri
    Tn = gi.shape[1]
    first = int(np.argwhere(diff)[:, 1].min()) if diff.any() else Tn
    if first < Tn:
        assert not diff[:, :first].any()
        rows = np.where(diff[:, first])[0]
        _, margin, gap = O.sample_margin(gop[:, first], rnd[first], hp)
        assert np.all((margin[rows] <= 0.25) | (gap[rows] <= 1e-4)), (first, margin[rows], gap[rows])
    fo = R['ar_mol_full/free_out_f64'][:, :max(first, 1)]
    assert np.abs(gop[:, :max(first, 1)] - fo).max() <= 2e-5 * max(1.0, np.abs(fo).max())
    print('wavenet_mol.json as shipped, {} step: max|out - reference code| = {:.2e} over 400 forced steps; free run identical '
          'for {} of {} steps'.format(mode, err, first, Tn))
    eng.close()


# ------------------------------------------------------------------------------------------------------------------
# teacher scoring: Wavenet.calculate_loss (wavenet.py:293-316) and its per-sample term, reference code executed
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag', TEACHER_GOLD + TEACHER_EXTRA)
def test_oracle_teacher_scoring_equals_the_reference_code(R, tag):
    """loss_func.mol_log_probs / gauss_log_prob / the cross entropy on Wavenet.encode_signal's targets, and the scalar
    Wavenet.calculate_loss returns: oracle equal to float64 rounding."""
    from oracle import wavenet_np as O
    g, cfgd, w = _case(R, tag)
    hp = O.HP(cfgd)
    ref = R[tag + '/logp_f64']
    lp = O.teacher_log_prob(R[tag + '/out_forced_f64'], g['forced'], hp, np.float64)
    assert lp.shape == ref.shape and np.abs(lp - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert abs(-lp.mean() - float(R[tag + '/loss_f64'])) <= 1e-10 * max(1.0, abs(float(R[tag + '/loss_f64'])))


def torch_equal(a, b):
    import torch
    return torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TEACHER_GOLD + TEACHER_EXTRA)
def test_engine_teacher_scoring_against_the_reference_code(R, tag):
    """wn_teacher_log_prob on the reference's out_params and on the engine's own (wn_teacher_forward), against the
    reference's float64 log-likelihoods; Wavenet.calculate_loss of the Python mirror against the reference's scalar."""
    from nsynth_wavenet_amd.engine import Engine
    from nsynth_wavenet_amd.wavenet.wavenet import Wavenet
    g, cfgd, w = _case(R, tag)
    eng = Engine(cfgd).load_weights(w)
    ref = R[tag + '/logp_f64']
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
    lp = _np(eng.teacher_log_prob(R[tag + '/out_forced_f64'].astype(np.float32), g['forced']))
    e1 = float(np.abs(lp - ref).max())
    assert lp.shape == ref.shape and e1 <= tol, (tag, e1)
    wn = Wavenet(cfgd, engine=eng)
    ff = wn.feed_forward({'wav': g['forced'], 'mel': g['mel']})
    res = wn.calculate_loss({'out_params': ff['out_params'], 'wav': g['forced']})
    e2 = float(np.abs(_np(res['log_probs']) - ref).max())
    assert e2 <= 10 * tol, (tag, e2)
    assert abs(float(res['loss']) - float(R[tag + '/loss_f64'])) <= 2e-5 * max(1.0, abs(float(R[tag + '/loss_f64'])))
    # the reference's own call sequence (train_wavenet.py:104-108, tests/test_wavenet.py:38-42) runs unchanged ...
    inputs = {'wav': g['forced'], 'mel': g['mel']}
    ff_dict = wn.feed_forward(inputs)
    ff_dict.update(wn.encode_signal(inputs))
    assert torch_equal(wn.calculate_loss(ff_dict)['log_probs'], res['log_probs'])
    # ... and so does a dictionary that holds only the reference's keys (no raw 'wav')
    ref_keys = {k: ff_dict[k] for k in ('out_params', 'real_targets', 'cate_targets')}
    assert torch_equal(wn.calculate_loss(ref_keys)['log_probs'], res['log_probs'])
    with pytest.raises(KeyError):
        wn.calculate_loss({'out_params': ff['out_params']})
    from oracle import wavenet_np as O
    es = wn.encode_signal({'wav': g['forced']})
    real, cate = O.encode_targets(g['forced'], O.HP(cfgd), np.float32)
    assert np.array_equal(es['real_targets'], real) and np.array_equal(es['cate_targets'], cate) and es['wav_scaled'] is es['real_targets']
    with pytest.raises(ValueError):
        eng.teacher_log_prob(np.zeros([2, 8, 3], np.float32), np.zeros([2, 8], np.float32))
    print('{}: max|log p - reference code| = {:.2e} on the reference\'s out_params, {:.2e} end to end (range {:.1f}); loss {:.6f} vs {:.6f}'.format(
        tag, e1, e2, float(np.abs(ref).max()), float(res['loss']), float(R[tag + '/loss_f64'])))
    eng.close()


# ------------------------------------------------------------------------------------------------------------------
# the shapes and seeds of the reference's own hot-path tests, through the reference's code
# ------------------------------------------------------------------------------------------------------------------
def _k8_noise(gauss):
    rs = np.random.RandomState(12346)
    d = (rs.standard_normal([4, 7680]) if gauss else rs.uniform(1e-5, 1 - 1e-5, [4, 7680])).astype(np.float32).astype(np.float64)
    return d if gauss else np.log(d) - np.log(1.0 - d)


@pytest.mark.parametrize('tag', ['k8_pw', 'k8_pw_gauss'])
def test_oracle_equals_the_reference_code_on_the_reference_test_shape(R, tag):
    """tests/test_parallel_wavenet.py:25-31: four utterances of 39 frames = 7 680 samples, the student JSONs as shipped."""
    from oracle import wavenet_np as O
    g, cfgd, w = _case(R, tag)
    assert cfgd == load_json('parallel_wavenet_gauss.json' if tag.endswith('gauss') else 'parallel_wavenet.json')
    hp = O.HP(cfgd)
    ff = O.iaf_feed_forward(g['mel'], _k8_noise(hp.loss_type == 'gauss'), w, hp, np.float64)
    ref = R[tag + '/x_f64']
    assert ff['x'].shape == (4, 7680) and np.abs(ff['x'] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    assert np.abs(ff['scale_tot'] - R[tag + '/scale_tot_f32']).max() <= 1e-6 * max(1.0, np.abs(ff['scale_tot']).max())


def test_oracle_equals_the_reference_code_on_the_fastgen_test_step(R):
    """tests/test_fastgen.py:17-32: one step of Fastgen.sample, wavenet_mol.json as shipped, batch 4, its seed-12345 inputs."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd import weights as wts, config as cfg
    cfgd = json.loads(str(R['k8_fastgen/in_cfg_json']))
    np.random.seed(12345)
    assert np.array_equal(np.random.uniform(-1, 1, [4, 1]).astype(np.float32), R['k8_fastgen/wav'])
    hp = O.HP(cfgd)
    w = wts.synthetic_weights(cfg.load_hparams(cfgd), seed=1234, init='unit')
    fg = O.Fastgen(w, hp, 4, np.float64)
    out = fg.out_params(R['k8_fastgen/wav'].astype(np.float64), R['k8_fastgen/encoding'].astype(np.float64))
    # (the reference fed the float64 draws; the fixture keeps their float32 rounding, what the device takes)
    assert np.abs(out - R['k8_fastgen/out_f64'][:, :]).max() <= 2e-6
    assert np.array_equal(fg.sample_from(R['k8_fastgen/out_f64'], R['k8_fastgen/rnd'][0]), R['k8_fastgen/sample'][:, 0])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['k8_pw', 'k8_pw_gauss'])
def test_engine_against_the_reference_code_on_the_reference_test_shape(R, tag):
    from nsynth_wavenet_amd.engine import Engine
    g, cfgd, w = _case(R, tag)
    eng = Engine(cfgd).load_weights(w)
    noise = _k8_noise(cfgd['loss_type'] == 'gauss')
    out = eng.iaf_generate(g['mel'], noise.astype(np.float32), want=('x', 'scale_tot'))
    ref = R[tag + '/x_f64']
    err = float(np.abs(_np(out['x']) - ref).max())
    assert err <= 2e-5 * max(1.0, float(np.abs(ref).max())), (tag, err)
    st = R[tag + '/scale_tot_f32']
    assert np.abs(_np(out['scale_tot']) - st).max() <= 2e-5 * max(1.0, float(np.abs(st).max()))
    print('{}: max|x - reference code| = {:.2e} on a range of {:.1f} (range fallbacks {})'.format(
        tag, err, float(np.abs(ref).max()), getattr(eng, 'range_fallbacks', 0)))
    eng.close()


@pytest.mark.gpu
def test_engine_against_the_reference_code_on_the_fastgen_test_step(R):
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    from nsynth_wavenet_amd import weights as wts, config as cfg
    cfgd = json.loads(str(R['k8_fastgen/in_cfg_json']))
    w = wts.synthetic_weights(cfg.load_hparams(cfgd), seed=1234, init='unit')
    eng = Engine(cfgd).load_weights(w)
    st = eng.ar_new_state(4)
    s, op = eng.ar_step(st, R['k8_fastgen/wav'][:, 0], R['k8_fastgen/encoding'], R['k8_fastgen/rnd'][0], want_out=True)
    ref = R['k8_fastgen/out_f64']
    err = float(np.abs(_np(op) - ref).max())
    assert err <= 2e-5 * max(1.0, float(np.abs(ref).max()))
    i64, margin, gap = O.sample_margin(_np(op), R['k8_fastgen/rnd'][0], O.HP(cfgd))
    d = _np(s).astype(np.int64) - R['k8_fastgen/sample'][:, 0]
    assert np.all((d == 0) | ((np.abs(d) <= 1) & (margin <= 0.02)) | (gap <= 1e-5))
    print('test_fastgen.py step: max|out - reference code| = {:.2e}; sample {} vs {}'.format(err, _np(s), R['k8_fastgen/sample'][:, 0]))
    eng.close()
