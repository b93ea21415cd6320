"""CPU tests of the oracle (the checker itself): the reference's structural
known-answers, its internal invariants (K1-K3), an independent
torch implementation (K9) and the committed golden vectors."""
import json
import os

import numpy as np
import pytest

from oracle import wavenet_np as O
from conftest import load_json

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FACTS = np.load(os.path.join(GOLD, 'ref_fixture_facts.npz'))


# ---- K4: lengths pinned by the reference's committed output wavs ----
def test_reference_fixture_lengths():
    n = int(FACTS['test_wav/n'])
    assert n == 154480 and int(FACTS['test_wav/sr']) == 16000
    F = 1 + n // 200                                   # librosa centred frames, hop 200
    assert F == 773
    hp = O.HP(load_json('parallel_wavenet.json'))
    ar_len = F * 200
    iaf_len = O.iaf_length(F, hp)
    assert int(FACTS['pred_data-no_mu_law+mol/gen_LJ001-0001.wav/n']) == ar_len == 154600
    assert int(FACTS['pred_data-use_mu_law+ce/gen_LJ001-0001.wav/n']) == ar_len
    for name in FACTS['names']:
        if 'pwn-failed_cases' in name:
            assert int(FACTS[name + '/n']) == iaf_len == 154112
    assert (ar_len - iaf_len) // 2 == 244             # centre-crop offset of the conditioning
    assert int(FACTS['pred_data-no_mu_law+mol/gen_LJ001-0002.wav/n']) == 152 * 200


# ---- K5: every non-mu-law output of the reference lies on the 2^-15 grid inside [-1, 1-2^-15] ----
def test_reference_outputs_on_grid_match_clip_quant():
    for name in FACTS['names']:
        name = str(name)
        if 'use_mu_law' in name:
            continue
        assert float(FACTS[name + '/grid_residue']) == 0.0
        assert float(FACTS[name + '/min']) >= -1.0 and float(FACTS[name + '/max']) <= 1 - 2.0 ** -15
    x = np.random.RandomState(0).uniform(-1.2, 1.2, 5000).astype(np.float32)
    wav, q = O.clip_quant_scale(x, 65536, False)
    assert np.all(wav * 32768 == np.round(wav * 32768)) and wav.min() >= -1 and wav.max() <= 1 - 2.0 ** -15
    assert q.min() >= -32768 and q.max() <= 32767


# ---- K6: the mu-law outputs of the reference are the inv_mu_law table (<= 2^-24 abs) ----
def test_inv_mu_law_table_matches_reference_outputs():
    table = O.inv_mu_law(np.arange(-128, 128), dtype=np.float32)
    assert table[128] == 0.0
    for name in FACTS['names']:
        name = str(name)
        if 'use_mu_law' not in name:
            continue
        u = FACTS[name + '/unique']
        assert len(u) <= 256
        d = np.abs(u[:, None].astype(np.float64) - table[None, :].astype(np.float64))
        j = d.argmin(axis=1)
        # (256^|s| - 1) is computed near 1.0, so numpy's and TF's pow may differ by one ulp OF 1.0
        # (2^-24 = 5.96e-8) -- the float table is not bit-pinned across libms, the index is
        assert np.all(d[np.arange(len(u)), j] <= 2.0 ** -24), name
        assert len(set(j.tolist())) == len(u)


# ---- K7: codec round trips ----
def test_codec_round_trips():
    q = np.arange(-128, 128)
    assert np.array_equal(O.mu_law(O.inv_mu_law(q)).astype(int), q)
    q = np.arange(-32768, 32768)
    assert np.array_equal(O.cast_quantize(O.inv_cast_quantize(q, 65536), 65536), q)


def test_softplus_tf_semantics():
    x = np.array([-100, -20, -13.95, -13.9, -1, 0, 1, 13.9, 13.95, 50], np.float32)
    y = O.softplus(x)
    assert np.allclose(y, np.log1p(np.exp(x.astype(np.float64))), rtol=1e-6, atol=1e-30)
    assert y[-1] == 50 and y[0] == np.exp(np.float32(-100))


# ---- K2 / K3 / K9 on the real student config ----
@pytest.mark.parametrize('init', ['tf', 'unit'])
def test_student_invariants_and_torch_agreement(init):
    import torch
    from oracle.torch_ref import StudentRef
    hp = O.HP(load_json('parallel_wavenet.json'))
    w = O.synth_weights(hp, 'student', init=init)
    B, F = 2, 6
    mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80]).astype(np.float32)
    T = O.iaf_length(F, hp)
    assert T == 1024 and (F * 200 - T) // 2 == 88
    noise = O.logistic_from_uniform(np.random.RandomState(2).uniform(1e-5, 1 - 1e-5, [B, T]))
    ff = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
    assert np.all(ff['scale_tot'] > 0)
    assert np.allclose(ff['x'], ff['rand_input'] * ff['scale_tot'] + ff['mean_tot'], rtol=0, atol=1e-12)  # K2
    assert np.abs(ff['iaf_x'] - ff['x']).max() < 1e-9 * max(1.0, np.abs(ff['x']).max())                   # K3
    tr = StudentRef(w, hp, torch.float64).feed_forward(mel, noise)                                        # K9
    assert np.abs(tr['x'] - ff['x']).max() < 1e-10 * max(1.0, np.abs(ff['x']).max())
    f32 = O.iaf_feed_forward(mel, noise, w, hp, np.float32)
    assert np.abs(f32['x'] - ff['x']).max() < 1e-4 * max(1.0, np.abs(ff['x']).max())


def test_gauss_student_has_private_deconv_stacks():
    hp = O.HP(load_json('parallel_wavenet_gauss.json'))
    w = O.synth_weights(hp, 'student')
    assert 'iaf_1/trans_conv_1/kernel' in w and 'iaf_share/trans_conv_1/kernel' not in w
    assert sum(v.size for v in w.values()) == 26189064          # SURVEY Appendix A / BASELINE.md
    hp2 = O.HP(load_json('parallel_wavenet.json'))
    assert sum(v.size for v in O.synth_weights(hp2, 'student').values()) == 8001288


# ---- K1: incremental teacher == full-sequence teacher ----
@pytest.mark.parametrize('name', ['wavenet_mol.json', 'wavenet_ce.json', 'wavenet_gauss.json'])
def test_incremental_equals_full_sequence(name):
    d = load_json(name)
    d.update(width=16, skip_width=8, deconv_width=12, num_layers=7, num_stages=3, deconv_config=[[8, 2], [12, 4]])
    hp = O.HP(d)
    w = O.synth_weights(hp, 'teacher', init='unit')
    B, F = 2, 6
    mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80])
    en = O.deconv_stack(mel, w, hp, '', np.float64)
    Tn = en.shape[1]
    assert Tn == F * 8
    wav = np.random.RandomState(3).uniform(-1, 1, [B, Tn])
    full = O.teacher_feed_forward(O.encode_signal(wav, hp, np.float64), en, w, hp, np.float64)
    fg = O.Fastgen(w, hp, B, np.float64)
    prev = np.zeros((B, 1))
    for t in range(Tn):
        out = fg.out_params(prev, en[:, t])
        assert np.abs(out - full[:, t]).max() < 1e-12
        prev = wav[:, t:t + 1]


def test_teacher_sizes_match_survey():
    hp = O.HP(load_json('wavenet_mol.json'))
    w = O.synth_weights(hp, 'teacher')
    assert sum(v.size for v in w.values()) == 39812382
    assert O.teacher_gate_width(hp) == 512 and O.teacher_out_width(hp) == 30
    hpce = O.HP(load_json('wavenet_ce.json'))
    assert O.teacher_gate_width(hpce) == 1024 and O.teacher_out_width(hpce) == 256   # default double gate


def test_trans_conv_matches_torch_both_layers():
    import torch
    import torch.nn.functional as Fn
    rs = np.random.RandomState(0)
    for (K, s, cin, cout) in ((40, 10, 5, 7), (80, 20, 6, 4), (8, 2, 3, 3)):
        x = rs.standard_normal([2, 9, cin])
        W = rs.standard_normal([1, K, cout, cin])
        b = rs.standard_normal([cout])
        y = O.trans_conv1d(x, W, b, s, None)
        yt = Fn.conv_transpose1d(torch.tensor(x).transpose(1, 2), torch.tensor(W[0]).permute(2, 1, 0),
                                 torch.tensor(b), stride=s, padding=(K - s) // 2)
        assert y.shape == (2, 9 * s, cout)
        assert np.abs(y - yt.transpose(1, 2).numpy()).max() < 1e-12


def test_causal_conv_matches_time_to_batch_emulation():
    # masked.py:72-122,202-230: time_to_batch -> left pad -> VALID conv -> batch_to_time
    rs = np.random.RandomState(0)
    x = rs.standard_normal([2, 32, 3])
    W = rs.standard_normal([1, 3, 3, 4])
    b = rs.standard_normal([4])
    for d in (1, 2, 4, 8):
        B, T, C = x.shape
        y = x.reshape(B, T // d, d, C).transpose(0, 2, 1, 3).reshape(B * d, T // d, C)
        y = np.pad(y, [(0, 0), (2, 0), (0, 0)])
        out = sum(y[:, k:k + T // d] @ W[0, k] for k in range(3)) + b
        out = out.reshape(B, d, T // d, 4).transpose(0, 2, 1, 3).reshape(B, T, 4)
        assert np.abs(out - O.conv1d(x, W, b, d)).max() < 1e-12


# ---- samplers ----
def test_samplers_with_injected_randoms():
    rs = np.random.RandomState(0)
    out = rs.standard_normal([5, 30]).astype(np.float32)
    q = O.mol_sample(out, 65536, rs.uniform(1e-5, 1 - 1e-5, [5, 10]), rs.uniform(1e-5, 1 - 1e-5, [5]))
    assert q.dtype == np.int32 and q.min() >= -32768 and q.max() <= 32767
    g = O.gauss_sample(np.array([[0.25, -20.0]], np.float32), 65536, np.array([3.0]))
    assert g[0] == int(np.floor((0.25 + np.exp(-7.0) * 3.0) * 32768))       # log-std floored at -7
    logits = np.full([1, 256], -50.0, np.float32)
    logits[0, 200] = 10.0
    assert O.ce_sample(logits, 256, np.array([0.5]))[0] == 200 - 128
    lg = np.log(np.array([[0.25, 0.25, 0.5]], np.float32))
    assert [int(O.ce_sample(lg, 2, np.array([u]))[0]) + 1 for u in (0.1, 0.3, 0.6, 0.99)] == [0, 1, 2, 2]


# ---- golden vectors are what the oracle produces today ----
@pytest.mark.parametrize('tag', ['iaf_logistic_unit', 'iaf_mulaw'])
def test_golden_iaf_reproduces(tag):
    g = np.load(os.path.join(GOLD, tag + '.npz'))
    hp = O.HP(json.loads(str(g['cfg_json'])))
    w = O.synth_weights(hp, 'student', seed=int(g['seed']), init=str(g['init']))
    wav, idx, ff = O.parallelgen(g['mel'], g['noise'], w, hp, np.float64)
    assert np.array_equal(idx, g['idx'])
    assert np.abs(ff['x'] - g['x']).max() == 0.0


def test_golden_ar_reproduces():
    g = np.load(os.path.join(GOLD, 'ar_gauss.npz'))
    hp = O.HP(json.loads(str(g['cfg_json'])))
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    wav, idx = O.fastgen_synthesis(g['enc'], g['rnd'], w, hp, np.float32)
    assert np.array_equal(idx, g['free_idx'])


def test_golden_codec():
    g = np.load(os.path.join(GOLD, 'codec.npz'))
    w16, q16 = O.clip_quant_scale(g['x'], 65536, False)
    w8, q8 = O.clip_quant_scale(g['x'], 256, True)
    assert np.array_equal(q16, g['idx16']) and np.array_equal(q8, g['idx8'])
    assert np.array_equal(w16, g['wav16'])


@pytest.mark.parametrize('fl,S', [(40, 10), (80, 20), (3, 2), (1, 4), (7, 3), (5, 1)])
def test_resize_conv_matches_direct_loops_and_phase_gemm_form(fl, S):
    """masked.py:294-322 (nearest-neighbour resize + non-causal SAME conv): the vectorised
    restatement equals the definition evaluated sample by sample, and equals the per-phase GEMM
    with the summed kernel that libwnhip packs (wn_deconv.hip wn_pack_deconv, resize branch)."""
    rs = np.random.RandomState(fl * 100 + S)
    L, cin, cout = 6, 3, 2
    x = rs.standard_normal((2, L, cin))
    W = rs.standard_normal((1, fl, cin, cout))
    b = rs.standard_normal(cout)
    y = O.resize_conv1d(x, W, b, S, None)
    T, pl = L * S, (fl - 1) // 2
    direct = np.zeros((2, T, cout)) + b
    for t in range(T):
        for k in range(fl):
            tau = t + k - pl
            if 0 <= tau < T:
                direct[:, t] += x[:, tau // S] @ W[0, k]
    assert np.allclose(y, direct, atol=1e-12)
    taps, c = (fl - 1 + S - 1) // S + 1, fl - 1 - pl
    Weff = np.zeros((S * taps, cin, cout))
    for r in range(S):
        for k in range(fl):
            m = r + k - (fl - 1)
            Weff[S * (0 if m >= 0 else (-m + S - 1) // S) + r] += W[0, k]
    gemm = np.zeros((2, T, cout)) + b
    for t in range(T):
        q, r = divmod(t + c, S)
        for j in range(taps):
            if 0 <= q - j < L:
                gemm[:, t] += x[:, q - j] @ Weff[S * j + r]
    assert np.allclose(y, gemm, atol=1e-12)


def test_resize_conv_student_runs_through_the_oracle():
    d = load_json('parallel_wavenet.json')
    d['use_resize_conv'] = True
    hp = O.HP(d)
    w = O.synth_weights(hp, 'student', init='unit')
    assert 'iaf_share/resize_conv_1/W' in w and w['iaf_share/resize_conv_2/W'].shape == (1, 80, 256, 256)
    assert not any('trans_conv' in k for k in w)
    mel = np.random.RandomState(0).uniform(0, 1, [1, 3, 80]).astype(np.float32)
    enc = O.deconv_stack(mel, w, hp, 'iaf_share', np.float64)
    assert enc.shape == (1, 600, 256) and np.isfinite(enc).all() and 0.05 < enc.std() < 5


def test_scale_path_statistics_of_reference_test_scale():
    """tests/test_scale.py:64-107 of the reference draws scale = clip(softplus(N(0,1)), e^-9, e^7) for four
    flows and multiplies them (get_scale / reduce(np.multiply), use_log_scale=False) -- 76 800 draws.  The
    restated `scale_log_scale` (parallel_wavenet.py:105-114) on the same kind of draws must reproduce the
    closed-form moments of that product: E = (E s)^4 = 0.42215, std = sqrt((E s^2)^4 - (E s)^8) = 0.73625.
    (The line the reference prints next to it, "scale.m 0.38296, scale.std 0.61160", is a log line of a
    weight-normalised TF model, i.e. of parameters ~ N(-0.01, 0.93^2), not of N(0,1) draws: solving the two
    moments for (mu, sigma) gives exactly that; it is therefore an order-of-magnitude check only.)"""
    from scipy import integrate
    pdf = lambda z: np.exp(-z * z / 2) / np.sqrt(2 * np.pi)
    m1 = integrate.quad(lambda z: np.log1p(np.exp(z)) * pdf(z), -12, 12)[0]
    m2 = integrate.quad(lambda z: np.log1p(np.exp(z)) ** 2 * pdf(z), -12, 12)[0]
    assert abs(m1 - O.SOFTPLUS_N01_M1) < 1e-12 and abs(m2 - O.SOFTPLUS_N01_M2) < 1e-12
    n = 7680 * 10
    rs = np.random.RandomState(94107)
    prods = []
    for dtype in (np.float32, np.float64):
        scales = [O.scale_log_scale(rs.standard_normal(n).astype(dtype))[0] for _ in range(4)]
        for sc in scales:                                        # one flow
            assert abs(sc.mean() - m1) < 5 * np.sqrt((m2 - m1 * m1) / n)
        prods.append(np.prod(np.stack(scales).astype(np.float64), axis=0))
    e, sd = m1 ** 4, np.sqrt(m2 ** 4 - m1 ** 8)
    assert abs(e - 0.42215) < 1e-5 and abs(sd - 0.73625) < 1e-5
    for pr in prods:
        assert abs(pr.mean() - e) < 5 * sd / np.sqrt(n)
        assert abs(pr.std() - sd) < 0.05
        assert pr.min() > 0 and 0.2 < pr.mean() / 0.38296 < 2.0 and 0.5 < pr.std() / 0.61160 < 2.0   # same regime
    # through the whole restated student with the probe weights: scale_tot[t] = clip(softplus(noise[t-1]))
    cfgd = dict(load_json('parallel_wavenet.json'), num_iaf_layers=[1])
    hp = O.HP(cfgd)
    w = O.scale_probe_weights(hp)
    noise = np.random.RandomState(3).standard_normal([1, 1024]).astype(np.float32)
    mel = np.zeros([1, 6, 80], np.float32)
    ff = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
    want = O.scale_log_scale(np.concatenate([[0.0], noise[0, :-1].astype(np.float64)]))[0]
    assert np.abs(ff['scale_tot'][0] - want).max() <= 1e-12
    assert np.abs(ff['mean_tot']).max() == 0.0
