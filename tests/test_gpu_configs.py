"""BASELINE.json configurations that the other GPU tests do not run at their own shape:

  configs[0]  the reference's own CPU-runnable case: the tests/test_data utterance (154 480 samples ->
              F = 773 mel frames) through fastgen (wavenet_mol.json teacher, 154 600 samples out) and
              through the parallel student (154 112 samples out, centre crop 244) -- the lengths of the
              reference's committed output wavs (tests/golden/ref_fixture_facts.npz);
  configs[4]  parallel_wavenet_gauss.json as shipped (Gaussian head, four PRIVATE deconv stacks), the
              per-GPU share of 128 utterances over 8 GPUs = 16 utterances of F = 384.

configs[0] runs on the mel of the reference's ACTUAL test utterance: tests/golden/fixture_mel.npz holds the [773, 80]
mel of /root/reference/tests/test_data/test.wav, computed in the build container by tests/golden/make_fixture_mel.py with
this repository's featuriser (data; the wav itself is not copied -- only its first 2 048 samples, the teacher-forced prefix of
the K1 check).  The device featuriser is exercised on a synthetic utterance of the same length.  Everything else --
upsampler, flows / autoregressive loop, quantiser -- is the product path through the C ABI.  Checked: the reference-held facts (lengths, 2^-15 grid, range), the reference's
invariants K1 / K2, and parity with the CPU restatements where they finish in seconds."""
import os

import numpy as np
import pytest

from conftest import load_json

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _np(t):
    return t.detach().cpu().numpy()


def _fixture_shaped_wav():
    """154 480 samples (the length of tests/test_data/test.wav): a few decaying partials + noise."""
    rs = np.random.RandomState(2024)
    n = 154480
    t = np.arange(n) / 16000.0
    y = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in ((0.30, 180.0, 0.1), (0.20, 410.0, 1.3), (0.10, 2250.0, 2.0)))
    y = y * (0.6 + 0.4 * np.sin(2 * np.pi * 1.7 * t)) + 0.02 * rs.standard_normal(n)
    return np.clip(y, -0.99, 0.99).astype(np.float32)[None, :]


def _fixture_mel():
    """([1, 773, 80] mel of the reference's tests/test_data/test.wav, its first 2 048 samples [1, 2048])."""
    fx = np.load(os.path.join(GOLD, 'fixture_mel.npz'))
    facts = np.load(os.path.join(GOLD, 'ref_fixture_facts.npz'))
    assert fx['mel'].shape == (773, 80) and int(fx['n_samples']) == int(facts['test_wav/n']) == 154480
    return fx['mel'][None].astype(np.float32), fx['wav_head'][None].astype(np.float32)


def test_config0_student_on_the_reference_fixture_mel():
    """The mel of the reference's own test utterance (F = 773) -> IAF: 154 112 samples with centre crop 244 (K4), K2,
    K5, both execution forms, and parity with the independent torch-CPU implementation on the whole utterance; the
    device featuriser gives the same frame count for an utterance of that length."""
    import torch
    from oracle import wavenet_np as O
    from oracle.torch_ref import StudentRef
    from nsynth_wavenet_amd.auxilaries import mel_extractor as M
    from nsynth_wavenet_amd.engine import Engine
    facts = np.load(os.path.join(GOLD, 'ref_fixture_facts.npz'))
    wav_in = _fixture_shaped_wav()
    mel_dev = M.batch_melspectrogram_device(wav_in)
    assert mel_dev.is_cuda and tuple(mel_dev.shape) == (1, 773, 80)
    mel = torch.from_numpy(_fixture_mel()[0]).cuda()
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    T = 154112
    assert O.iaf_length(773, hp) == T and (773 * 200 - T) // 2 == 244
    assert int(facts['test_wav/n']) == wav_in.shape[1]
    assert int(facts['pred_data-pwn-failed_cases/gen_LJ001-0001-cl.wav/n']) == T     # the reference's own output length
    noise = O.logistic_from_uniform(np.random.RandomState(12346).uniform(1e-5, 1 - 1e-5, [1, T]))
    outs = {}
    for prec in ('f16x3', 'f16x3-fused'):
        eng = Engine(cfgd, precision=prec).load_weights(w)
        assert eng.iaf_length(773) == T
        a = eng.iaf_generate(mel, noise, want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot', 'rand_input'))
        b = eng.iaf_generate(mel, noise, want=('x',))
        assert torch.equal(a['x'], b['x'])                                   # deterministic
        outs[prec] = {k: _np(v) for k, v in a.items()}
        eng.close()
    o = outs['f16x3']
    x, m, s, r = (o[k].astype(np.float64) for k in ('x', 'mean_tot', 'scale_tot', 'rand_input'))
    assert x.shape == (1, T) and np.all(np.isfinite(x)) and np.all(s > 0)
    assert np.abs(x - (r * s + m)).max() <= 1e-6 * max(1.0, np.abs(x).max())                         # K2
    wv = o['wav'].astype(np.float64)
    assert np.all(wv * 32768 == np.round(wv * 32768)) and wv.min() >= -1 and wv.max() <= 1 - 2.0 ** -15   # K5
    assert np.array_equal(o['idx'], (wv * 32768).astype(np.int32))
    scale = max(1.0, np.abs(x).max())
    assert np.abs(outs['f16x3-fused']['x'] - o['x']).max() <= 2e-6 * scale
    ref_x = StudentRef(w, hp).feed_forward(_np(mel), noise)['x'].astype(np.float64)   # torch-CPU fp32, whole utterance
    assert np.abs(o['x'] - ref_x).max() <= 2e-5 * scale


@pytest.mark.parametrize('batch', [2, 3])
def test_layer_groups_at_two_and_three_full_size_utterances(batch):
    """BASELINE configs[1] length (F = 384 -> T = 76 800) at the other batch sizes the launch policy gives to the
    layer-group kernel (the default form): the group form against the independent
    torch-CPU implementation on every sample of every utterance, the per-layer form on the same call, and row
    independence (utterance b of the batched call == the same utterance alone, bit for bit)."""
    import torch
    from oracle import wavenet_np as O
    from oracle.torch_ref import StudentRef
    from nsynth_wavenet_amd.engine import Engine
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    F, T = 384, 76800
    assert O.iaf_length(F, hp) == T
    mel = np.random.RandomState(12345 + batch).uniform(0, 1, [batch, F, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(np.random.RandomState(12346 + batch).uniform(1e-5, 1 - 1e-5, [batch, T]))
    eng = Engine(cfgd).load_weights(w)
    assert eng.iaf_cond_hoisted(batch, F) and eng.iaf_layer_groups(batch, F)        # the policy picks the group kernel here
    out = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot', 'idx'))
    x = _np(out['x']).astype(np.float64)
    m, sc = _np(out['mean_tot']).astype(np.float64), _np(out['scale_tot']).astype(np.float64)
    assert x.shape == (batch, T) and np.isfinite(x).all() and np.all(sc > 0)
    scale = max(1.0, np.abs(x).max())
    assert np.abs(x - (noise.astype(np.float64) * sc + m)).max() <= 1e-6 * scale                      # K2
    ref = StudentRef(w, hp).feed_forward(mel, noise)
    assert np.abs(x - ref['x'].astype(np.float64)).max() <= 2e-5 * scale
    assert np.abs(sc - ref['scale_tot'].astype(np.float64)).max() <= 2e-5 * max(1.0, float(ref['scale_tot'].max()))
    one = eng.iaf_generate(mel[batch - 1:], noise[batch - 1:], want=('x',))
    assert torch.equal(one['x'][0], out['x'][batch - 1])
    eng.close()


def test_config0_fastgen_on_the_reference_fixture_mel():
    """wavenet_mol.json as shipped on the mel of the reference's own test utterance: fastgen.encode (F*200 = 154 600 conditioning steps),
    K1 on a 2 048-step prefix (incremental step == full-sequence teacher == float64 oracle), then the free-running
    loop over ALL 154 600 steps: length, 2^-15 grid and range of the reference's gen_LJ001-0001.wav (K4, K5)."""
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.auxilaries import mel_extractor as M
    from nsynth_wavenet_amd.engine import Engine
    import torch
    mel_np, wav_head = _fixture_mel()
    mel = torch.from_numpy(mel_np).cuda()
    cfgd = load_json('wavenet_mol.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'teacher', seed=1234, init='unit')
    eng = Engine(cfgd).load_weights(w)
    Tn = eng.ar_length(773)
    facts = np.load(os.path.join(GOLD, 'ref_fixture_facts.npz'))
    assert Tn == 154600 == int(facts['pred_data-no_mu_law+mol/gen_LJ001-0001.wav/n'])
    enc = eng.deconv(mel)
    assert tuple(enc.shape) == (1, Tn, 256)
    # K1 on a prefix: conditioning of the first 11 frames, centre-cropped to 2 048 steps like Wavenet.feed_forward
    P, Fp = 2048, 11
    left = (Fp * 200 - P) // 2
    mel_p = mel[:, :Fp]
    enc_p = _np(eng.deconv(mel_p))[:, left:left + P]
    forced = wav_head[:, :P]                       # the utterance's own first 2 048 samples
    rnd = np.random.RandomState(777).uniform(1e-5, 1 - 1e-5, [P, 1, eng.ar_n_rand()]).astype(np.float32)
    inc = _np(eng.ar_generate(enc_p, rnd, forced_wav=forced, want_out=True)['out_params'])
    full = _np(eng.teacher_forward(forced, mel_p))
    enc_o = O.deconv_stack(_np(mel_p), w, hp, '', np.float64)[:, left:left + P]
    ref = O.teacher_feed_forward(O.encode_signal(forced, hp, np.float64), enc_o, w, hp, np.float64)
    sc = max(1.0, np.abs(ref).max())
    assert np.abs(inc - ref).max() <= 5e-5 * sc and np.abs(full - ref).max() <= 5e-5 * sc
    # the whole utterance, free running, randoms drawn on the device
    out = eng.ar_generate(enc, None, seed=3)
    idx, wv = _np(out['idx']), _np(out['wav']).astype(np.float64)
    assert idx.shape == (1, Tn) and wv.shape == (1, Tn)
    assert idx.min() >= -32768 and idx.max() <= 32767
    assert np.all(wv * 32768 == np.round(wv * 32768)) and wv.min() >= -1 and wv.max() <= 1 - 2.0 ** -15
    assert np.array_equal(idx, (wv * 32768).astype(np.int32))
    assert len(np.unique(idx)) > 100                                          # a signal, not a stuck loop
    eng.close()


def test_config4_gauss_student_as_shipped_batch16():
    """parallel_wavenet_gauss.json unmodified (four private deconv stacks, N(0,1) noise), 16 utterances of
    F = 384: K2, determinism, every checked row equal to the single-utterance call, parity of one row with
    the torch-CPU implementation."""
    import torch
    from oracle import wavenet_np as O
    from oracle.torch_ref import StudentRef
    from nsynth_wavenet_amd.engine import Engine
    cfgd = load_json('parallel_wavenet_gauss.json')
    hp = O.HP(cfgd)
    assert not hp.get('use_share_deconv', False)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    assert 'iaf_4/trans_conv_2/kernel' in w and 'iaf_share/trans_conv_1/kernel' not in w
    eng = Engine(cfgd).load_weights(w)
    B, F, T = 16, 384, 76800
    mel = np.random.RandomState(12345).uniform(0, 1, [B, F, 80]).astype(np.float32)
    noise = np.random.RandomState(12346).standard_normal([B, T]).astype(np.float32)
    a = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot', 'rand_input', 'wav', 'idx'))
    b = eng.iaf_generate(mel, noise, want=('x',))
    assert torch.equal(a['x'], b['x'])
    x, m, s, r = (_np(a[k]).astype(np.float64) for k in ('x', 'mean_tot', 'scale_tot', 'rand_input'))
    assert x.shape == (B, T) and np.all(np.isfinite(x)) and np.all(s > 0)
    assert np.array_equal(_np(a['rand_input']), noise)
    scale = max(1.0, np.abs(x).max())
    assert np.abs(x - (r * s + m)).max() <= 1e-6 * scale
    wv = _np(a['wav']).astype(np.float64)
    assert np.all(wv * 32768 == np.round(wv * 32768))
    assert np.array_equal(_np(a['idx']), (wv * 32768).astype(np.int32))
    for row in (0, 7, 15):
        one = _np(eng.iaf_generate(mel[row:row + 1], noise[row:row + 1], want=('x',))['x'])
        assert np.abs(one[0] - x[row]).max() <= 5e-6 * scale
    # device-drawn Gaussian noise: rows differ, the call is reproducible per seed
    c = eng.iaf_generate(mel, None, seed=11, want=('rand_input', 'x'))
    d = eng.iaf_generate(mel, None, seed=11, want=('x',))
    assert torch.equal(c['x'], d['x'])
    rr = _np(c['rand_input']).astype(np.float64)
    assert abs(rr.mean()) < 0.01 and abs(rr.var() - 1.0) < 0.01 and not np.array_equal(rr[0], rr[1])
    eng.close()
    ref_x = StudentRef(w, hp).feed_forward(mel[3:4], noise[3:4])['x'].astype(np.float64)
    assert np.abs(x[3:4] - ref_x).max() <= 2e-5 * scale


def test_student_loads_through_a_tf_bundle_and_the_cli(tmp_path, monkeypatch):
    """Row f2 on the GPU: the student reaches the engine THROUGH a TensorFlow V2 checkpoint bundle, not an .npz.
    The bundle is hand-assembled (tests/bundle_assembler.py, never tf_bundle.write_bundle): two data shards, several
    prefix-compressed index blocks, '<var>/ExponentialMovingAverage' keys, the teacher-owned upsampler under its RAW
    names (use_teacher_deconv, parallelgen.py:31-39), a `global_step`, and three tensors stored with a size-equal
    but DIFFERENT shape -- what Saver(var_dict, reshape=True) accepts (parallelgen.py:40).  eval_parallel_wavenet.py
    runs on it (checkpoint state file, single *.json, .npy mel in, gen_<name>.wav out) and the written audio is held
    to the float64 oracle on the same weights and the same noise."""
    import json
    import torch
    from scipy.io import wavfile
    import bundle_assembler as ba
    import eval_parallel_wavenet as cli_main
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd import cli, config as cfg, weights as wts
    from nsynth_wavenet_amd.engine import Engine
    cfgd = dict(load_json('parallel_wavenet.json'), num_iaf_layers=[10, 10], use_teacher_deconv=True)
    cfgd.pop('use_share_deconv', None)
    hp = cfg.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=77, init='tf')
    raw = wts.raw_name_variables(w, hp)
    assert raw == {k for k in w if k.startswith('iaf_share/trans_conv')} and len(raw) == 4
    stored = {}
    for k, v in w.items():
        key = k if k in raw else k + wts.EMA
        a = np.asarray(v, '<f4')
        if k == 'iaf_1/out2_scale/W':
            a = a.reshape(64, 1)                       # [1,1,64,1] stored as [64,1]
        elif k == 'iaf_2/dilated_conv_3/W':
            a = a.reshape(3 * 64, 64)                  # [1,3,64,64] stored as [192,64]
        elif k == 'iaf_share/trans_conv_1/kernel':
            a = a.reshape(-1)                          # raw-name deconv kernel stored flat
        stored[key] = a
    stored['global_step'] = np.array(31337, '<i8')
    ck = tmp_path / 'ckpt'
    ck.mkdir()
    prefix = str(ck / 'model.ckpt-31337')
    ba.assemble(prefix, stored, shard_of={k: (1 if 'iaf_2' in k else 0) for k in stored}, n_shards=2, blocks=7,
                restart_every=3)
    (ck / 'checkpoint').write_text('model_checkpoint_path: "model.ckpt-31337"\nall_model_checkpoint_paths: "model.ckpt-31337"\n')
    (ck / 'parallel_wavenet.json').write_text(json.dumps(cfgd))
    assert not any(f.endswith('.npz') for f in os.listdir(ck))
    # the loader's view of the bundle: every variable, reshaped back
    hp2, ckpath = cli.resolve_model(str(ck))
    assert ckpath == prefix
    back = wts.load_checkpoint(ckpath, hp2)
    assert set(back) == set(w) and all(back[k].shape == w[k].shape and np.array_equal(back[k], w[k]) for k in w)
    # the CLI, in process, with the "unseeded" draw of the reference pinned so that the noise can be reproduced
    src = tmp_path / 'in'
    src.mkdir()
    F = 24
    mel = np.random.RandomState(5).uniform(0, 1, [F, 80]).astype(np.float32)
    np.save(src / 'utt.npy', mel)
    out = tmp_path / 'out'
    np.random.seed(2468)
    seed = int(np.random.RandomState(2468).randint(0, 2 ** 31 - 1))
    args = cli.build_parser('t').parse_args(['--ckpt_dir', str(ck), '--source_path', str(src), '--save_path', str(out)])
    cli_main.generate(args)
    rate, audio = wavfile.read(out / 'gen_utt.wav')
    T = O.iaf_length(F, O.HP(cfgd))
    assert rate == 16000 and audio.dtype == np.float32 and audio.shape == (T,)
    eng = Engine(cfgd).load_weights(w)                # same weights from memory: the noise of that seed, and the wav
    ref_run = eng.iaf_generate(mel[None], None, seed=seed, want=('wav', 'rand_input'))
    assert np.array_equal(_np(ref_run['wav'])[0], audio)
    noise = _np(ref_run['rand_input'])
    want = O.iaf_feed_forward(mel[None], noise, w, O.HP(cfgd), np.float64)
    wav_ref, _ = O.clip_quant_scale(want['x'], 65536, False, np.float64)
    assert np.abs(audio - wav_ref[0]).max() <= 1e-3 and np.mean(audio != wav_ref[0].astype(np.float32)) < 0.02
    eng.close()
    torch.cuda.synchronize()


def test_pipelined_cli_writes_what_the_serial_loop_writes(tmp_path):
    """eval_parallel_wavenet.py over a directory of wavs (ragged lengths, batches of 3, a short last batch): the overlapped
    reader | GPU | writer stages (cli.run_pipelined, the default) write, file for file and bit for bit, what the reference's
    one-batch-at-a-time loop (eval_parallel_wavenet.py:52-69; --serial) writes, with the reference's unseeded draws pinned."""
    import json
    import torch
    from scipy.io import wavfile
    import eval_parallel_wavenet as cli_main
    from nsynth_wavenet_amd import cli, config as cfg, weights as wts
    cfgd = dict(load_json('parallel_wavenet.json'), num_iaf_layers=[10, 10])
    hp = cfg.load_hparams(cfgd)
    ck = tmp_path / 'ckpt'
    ck.mkdir()
    wts.save_checkpoint(str(ck / 'model.ckpt-7'), wts.synthetic_weights(hp, seed=5, init='tf'), hp)
    (ck / 'parallel_wavenet.json').write_text(json.dumps(cfgd))
    src = tmp_path / 'wavs'
    src.mkdir()
    rs = np.random.RandomState(9)
    lens = [4700, 5200, 3900, 6100, 5000, 4400, 5600]
    for i, n in enumerate(lens):
        y = 0.3 * np.sin(2 * np.pi * (200.0 + 40 * i) * np.arange(n) / 16000.0) + 0.02 * rs.standard_normal(n)
        wavfile.write(str(src / ('u%02d.wav' % i)), 16000, (np.clip(y, -1, 1) * 32767).astype(np.int16))
    outs = {}
    for mode in ('serial', 'pipelined'):
        dst = tmp_path / ('out_' + mode)
        argv = ['--ckpt_dir', str(ck), '--source_path', str(src), '--save_path', str(dst), '--batch_size', '3']
        np.random.seed(4242)
        cli_main.generate(cli.build_parser('t').parse_args(argv + (['--serial'] if mode == 'serial' else [])))
        torch.cuda.synchronize()
        outs[mode] = {f: wavfile.read(str(dst / f))[1] for f in sorted(os.listdir(dst))}
    assert sorted(outs['serial']) == ['gen_u%02d.wav' % i for i in range(len(lens))] == sorted(outs['pipelined'])
    for f, a in outs['serial'].items():
        b = outs['pipelined'][f]
        assert a.dtype == np.float32 and a.shape == b.shape and a.size > 0 and np.array_equal(a, b), f
    # a batch is padded to its longest utterance (fastgen.load_batch): files of one batch share a length, a multiple of 512
    assert len({outs['serial']['gen_u%02d.wav' % i].shape for i in (0, 1, 2)}) == 1
    assert all(a.shape[0] % 512 == 0 for a in outs['serial'].values())
