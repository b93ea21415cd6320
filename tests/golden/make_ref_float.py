"""Float-path vectors produced by the REFERENCE'S OWN Python, executed in the build container over a numpy evaluator
for the TensorFlow primitives (tests/golden/tf_standin.py -- read its header for what that does and does not pin).

Run in the build container only (needs /root/reference; nothing of it travels -- only the .npz below):
    python tests/golden/make_ref_float.py
Writes tests/golden/ref_float.npz;  `--full` writes tests/golden/ref_float_full.npz (BASELINE configs[1] at full size) instead.

For every case of tests/golden/make_golden.py (same configs, same mel / random seeds, same synthetic weights) the
reference's files are imported unmodified from /root/reference and DRIVEN THE WAY THE REFERENCE DRIVES THEM:

  student  wavenet/parallelgen.py: load_parallelgen() builds ParallelWavenet.feed_forward + _clip_quant_scale on a
           placeholder; synthesis() restores the checkpoint through its own Saver map, runs the session and writes the
           wav files.  Recorded: the noise the graph drew (injected uniforms / normals), mean_tot, scale_tot, the
           pre-quantisation x = rand_input * scale_tot + mean_tot, the quantised audio, the wav file synthesis() wrote,
           a strided sample of the upsampler output, the variables the graph created and the checkpoint keys its
           Saver asked for.
  teacher  wavenet/fastgen.py: load_deconv_stack() (the encoding), Wavenet.encode_signal + Wavenet.feed_forward on a
           forced waveform (the full-sequence teacher), Fastgen.cond_vars, load_fastgen() + the sample-by-sample loop
           of synthesis() (FIFO queues, push ops, the de-quantised feedback) with the sampler's randoms injected;
           synthesis() itself is run as written and the wav files it writes are recorded.

Each case is evaluated twice: with tf.float32 mapped to float64 (master copy, `*_f64`) and to float32 (the arithmetic
TensorFlow would run, `*_f32`).  The checkpoint the reference restores from is written by the PRODUCT's
nsynth_wavenet_amd.weights.save_checkpoint, so a key the reference's Saver asks for and the product does not write
fails here.
"""
import json
import os
import sys
import tempfile
import types
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
OUT = os.path.join(HERE, 'ref_float.npz')
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf_standin as tf  # noqa: E402
from oracle import wavenet_np as O  # noqa: E402   (synthetic weights only: the same dict the other goldens use)
from nsynth_wavenet_amd import weights as wts, config as cfgmod  # noqa: E402   (checkpoint writer under test)


class _Inert(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Inert(self.__name__ + '.' + name)

    def __call__(self, *a, **k):
        raise RuntimeError('placeholder for an absent dependency was CALLED: ' + self.__name__)


def import_reference():
    tf.install()
    for name in ('librosa', 'librosa.filters'):
        sys.modules[name] = _Inert(name)
    sys.path.insert(0, REF)
    import importlib
    mods = {}
    for m in ('wavenet.masked', 'wavenet.loss_func', 'wavenet.wavenet', 'wavenet.parallel_wavenet', 'wavenet.fastgen',
              'wavenet.parallelgen', 'auxilaries.utils'):
        mods[m.split('.')[1]] = importlib.import_module(m)
        assert os.path.realpath(mods[m.split('.')[1]].__file__).startswith(REF)
    return Namespace(**mods)


class TableSource(object):
    """random_source for the stand-in: node index -> function(step) -> array; one step per evaluation of the node."""

    def __init__(self, table):
        self.table, self.count = table, {}

    def __call__(self, index, kind, shape, run_no, lo, hi):
        step = self.count.get(index, 0)
        self.count[index] = step + 1
        v = np.asarray(self.table[index](step))
        assert list(v.shape) == list(shape), (index, v.shape, shape)
        if kind == 'uniform':
            assert v.min() >= lo and v.max() < hi, (index, v.min(), v.max(), lo, hi)
        return v


def reference_hparams(cfgd, kind):
    """The reference's own JSON of that model (it carries the training keys the classes read in __init__: num_iters,
    wave_length), overlaid with the case's settings -- Namespace(**configs) as eval_*.py:28-30 build it."""
    lt = cfgd['loss_type']
    name = ('parallel_wavenet_gauss.json' if lt == 'gauss' else 'parallel_wavenet.json') if kind == 'student' \
        else 'wavenet_{}.json'.format(lt)
    with open(os.path.join(REF, 'config_jsons', name)) as f:
        d = json.load(f)
    d.update(cfgd)
    return Namespace(**d)


def var_list():
    return [(v.name[:-2], list(v.sample.shape)) for v in tf.trainable_variables()]


def find_bias_add(root, bias_name):
    """The bias_add node of the variable scope `bias_name` upstream of `root` (the reference returns only the sample)."""
    seen, stack = set(), [root]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        if n.name == 'bias_add' and n.inputs[1].kind == 'var' and n.inputs[1].full == bias_name:
            return n
        stack.extend(n.inputs)
    raise KeyError(bias_name)


def student_case(R, g, out, tag, tmp, w=None, floats=(('f64', np.float64), ('f32', np.float32))):
    cfgd = json.loads(str(g['cfg_json']))
    hp_o = O.HP(cfgd)
    if w is None:
        w = O.synth_weights(hp_o, 'student', seed=int(g['seed']), init=str(g['init']))
    ckpt = wts.save_checkpoint(os.path.join(tmp, tag + '.npz'), w, cfgmod.load_hparams(cfgd))
    mel = g['mel']
    B, F, _ = mel.shape
    T = O.iaf_length(F, hp_o)
    rs = np.random.RandomState(12346)            # make_golden.iaf_case draws its noise from the same stream
    gauss = cfgd['loss_type'] == 'gauss'
    draw = rs.standard_normal([B, T]) if gauss else rs.uniform(1e-5, 1 - 1e-5, [B, T])
    draw = draw.astype(np.float32).astype(np.float64)          # float32-valued, so both arithmetics see the same numbers
    for fl, dt in floats:
        tf.set_float(dt)
        tf.Saver.requested = []
        hparams = reference_hparams(cfgd, 'student')
        # -- parallelgen.synthesis as written (restores through its own Saver map, writes the wav files)
        tf.set_random_source(TableSource({0: lambda step: draw}))
        paths = [os.path.join(tmp, '{}_{}_{}.wav'.format(tag, fl, b)) for b in range(B)]
        R.parallelgen.synthesis(hparams, mel, paths, ckpt)
        from scipy.io import wavfile
        wav_files = np.stack([wavfile.read(p)[1] for p in paths])
        requested = list(tf.Saver.requested[-1])
        # -- the same graph once more, to fetch what synthesis() does not return
        tf.set_random_source(TableSource({0: lambda step: draw}))
        with tf.Graph().as_default(), tf.Session() as sess:
            fg = R.parallelgen.load_parallelgen(hparams, B, F, mel.shape[2])
            vl = var_list()
            pw = R.parallel_wavenet.ParallelWavenet(hparams)
            share = pw.use_share_deconv or pw.use_teacher_deconv
            enc_t = pw.deconv_stack({'mel': fg['mel_in']}, name='iaf_share' if share else 'iaf_1')['encoding']
            assert var_list() == vl                            # AUTO_REUSE: no new variables
            tf_vars = tf.trainable_variables()                 # the restore map of parallelgen.py:29-41, from its own helpers
            filtered = pw.filter_update_variables(tf_vars)
            var_dict = R.fastgen.get_ema_shadow_dict(filtered)
            var_dict.update(R.parallelgen.get_default_shadow_dict([v for v in tf_vars if v not in filtered]))
            tf.train.Saver(var_dict, reshape=True).restore(sess, ckpt)
            assert sorted(tf.Saver.requested[-1]) == sorted(requested)
            vals = sess.run({k: fg[k] for k in ('x', 'mean_tot', 'scale_tot', 'log_scale_tot', 'rand_input')} |
                            {'enc': enc_t}, feed_dict={fg['mel_in']: mel})
        assert np.array_equal(wav_files.astype(np.float64), vals['x'].astype(np.float64)), 'synthesis() wav != fetched x'
        x_pre = vals['rand_input'] * vals['scale_tot'] + vals['mean_tot']      # parallel_wavenet.py:326 (new_x)
        for k in ('mean_tot', 'scale_tot', 'log_scale_tot', 'rand_input'):
            out['{}/{}_{}'.format(tag, k, fl)] = vals[k]
        out['{}/x_{}'.format(tag, fl)] = x_pre
        out['{}/wav_{}'.format(tag, fl)] = vals['x'].astype(np.float32)
        out['{}/enc_sub_{}'.format(tag, fl)] = vals['enc'][:, ::7, ::5]
        print(tag, fl, 'T', T, '|x| max', float(np.abs(x_pre).max()), 'scale_tot', float(vals['scale_tot'].min()),
              float(vals['scale_tot'].max()))
    out[tag + '/vars'] = np.array(json.dumps(vl))
    out[tag + '/ckpt_keys'] = np.array(json.dumps(requested))
    out[tag + '/kind'] = np.array('student')


def teacher_case(R, g, out, tag, tmp, w=None):
    cfgd = json.loads(str(g['cfg_json']))
    hp_o = O.HP(cfgd)
    if w is None:
        w = O.synth_weights(hp_o, 'teacher', seed=int(g['seed']), init=str(g['init']))
    ckpt = wts.save_checkpoint(os.path.join(tmp, tag + '.npz'), w, cfgmod.load_hparams(cfgd))
    mel, rnd, forced = g['mel'], g['rnd'], g['forced']
    B, F, n_mel = mel.shape
    Tn = forced.shape[1]
    M = cfgd.get('mol_mix', 10)
    lt = cfgd['loss_type']
    if lt == 'mol':
        table = {0: lambda t: rnd[t][:, :M].reshape(B, 1, M), 1: lambda t: rnd[t][:, M].reshape(B, 1)}
    elif lt == 'gauss':
        table = {0: lambda t: rnd[t][:, 0].reshape(B, 1)}
    else:
        table = {0: lambda t: rnd[t][:, 0].reshape(1, B, 1)}
    Q = 256 if cfgd['use_mu_law'] else 65536
    from scipy.io import wavfile
    for fl, dt in (('f64', np.float64), ('f32', np.float32)):
        tf.set_float(dt)
        tf.Saver.requested = []
        hparams = reference_hparams(cfgd, 'teacher')
        ema = R.fastgen.get_ema_shadow_dict
        # -- the encoding: fastgen.encode without its librosa front end (load_deconv_stack + the restore lines)
        with tf.Graph().as_default(), tf.Session() as sess:
            ds = R.fastgen.load_deconv_stack(hparams, B, F, n_mel)
            tf.train.Saver(ema(tf.trainable_variables())).restore(sess, ckpt)
            enc = sess.run(ds['encoding'], feed_dict={ds['mel_in']: mel})
        assert enc.shape[1] == Tn
        # -- the full-sequence teacher on a forced waveform (wavenet.py:157-291)
        with tf.Graph().as_default(), tf.Session() as sess:
            # dropout off the way the reference switches it off for a teacher it evaluates (train_parallel_wavenet.py:37)
            wn = R.wavenet.Wavenet(Namespace(**dict(vars(hparams), use_as_teacher=True)))
            wav_ph = tf.placeholder(tf.float32, [B, Tn])
            mel_ph = tf.placeholder(tf.float32, [B, F, n_mel])
            es = wn.encode_signal({'wav': wav_ph})
            ff = wn.feed_forward({'mel': mel_ph, 'wav_scaled': es['wav_scaled']})
            vl = var_list()
            tf.train.Saver(ema(tf.trainable_variables())).restore(sess, ckpt)
            requested_ff = list(tf.Saver.requested[-1])
            # teacher scoring: Wavenet.calculate_loss as written (wavenet.py:293-316) and its per-sample term
            loss_t = wn.calculate_loss({'real_targets': es['real_targets'], 'cate_targets': es['cate_targets'],
                                        'out_params': ff['out_params']})['loss']
            if lt == 'mol':
                logp_t = R.loss_func.mol_log_probs(ff['out_params'], es['real_targets'], Q)
            elif lt == 'gauss':
                logp_t = R.loss_func.gauss_log_prob(ff['out_params'], es['real_targets'])
            else:
                logp_t = -tf.nn.sparse_softmax_cross_entropy_with_logits(logits=ff['out_params'], labels=es['cate_targets'])
            out_forced, enc_ff, loss_v, logp_v = sess.run([ff['out_params'], ff['encoding'], loss_t, logp_t],
                                                          feed_dict={wav_ph: forced, mel_ph: mel})
        assert np.array_equal(enc_ff, enc)
        assert abs(float(loss_v) + float(np.mean(logp_v))) <= 1e-5 * max(1.0, abs(float(loss_v)))
        out['{}/logp_{}'.format(tag, fl)] = logp_v
        out['{}/loss_{}'.format(tag, fl)] = np.array(float(loss_v))
        # -- Fastgen.cond_vars through fastgen.calculate_cond_vars as written
        cond = R.fastgen.calculate_cond_vars(hparams, enc, ckpt)
        # -- the incremental sampler, free running: synthesis() as written ...
        enc32 = enc.astype(np.float32)
        tf.set_random_source(TableSource(table))
        paths = [os.path.join(tmp, '{}_{}_{}.wav'.format(tag, fl, b)) for b in range(B)]
        R.fastgen.synthesis(hparams, enc32, paths, ckpt)
        wav_files = np.stack([wavfile.read(p)[1] for p in paths])
        requested_fg = list(tf.Saver.requested[-1])
        # ... and its loop once more (fastgen.py:128-169) with the network output of every step fetched as well
        tf.set_random_source(TableSource(table))
        with tf.Graph().as_default(), tf.Session() as sess:
            fg = R.fastgen.load_fastgen(hparams, B)
            vl_fg = var_list()
            out_node = find_bias_add(fg['sample'], 'out2/biases')
            tf.train.Saver(ema(tf.trainable_variables())).restore(sess, ckpt)
            sess.run(fg['init_ops'])
            audio = np.zeros([B, 1])
            idx = np.zeros([B, Tn], np.int32)
            outs = []
            wav = np.zeros([B, Tn], np.float32)
            for t in range(Tn):
                q, o, _ = sess.run([fg['sample'], out_node, fg['push_ops']],
                                   feed_dict={fg['wav_in']: audio, fg['encoding_in']: enc32[:, t, :]})
                audio = (R.utils.inv_mu_law_numpy(q) if cfgd['use_mu_law'] else R.utils.inv_cast_quantize_numpy(q, Q))
                idx[:, t] = q[:, 0]
                wav[:, t] = audio[:, 0]
                outs.append(o)
        assert np.array_equal(wav, wav_files), 'synthesis() wav != the re-run loop'
        # -- teacher forced through the incremental graph: K1 (incremental == full sequence) on the reference itself
        with tf.Graph().as_default(), tf.Session() as sess:
            tf.set_random_source(TableSource(table))
            fg = R.fastgen.load_fastgen(hparams, B)
            out_node = find_bias_add(fg['sample'], 'out2/biases')
            tf.train.Saver(ema(tf.trainable_variables())).restore(sess, ckpt)
            sess.run(fg['init_ops'])
            prev = np.zeros([B, 1])
            inc = []
            for t in range(Tn):
                o, _ = sess.run([out_node, fg['push_ops']], feed_dict={fg['wav_in']: prev, fg['encoding_in']: enc[:, t, :]})
                inc.append(o)
                prev = forced[:, t:t + 1]
        inc = np.stack(inc, axis=1)
        k1 = float(np.abs(inc - out_forced).max())
        out['{}/enc_{}'.format(tag, fl)] = enc
        out['{}/out_forced_{}'.format(tag, fl)] = out_forced
        out['{}/free_idx_{}'.format(tag, fl)] = idx
        out['{}/free_wav_{}'.format(tag, fl)] = wav
        out['{}/free_out_{}'.format(tag, fl)] = np.stack(outs, axis=1)
        out['{}/k1_{}'.format(tag, fl)] = np.array(k1)
        names = sorted(cond.keys())
        out['{}/cond_sub_{}'.format(tag, fl)] = np.stack([cond[k][:, ::5, ::7] for k in names if k != 'mel_cond_out1'])
        out['{}/cond_out1_sub_{}'.format(tag, fl)] = cond['mel_cond_out1'][:, ::5, ::7]
        print(tag, fl, 'Tn', Tn, 'K1 (incremental vs full sequence, reference code)', k1)
    assert set(requested_fg) < set(requested_ff) and all('trans_conv' in k or 'resize_conv' in k
                                                         for k in set(requested_ff) - set(requested_fg))
    assert all(v in vl for v in vl_fg)
    out[tag + '/ckpt_keys_fastgen'] = np.array(json.dumps(requested_fg))
    out[tag + '/vars'] = np.array(json.dumps(vl))
    out[tag + '/ckpt_keys'] = np.array(json.dumps(requested_ff))
    out[tag + '/kind'] = np.array('teacher')


def repo_cfg(name):
    with open(os.path.join(ROOT, 'config_jsons', name)) as f:
        return json.load(f)


def extra_cases(R, out, tmp):
    """Rows a13 / a14 beyond make_golden's cases: weight normalisation + resize-conv upsampler (student and teacher), and a
    student on the teacher's upsampler (use_teacher_deconv: those variables are restored under their RAW names,
    parallelgen.py:29-41).  Weights: nsynth_wavenet_amd.weights.synthetic_weights(hp, seed=9, init='unit'); the inputs are
    stored in the fixture."""
    small = dict(width=128, skip_width=64, deconv_width=64, num_layers=7, num_stages=3, deconv_config=[[8, 2], [12, 4]])
    cases = [
        ('iaf_wn_resize', 'student', dict(repo_cfg('parallel_wavenet.json'), use_weight_norm=True, use_resize_conv=True,
                                          upsample_act='tanh', num_iaf_layers=[10, 10]), 1, 6),
        ('iaf_teacher_deconv', 'student', dict(repo_cfg('parallel_wavenet.json'), use_share_deconv=False,
                                               use_teacher_deconv=True, num_iaf_layers=[10]), 2, 6),
        ('ar_wn_resize', 'teacher', dict(repo_cfg('wavenet_mol.json'), use_weight_norm=True, use_resize_conv=True,
                                         upsample_act='tanh', **small), 2, 5),
    ]
    for tag, kind, cfgd, B, F in cases:
        hp = cfgmod.load_hparams(cfgd)
        w = wts.synthetic_weights(hp, seed=9, init='unit')
        g = {'cfg_json': np.array(json.dumps(cfgd)), 'seed': np.array(9), 'init': np.array('unit'),
             'mel': np.random.RandomState(21).uniform(0, 1, [B, F, 80]).astype(np.float32)}
        if kind == 'teacher':
            Tn = F * int(np.prod([c[1] for c in cfgd['deconv_config']]))
            n_rand = cfgd.get('mol_mix', 10) + 1
            g['rnd'] = np.random.RandomState(22).uniform(1e-5, 1 - 1e-5, [Tn, B, n_rand]).astype(np.float32)
            g['forced'] = np.random.RandomState(23).uniform(-1, 1, [B, Tn]).astype(np.float32)
            teacher_case(R, g, out, tag, tmp, w)
        else:
            student_case(R, g, out, tag, tmp, w)
        for k, v in g.items():
            out['{}/in_{}'.format(tag, k)] = v


def teacher_full_width_case(R, out, tmp):
    """BASELINE.json configs[0] / [3]: wavenet_mol.json AS SHIPPED (width 512, 30 layers, MoL-10) through the reference's
    incremental graph (load_fastgen + the loop of fastgen.synthesis, and synthesis() itself), teacher-forced and free
    running, float64, 400 steps on a seeded encoding (the upsampler at these widths is the students' iaf_share case)."""
    from scipy.io import wavfile
    tag = 'ar_mol_full'
    cfgd = repo_cfg('wavenet_mol.json')
    hp = cfgmod.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=1234, init='unit')
    ckpt = wts.save_checkpoint(os.path.join(tmp, tag + '.npz'), w, hp)
    B, Tn, M = 2, 400, cfgd['mol_mix']
    rs = np.random.RandomState(41)
    enc32 = (rs.standard_normal([B, Tn, cfgd['deconv_width']]) * 0.3).astype(np.float32)
    rnd = rs.uniform(1e-5, 1 - 1e-5, [Tn, B, M + 1]).astype(np.float32)
    forced = rs.uniform(-1, 1, [B, Tn]).astype(np.float32)
    table = {0: lambda t: rnd[t][:, :M].reshape(B, 1, M), 1: lambda t: rnd[t][:, M].reshape(B, 1)}
    tf.set_float(np.float64)
    hparams = reference_hparams(cfgd, 'teacher')
    tf.set_random_source(TableSource(table))
    paths = [os.path.join(tmp, '{}_{}.wav'.format(tag, b)) for b in range(B)]
    R.fastgen.synthesis(hparams, enc32, paths, ckpt)
    wav_files = np.stack([wavfile.read(p)[1] for p in paths])
    res = {}
    for mode in ('free', 'forced'):
        tf.set_random_source(TableSource(table))
        with tf.Graph().as_default(), tf.Session() as sess:
            fg = R.fastgen.load_fastgen(hparams, B)
            out_node = find_bias_add(fg['sample'], 'out2/biases')
            tf.train.Saver(R.fastgen.get_ema_shadow_dict(tf.trainable_variables())).restore(sess, ckpt)
            sess.run(fg['init_ops'])
            audio = np.zeros([B, 1])
            idx, outs = np.zeros([B, Tn], np.int32), []
            for t in range(Tn):
                q, o, _ = sess.run([fg['sample'], out_node, fg['push_ops']],
                                   feed_dict={fg['wav_in']: audio, fg['encoding_in']: enc32[:, t, :]})
                audio = R.utils.inv_cast_quantize_numpy(q, 65536) if mode == 'free' else forced[:, t:t + 1]
                idx[:, t] = q[:, 0]
                outs.append(o)
        res[mode] = (idx, np.stack(outs, axis=1))
    assert np.array_equal(res['free'][0] / 32768.0, wav_files.astype(np.float64)), 'synthesis() wav != the re-run loop'
    out[tag + '/free_idx_f64'] = res['free'][0]
    out[tag + '/free_out_f64'] = res['free'][1].astype(np.float32)
    out[tag + '/out_forced_f64'] = res['forced'][1]
    out[tag + '/in_cfg_json'] = np.array(json.dumps(cfgd))
    out[tag + '/kind'] = np.array('teacher')
    print(tag, 'Tn', Tn, 'out range', float(np.abs(res['forced'][1]).max()))


def k8_cases(R, out, tmp):
    """The shapes and seeds of the reference's own hot-path tests, through the reference's code, float64:
    tests/test_parallel_wavenet.py:25-31 -- four utterances of 39 frames = 7 680 samples, the student JSONs AS SHIPPED
    (parallel_wavenet.json: shared upsampler; parallel_wavenet_gauss.json: four private ones, normal noise);
    tests/test_fastgen.py:17-32 -- ONE step of Fastgen.sample on wavenet_mol.json as shipped, batch 4, wav [4,1] and
    encoding [4,256] drawn U(-1,1) after np.random.seed(12345) exactly as that test draws them (its TF-initialised
    weights are replaced by the synthetic ones below; the reference prints the sample, here the network output is kept)."""
    for tag, name in (('k8_pw', 'parallel_wavenet.json'), ('k8_pw_gauss', 'parallel_wavenet_gauss.json')):
        cfgd = repo_cfg(name)
        hp = cfgmod.load_hparams(cfgd)
        w = wts.synthetic_weights(hp, seed=1234, init='unit')
        g = {'cfg_json': np.array(json.dumps(cfgd)), 'seed': np.array(1234), 'init': np.array('unit'),
             'mel': np.random.RandomState(51).uniform(0, 1, [4, 39, 80]).astype(np.float32)}
        tmp_out = {}
        student_case(R, g, tmp_out, tag, tmp, w, floats=(('f64', np.float64),))
        for k in ('x_f64', 'vars', 'ckpt_keys', 'kind'):
            out['{}/{}'.format(tag, k)] = tmp_out['{}/{}'.format(tag, k)]
        # the noise is not stored: the tests recompute it from the seeded float32 draws student_case injected
        rs = np.random.RandomState(12346)
        d = (rs.standard_normal([4, 7680]) if cfgd['loss_type'] == 'gauss' else rs.uniform(1e-5, 1 - 1e-5, [4, 7680]))
        d = d.astype(np.float32).astype(np.float64)
        d = d if cfgd['loss_type'] == 'gauss' else np.log(d) - np.log(1.0 - d)
        assert np.abs(d - tmp_out[tag + '/rand_input_f64']).max() <= 1e-14
        out[tag + '/scale_tot_f32'] = tmp_out[tag + '/scale_tot_f64'].astype(np.float32)
        for k, v in g.items():
            out['{}/in_{}'.format(tag, k)] = v
    tag = 'k8_fastgen'
    cfgd = repo_cfg('wavenet_mol.json')
    hp = cfgmod.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=1234, init='unit')
    ckpt = wts.save_checkpoint(os.path.join(tmp, tag + '.npz'), w, hp)
    B = 4
    np.random.seed(12345)                                   # tests/test_fastgen.py:28-30
    wav_val = np.random.uniform(-1, 1, [B, 1])
    mel_en_val = np.random.uniform(-1, 1, [B, cfgd['deconv_width']])
    M = cfgd['mol_mix']
    rnd = np.random.RandomState(52).uniform(1e-5, 1 - 1e-5, [1, B, M + 1]).astype(np.float32)
    tf.set_float(np.float64)
    tf.set_random_source(TableSource({0: lambda t: rnd[t][:, :M].reshape(B, 1, M), 1: lambda t: rnd[t][:, M].reshape(B, 1)}))
    with tf.Graph().as_default(), tf.Session() as sess:
        fgen = R.wavenet.Fastgen(reference_hparams(cfgd, 'teacher'), B)          # tests/test_fastgen.py:17-27
        wav_ph = tf.placeholder(tf.float32, [B, 1])
        mel_en_ph = tf.placeholder(tf.float32, [B, cfgd['deconv_width']])
        fg = fgen.sample({'wav': wav_ph, 'encoding': mel_en_ph})
        out_node = find_bias_add(fg['sample'], 'out2/biases')
        tf.train.Saver(R.fastgen.get_ema_shadow_dict(tf.trainable_variables())).restore(sess, ckpt)
        sess.run(fg['init_ops'])
        s_val, o_val, _ = sess.run([fg['sample'], out_node, fg['push_ops']], feed_dict={wav_ph: wav_val, mel_en_ph: mel_en_val})
    out[tag + '/wav'] = wav_val.astype(np.float32)
    out[tag + '/encoding'] = mel_en_val.astype(np.float32)
    out[tag + '/rnd'] = rnd
    out[tag + '/out_f64'] = o_val
    out[tag + '/sample'] = s_val.astype(np.int32)
    out[tag + '/in_cfg_json'] = np.array(json.dumps(cfgd))
    print(tag, 'sample', s_val[:, 0], 'out range', float(np.abs(o_val).max()))


def full_size_case(R, tmp):
    """BASELINE.json configs[1] at its full size -- parallel_wavenet.json as shipped, one utterance of 384 frames = 76 800
    samples, bench.py's weights (synthetic_weights(seed=1234, init='tf')) -- through parallelgen.synthesis as written, float64.
    Own file (ref_float_full.npz): noise and x in float64, the rest as float32."""
    cfgd = repo_cfg('parallel_wavenet.json')
    hp = cfgmod.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=1234, init='tf')
    g = {'cfg_json': np.array(json.dumps(cfgd)), 'seed': np.array(1234), 'init': np.array('tf'),
         'mel': np.random.RandomState(31).uniform(0, 1, [1, 384, 80]).astype(np.float32)}
    out = {}
    student_case(R, g, out, 'full', tmp, w, floats=(('f64', np.float64),))
    # kept small: the noise is NOT stored (the test recomputes log(u) - log(1 - u) from the same seeded float32 uniforms),
    # mean_tot follows from x, the noise and scale_tot, the audio is stored as the int16 index synthesis() wrote * 2^-15
    res = {'full/x_f64': out['full/x_f64'], 'full/scale_tot_f32': out['full/scale_tot_f64'].astype(np.float32)}
    idx = np.round(out['full/wav_f64'].astype(np.float64) * 32768.0)
    assert np.array_equal(idx / 32768.0, out['full/wav_f64']) and np.abs(idx).max() <= 32768
    res['full/idx_i16'] = idx.astype(np.int16)
    u = np.random.RandomState(12346).uniform(1e-5, 1 - 1e-5, [1, 76800]).astype(np.float32).astype(np.float64)
    assert np.abs((np.log(u) - np.log(1.0 - u)) - out['full/rand_input_f64']).max() <= 1e-14
    res['full/enc_sub_f32'] = out['full/enc_sub_f64'][:, ::16].astype(np.float32)
    for k, v in g.items():
        res['full/in_' + k] = v
    res['full/kind'] = np.array('student')
    p = os.path.join(HERE, 'ref_float_full.npz')
    np.savez_compressed(p, **res)
    print('wrote', p, os.path.getsize(p), 'bytes')


def main():
    R = import_reference()
    if '--full' in sys.argv:
        with tempfile.TemporaryDirectory() as tmp:
            full_size_case(R, tmp)
        return
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag in ('iaf_logistic_tf', 'iaf_logistic_unit', 'iaf_gauss_perflow', 'iaf_mulaw'):
            student_case(R, np.load(os.path.join(HERE, tag + '.npz')), out, tag, tmp)
        for tag in ('ar_mol', 'ar_ce_mulaw', 'ar_gauss'):
            teacher_case(R, np.load(os.path.join(HERE, tag + '.npz')), out, tag, tmp)
        extra_cases(R, out, tmp)
        teacher_full_width_case(R, out, tmp)
        k8_cases(R, out, tmp)
    out['numpy_version'] = np.array(np.__version__)
    # the float32-arithmetic twins are kept for the principal outputs only
    for k in [k for k in out if k.endswith('_f32') and k.split('/')[1].rsplit('_', 1)[0] not in
              ('x', 'mean_tot', 'scale_tot', 'rand_input', 'enc', 'out_forced', 'free_idx', 'free_wav', 'k1')]:
        del out[k]
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
