"""SURVEY 8(b) "Threading / streams": a finalized handle is immutable and may be shared by concurrent callers using distinct
workspaces and streams.  One C handle, two host threads, two streams, calls in flight at the same time -- for the parallel
student (wn_iaf_generate) and for the autoregressive teacher (wn_ar_generate, plain launches and hipGraph replay) -- and every
result is held to the SERIAL run of the same call bit for bit and to the float64 oracle.  The switches of a handle
(wn_iaf_set_groups, wn_ar_set_graph, wn_profile_*) are refused with WN_ESTATE while a work call of another thread is inside
the library, and the per-thread error message of one caller is never another caller's."""
import json
import os
import threading

import numpy as np
import pytest

from conftest import load_json

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _np(t):
    return t.detach().cpu().numpy()


def _run_threads(fns):
    """Start one thread per callable behind a barrier; re-raise the first exception of any of them."""
    errs, outs = [], [None] * len(fns)
    gate = threading.Barrier(len(fns))

    def wrap(i, fn):
        try:
            gate.wait()
            outs[i] = fn()
        except BaseException as e:               # noqa: B902 (reported to the main thread)
            errs.append(e)
    ths = [threading.Thread(target=wrap, args=(i, fn)) for i, fn in enumerate(fns)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return outs


def test_one_student_handle_two_threads_two_streams():
    """Two threads, each with its own fork (workspace) and stream, issue 12 generate calls each on ONE handle while the other
    is doing the same (different utterance lengths, one of them cropped): every call equals the serial call of the same
    inputs bit for bit, and the oracle at 2e-5 of the range."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    eng = Engine(cfgd).load_weights(w)
    cases = []
    for k, (B, F) in enumerate(((1, 64), (2, 23))):            # T = 12 800 (no crop) / 4 096 (crop 252)
        T = O.iaf_length(F, hp)
        mel = np.random.RandomState(100 + k).uniform(0, 1, [B, F, 80]).astype(np.float32)
        noise = O.logistic_from_uniform(np.random.RandomState(200 + k).uniform(1e-5, 1 - 1e-5, [B, T]))
        cases.append((mel, noise))
    serial = [{k: v.clone() for k, v in eng.iaf_generate(m, n, want=('x', 'idx', 'wav')).items()} for m, n in cases]
    torch.cuda.synchronize()
    for (m, n), s in zip(cases, serial):
        ref = O.iaf_feed_forward(m, n, w, hp, np.float64)['x']
        assert np.abs(_np(s['x']) - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())

    reps = 12

    def worker(i):
        def run():
            f = eng.fork()
            st = torch.cuda.Stream()
            got = []
            with torch.cuda.stream(st):
                m, n = torch.from_numpy(cases[i][0]).cuda(), torch.from_numpy(cases[i][1]).cuda()
                for _ in range(reps):
                    got.append(f.iaf_generate(m, n, want=('x', 'idx', 'wav'), check_range=False))
                f.check_range()
            st.synchronize()
            f.close()
            return got
        return run
    outs = _run_threads([worker(0), worker(1)])
    for i in range(2):
        for g in outs[i]:
            for k in ('x', 'idx', 'wav'):
                assert torch.equal(g[k], serial[i][k]), (i, k)
    eng.close()


def test_switches_are_refused_while_a_work_call_is_in_flight_and_errors_are_per_thread():
    """A thread keeps generate calls in flight; the main thread hammers the switches: each attempt either succeeds
    (between two calls) or returns WN_ESTATE -- and whatever happened, the generate results never change (a call reads
    its switches once, at entry).  wn_last_error is per thread: the worker's failing call does not leak into this one."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd import _lib
    from nsynth_wavenet_amd.engine import Engine
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    eng = Engine(cfgd).load_weights(O.synth_weights(hp, 'student', seed=1234, init='tf'))
    F = 128                                                    # T = 25 600: groups and per-layer forms both apply
    mel = np.random.RandomState(7).uniform(0, 1, [1, F, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(np.random.RandomState(8).uniform(1e-5, 1 - 1e-5, [1, O.iaf_length(F, hp)]))
    by_mode = {}
    for mode in (True, False):
        eng.set_layer_groups(mode)
        by_mode[mode] = eng.iaf_generate(mel, noise, want=('x',))['x'].clone()
    eng.set_layer_groups(None)
    stop = threading.Event()
    stats = {'calls': 0, 'bad': 0, 'worker_err': None}

    def worker():
        f = eng.fork()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            m, n = torch.from_numpy(mel).cuda(), torch.from_numpy(noise).cuda()
            while not stop.is_set():
                x = f.iaf_generate(m, n, want=('x',), check_range=False)['x']
                st.synchronize()
                stats['calls'] += 1
                if not (torch.equal(x, by_mode[True]) or torch.equal(x, by_mode[False])):
                    stats['bad'] += 1
            try:                                               # a failing call of THIS thread
                f.iaf_generate(np.zeros([1, F, 79], np.float32))
            except ValueError as e:
                stats['worker_err'] = str(e)
        f.close()
    th = threading.Thread(target=worker)
    th.start()
    lib = _lib.load()
    refused = accepted = 0
    import time
    t_end = time.time() + 3.0
    while time.time() < t_end:
        for fn, arg in ((lib.wn_iaf_set_groups, 1), (lib.wn_iaf_set_groups, -1), (lib.wn_iaf_set_groups, 0)):
            rc = fn(eng._h, arg)
            assert rc in (0, -1), rc                            # WN_OK or WN_ESTATE, nothing else
            if rc == -1:
                refused += 1
                assert b'in flight' in lib.wn_last_error(eng._h)
            else:
                accepted += 1
    stop.set()
    th.join()
    assert stats['calls'] > 20 and stats['bad'] == 0, stats
    assert accepted > 0                                         # (refusals depend on timing; both outcomes are legal)
    assert stats['worker_err'] and 'mel must be' in stats['worker_err']
    # this thread's last error is its own: provoke one and read it back
    assert lib.wn_iaf_set_groups(eng._h, 5) == -22 and b'wn_iaf_set_groups' in lib.wn_last_error(eng._h)
    print('switch attempts while calls were in flight: {} accepted, {} refused (WN_ESTATE); {} generate calls, all equal to one of '
          'the two forms'.format(accepted, refused, stats['calls']))
    eng.set_layer_groups(None)
    eng.close()


@pytest.mark.parametrize('use_graph', [False, True])
def test_one_teacher_handle_two_threads_two_streams(use_graph):
    """The autoregressive loop from two threads on ONE teacher handle (own queue state and stream each; hipGraph replay: every
    call owns its graphs): both index streams equal the serial runs bit for bit, which the golden test holds to the oracle."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, 'ar_mol.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    eng = Engine(cfgd).load_weights(O.synth_weights(hp, 'teacher', seed=1234, init='unit'))
    enc, rnd = g['enc'], g['rnd']
    B, Tn = enc.shape[0], enc.shape[1]
    cases = [(enc, rnd), (enc[::-1].copy(), rnd[:, ::-1].copy())]
    eng._set_ar_graph(use_graph)
    serial = []
    for e, r in cases:
        o = eng.ar_generate(e, r, want_out=True, use_graph=use_graph)
        serial.append({k: v.clone() for k, v in o.items()})
    torch.cuda.synchronize()
    assert np.abs(_np(serial[0]['idx']).astype(np.int64) - g['free_idx']).max() <= 1   # the committed golden stream (+-1 LSB: libm)

    def worker(i):
        def run():
            f = eng.fork()
            st = torch.cuda.Stream()
            got = []
            with torch.cuda.stream(st):
                for _ in range(4):
                    got.append(f.ar_generate(cases[i][0], cases[i][1], want_out=True, use_graph=use_graph))
            st.synchronize()             # (its own stream only: ROCm refuses a device-wide synchronise while ANY thread is
            f.close()                    #  capturing, thread-local capture mode or not, and invalidates that capture)
            return got
        return run
    outs = _run_threads([worker(0), worker(1)])
    torch.cuda.synchronize()
    for i in range(2):
        for o in outs[i]:
            for k in ('idx', 'wav', 'out_params'):
                assert torch.equal(o[k], serial[i][k]), (i, k, use_graph)
    eng.close()


def test_a_capture_lost_to_another_threads_device_synchronise_falls_back_to_plain_launches():
    """ROCm 7.2 invalidates a stream capture when ANY thread synchronises the device and leaves the captured stream unusable
    for good.  wn_ar_generate captures on a private stream of the handle, so what is lost is that stream: the call falls back
    to plain launches on the caller's stream and every result still equals the serial call (scripts/dev_ar_capture_hammer.py:
    180 calls, 357 lost captures, 0 wrong results)."""
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, 'ar_mol.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    eng = Engine(cfgd).load_weights(O.synth_weights(hp, 'teacher', seed=1234, init='unit'))
    enc, rnd = g['enc'], g['rnd']
    eng._set_ar_graph(True)
    serial = {k: v.clone() for k, v in eng.ar_generate(enc, rnd, want_out=True, use_graph=True).items()}
    torch.cuda.synchronize()
    stop = threading.Event()

    def hammer():
        while not stop.is_set():
            try:
                torch.cuda.synchronize()
            except RuntimeError:                 # refused by the runtime while the other thread captures: the hammer's problem
                pass
    th = threading.Thread(target=hammer)
    th.start()
    try:
        f = eng.fork()
        st = torch.cuda.Stream()
        got = []
        with torch.cuda.stream(st):
            for _ in range(8):
                got.append(f.ar_generate(enc, rnd, want_out=True, use_graph=True))
        st.synchronize()
    finally:
        stop.set()
        th.join()
    f.close()
    for o in got:
        for k in ('idx', 'wav', 'out_params'):
            assert torch.equal(o[k], serial[k]), k
    eng.close()


def test_ar_groups_on_streams_equal_the_single_stream_call(monkeypatch):
    """Engine.ar_generate(streams=G): G utterance groups as independent chains on G streams against one handle, injected
    randoms -- row for row the single-stream call (the batched MFMA step forced for every group size: the GEMV step small
    groups would otherwise take sums in another order)."""
    monkeypatch.setenv('WN_AR_MODE', 'mfma')
    import torch
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    g = np.load(os.path.join(GOLD, 'ar_mol.npz'))
    cfgd = json.loads(str(g['cfg_json']))
    hp = O.HP(cfgd)
    eng = Engine(cfgd).load_weights(O.synth_weights(hp, 'teacher', seed=1234, init='unit'))
    enc = np.concatenate([g['enc'], g['enc'][::-1]], 0)         # 2 B utterances
    rnd = np.concatenate([g['rnd'], g['rnd'][:, ::-1]], 1)
    B = enc.shape[0]
    one = eng.ar_generate(enc, rnd, want_out=True)
    for G in (2, B):
        many = eng.ar_generate(enc, rnd, want_out=True, streams=G)
        for k in ('idx', 'wav', 'out_params'):
            assert torch.equal(many[k], one[k]), (G, k)
    with pytest.raises(ValueError):
        eng.ar_generate(enc, rnd, streams=B + 1)
    eng.close()
