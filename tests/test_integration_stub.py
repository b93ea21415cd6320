"""The ctypes stub INTEGRATION.md tells a maintainer of the reference to paste (`wavenet/parallelgen_hip.py`) is
EXECUTED here as written in the document -- only the library path is pointed at the in-tree build -- and held to the
float64 oracle: a normal call, and a call that leaves the fp16 range of the default arithmetic, which the stub must
notice (wn_iaf_range_status) and re-run on the fp32-MFMA form instead of returning NaN with return code 0."""
import argparse
import os
import re

import numpy as np
import pytest

from conftest import load_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    src = next(b for b in blocks if 'def synthesis_hip' in b)
    assert "C.CDLL('libwnhip.so')" in src
    return src


def test_stub_text_is_complete_python_and_asks_the_range_guard():
    src = _stub_source()
    compile(src, 'INTEGRATION.md:parallelgen_hip', 'exec')            # syntactically whole
    body = src[src.index('def synthesis_hip'):]
    assert 'wn_iaf_range_status' in body and 'WN_FORM_F32' in body and body.index('wn_iaf_range_status') < body.index('.cpu().numpy()')


@pytest.mark.gpu
def test_stub_runs_as_written_and_falls_back_outside_the_fp16_range():
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd import _lib
    from test_gpu_iaf import _overflow_case
    ns = {}
    exec(compile(_stub_source().replace("C.CDLL('libwnhip.so')", 'C.CDLL({!r})'.format(_lib.LIB_PATH)),
                 'INTEGRATION.md:parallelgen_hip', 'exec'), ns)
    # (1) the shipped student, device-drawn noise: finite audio on the 2^-15 grid of the right length
    cfgd = load_json('parallel_wavenet.json')
    hp = O.HP(cfgd)
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    hparams = argparse.Namespace(**cfgd)
    h = ns['build'](hparams, w)
    mel = np.random.RandomState(5).uniform(0, 1, [2, 24, 80]).astype(np.float32)
    wav = ns['synthesis_hip'](h, mel, seed=3)
    T = O.iaf_length(24, hp)
    assert wav.shape == (2, T) and np.isfinite(wav).all() and np.all(wav * 32768 == np.round(wav * 32768))
    assert np.array_equal(wav, ns['synthesis_hip'](h, mel, seed=3)) and not np.array_equal(wav, ns['synthesis_hip'](h, mel, seed=4))
    ns['lib'].wn_destroy(h)
    # (2) scale = e^7 in two flows: the split-fp16 call overflows; the stub must come back with the fp32 form's audio
    cfgd2, hp2, w2, mel2, noise2 = _overflow_case(1200.0)
    h2 = ns['build'](argparse.Namespace(**cfgd2), w2)
    got = ns['synthesis_hip'](h2, mel2, seed=11)
    assert np.isfinite(got).all()                       # never NaN with rc 0
    # same seed on the engine's fp32 form = the same Philox draws = the same audio
    from nsynth_wavenet_amd.engine import Engine
    eng = Engine(cfgd2, precision='f32').load_weights(w2)
    ref = eng.iaf_generate(mel2, None, seed=11, want=('wav',))['wav'].cpu().numpy()
    eng.close()
    assert np.array_equal(got, ref)
    ns['lib'].wn_destroy(h2)
