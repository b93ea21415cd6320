"""Codecs against the REFERENCE'S OWN numpy functions (tests/golden/ref_codec.npz, produced by
tests/golden/make_ref_codec.py, which executes /root/reference/auxilaries/utils.py:90-105,125-139,162-169 in the build
container).  This is the one place where expected values come from reference code that ran, not from a restatement:
it pins SURVEY.md section 8 rows a8 (quantiser) and a11 (mu-law) on the numpy side.  The float IAF / AR path remains
"parity unpinned" (oracle/__init__.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
REF = np.load(os.path.join(HERE, 'golden', 'ref_codec.npz'))


def _inputs():
    import make_ref_codec as M            # input generator only (seeded RandomState + edge values): no reference access
    assert M.SEED == int(REF['seed']) and M.N_RANDOM == int(REF['n_random'])
    x = M.inputs()
    assert x.size == int(REF['n_random']) + int(REF['n_edge'])
    return x


GRID16 = np.arange(-32768, 32768, dtype=np.float32) / np.float32(32768.0)


def _floor_from_trunc(x, trunc, Q):
    """The generation path floors (utils.py:153-154 through parallel_wavenet.py:349-357), the reference's numpy twin
    truncates (:162-164): identical for x >= 0 and where x * Q/2 is an integer, one step lower otherwise."""
    y = x.astype(np.float64) * (Q / 2)
    return trunc.astype(np.int64) - ((y < 0) & (y != np.round(y)))


def test_reference_round_trips_held_when_the_vectors_were_made():
    assert bool(REF['roundtrip_mu_law_ok']) and bool(REF['roundtrip_cast_ok'])        # SURVEY K7, by the reference itself


@pytest.mark.parametrize('impl', ['host', 'oracle'])
def test_numpy_codecs_equal_the_reference_outputs(impl):
    x = _inputs()
    if impl == 'host':
        from nsynth_wavenet_amd.auxilaries import utils as U
        mu, imu = U.mu_law_numpy, U.inv_mu_law_numpy
        assert np.array_equal(U.cast_quantize_numpy(x, 65536), REF['cast_trunc_x_65536'])     # the truncating twin, as is
        assert np.array_equal(U.cast_quantize_numpy(x, 256), REF['cast_trunc_x_256'])
        assert np.array_equal(U.cast_quantize_numpy(GRID16, 65536), REF['cast_trunc_grid16'])
        assert np.array_equal(U.inv_cast_quantize_numpy(np.arange(-32768, 32768, dtype=np.int32), 65536),
                              REF['inv_cast_codes_65536'])
        assert np.array_equal(U.inv_cast_quantize_numpy(np.arange(-128, 128, dtype=np.int32), 256), REF['inv_cast_codes_256'])
    else:
        from oracle import wavenet_np as O
        mu, imu = O.mu_law, O.inv_mu_law
        for Q, key in ((65536, 'cast_trunc_x_65536'), (256, 'cast_trunc_x_256')):
            assert np.array_equal(O.cast_quantize(x, Q).astype(np.int64), _floor_from_trunc(x, REF[key], Q))
        assert np.array_equal(O.cast_quantize(GRID16, 65536), REF['cast_trunc_grid16'])
        assert np.array_equal(O.inv_cast_quantize(np.arange(-32768, 32768), 65536).astype(np.float32),
                              REF['inv_cast_codes_65536'])
    # integer mu-law codes, bit-exact.  Two reference vectors (make_ref_codec.py): the function as written runs in
    # float64 under this image's NumPy 2.2 (a float64 scalar np.log(1 + mu) meets a float32 array), in float32 under the
    # NumPy 1.x the reference was written for (= the call with mu=np.float32(255)).  The host twin is the reference's
    # expression verbatim and follows NumPy like it; the oracle restates the float32 arithmetic of the reference's era,
    # which is also what the device computes in.  The two vectors differ at x = -1.0 only (-129 against -128).
    sfx = '' if impl == 'host' else '_f32'
    assert np.array_equal(np.asarray(mu(x)).astype(np.int16), REF['mu_law_x' + sfx])
    assert np.array_equal(np.asarray(mu(GRID16)).astype(np.int16), REF['mu_law_grid16' + sfx])
    d = np.nonzero(REF['mu_law_x'] != REF['mu_law_x_f32'])[0]
    assert d.size == 1 and x[d[0]] == -1.0 and REF['mu_law_x'][d[0]] == -129 and REF['mu_law_x_f32'][d[0]] == -128
    # decode table: SURVEY K6 -- float table is pinned to one float32 ulp across numpy builds, exactly 0 at code 0
    tab = np.asarray(imu(np.arange(-128, 128))).astype(np.float32)
    assert np.abs(tab - REF['inv_mu_law_codes']).max() <= 2.0 ** -23 and tab[128] == 0.0 and REF['inv_mu_law_codes'][128] == 0.0


@pytest.mark.gpu
def test_device_quantiser_and_mulaw_decode_equal_the_reference_outputs():
    """wn_clip_quant (parallel_wavenet.py:347-359 on the device) against the reference-derived vectors: the int32 index
    bit-exact for both channel counts, the 16-bit audio exact, the mu-law audio within one float32 ulp of the reference's
    table with exact zero at code 0."""
    from conftest import load_json
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.engine import Engine
    x = _inputs()
    w = O.synth_weights(O.HP(load_json('parallel_wavenet.json')), 'student')
    for mu_law, Q, key in ((False, 65536, 'cast_trunc_x_65536'), (True, 256, 'cast_trunc_x_256')):
        eng = Engine(dict(load_json('parallel_wavenet.json'), use_mu_law=mu_law)).load_weights(w)
        wav, idx = eng.clip_quant(x)
        wav, idx = wav.cpu().numpy(), idx.cpu().numpy()
        xc = np.clip(x, np.float32(-1.0), np.float32(1.0 - 2.0 / Q))
        inside = xc == x                                   # the clip is the identity there: the reference vector applies directly
        want = _floor_from_trunc(x, REF[key], Q)
        assert inside.mean() > 0.99 and np.array_equal(idx[inside].astype(np.int64), want[inside])
        assert np.all(idx[x >= np.float32(1.0 - 2.0 / Q)] == Q // 2 - 1) and np.all(idx[x <= -1.0] == -Q // 2)
        if mu_law:
            tab = REF['inv_mu_law_codes']
            assert np.abs(wav - tab[idx + 128]).max() <= 2.0 ** -23 and np.all(wav[idx == 0] == 0.0)
        else:
            assert np.array_equal(wav, REF['inv_cast_codes_65536'][idx + 32768])
        eng.close()
