"""Log-mel featuriser: the host featuriser of the product package and the oracle restatement
(oracle/mel_np.py, float64, explicit DFT, no shared code) are EACH held against closed-form analysis -- two
stationary tones and a unit impulse -- and against each other on noise.  The GPU twin of this test is
tests/test_gpu_iaf.py::test_device_mel_featuriser_against_analysis."""
import os

import numpy as np

from nsynth_wavenet_amd.auxilaries import mel_extractor as M
from oracle import mel_np as OM


def _signals(n=8000):
    t = np.arange(n) / 16000.0
    return {
        'tone_a': ((0.5 * np.sin(2 * np.pi * 1000.0 * t + 0.3)).astype(np.float32), (1000.0, 0.5)),
        'tone_b': ((0.25 * np.sin(2 * np.pi * 3437.5 * t + 1.1)).astype(np.float32), (3437.5, 0.25)),   # off-bin: 440.0 bins
    }


def test_closed_forms_hold_for_the_oracle_and_the_host_featuriser():
    # the window's closed-form DTFT against a direct sum
    om = np.linspace(-0.2, 0.2, 41)
    w = OM.hann_periodic(OM.WIN)
    direct = np.abs((w[None, :] * np.exp(-1j * om[:, None] * np.arange(OM.WIN)[None, :])).sum(axis=1))
    assert np.abs(OM.hann_dtft_mag(om) - direct).max() < 1e-9 and abs(OM.hann_dtft_mag(0.0) - 400.0) < 1e-9
    # filterbank: Slaney area normalisation, 80 non-empty triangles inside [125, 7600] Hz, product == oracle
    fb = OM.mel_basis()
    freqs = np.arange(1025) * 16000.0 / 2048
    assert fb.shape == (80, 1025) and np.all(fb >= 0) and np.all(fb.sum(axis=1) > 0)
    assert np.all(fb[:, freqs < 125.0] == 0) and np.all(fb[:, freqs > 7600.0] == 0)
    area = (fb * (16000.0 / 2048)).sum(axis=1)                     # integral of each triangle over Hz ~ 1
    assert np.abs(area[10:] - 1.0).max() < 0.02
    assert np.abs(M.mel_filterbank().astype(np.float64) - fb).max() < 1e-7
    for name, (y, (f, a)) in _signals().items():
        want = OM.analytic_tone_mel(f, a)
        for label, got in (('oracle', OM.melspectrogram(y)), ('host', M.melspectrogram(y).astype(np.float64))):
            mid = got[15:26]                                       # interior frames: no reflect-padding effects
            # bands within 60 dB of the strongest one: there the line at -f (whose phase differs from frame to
            # frame and which the closed form leaves out) is below 1e-3 of the band's magnitude
            near = want >= want.max() - 60.0 / 140.0
            assert near.sum() >= 3
            assert np.abs(mid[:, near] - want[None, near]).max() < 2e-4, (name, label)
            assert np.abs(mid - want[None, :]).max() < 0.05, (name, label)      # far skirts: leakage at -90 dB
        # the band that holds the tone carries its closed-form level: 20 log10(a/2 * 400 * basis) dB
        k0 = int(round(f * 2048 / 16000.0))
        if abs(f * 2048 / 16000.0 - k0) < 1e-9:
            assert abs(OM.stft_mag(y)[k0, 20] - 0.5 * a * 400.0) < 1e-3
    # impulse: flat spectrum = window sample; every band follows the window through the frames
    n = 8000
    y = np.zeros(n, np.float32)
    y[4000] = 1.0
    want = OM.analytic_impulse_mel(4000, n)
    assert np.abs(OM.melspectrogram(y) - want).max() < 1e-9
    assert np.abs(M.melspectrogram(y).astype(np.float64) - want).max() < 2e-5
    assert want[20].max() > 40.0 / 140.0 + 0.2 and np.allclose(want[5], 40.0 / 140.0)   # inside / outside the window
    # silence sits on the floor: max(1e-5, 0) -> -100 dB -> (140 - 100) / 140
    assert np.allclose(OM.melspectrogram(np.zeros(2000)), 40.0 / 140.0)


def test_host_featuriser_matches_the_oracle_on_noise_and_the_fixture_length():
    rs = np.random.RandomState(0)
    y = rs.uniform(-0.5, 0.5, 6000).astype(np.float32)
    a, b = M.melspectrogram(y).astype(np.float64), OM.melspectrogram(y)
    assert a.shape == b.shape == (31, 80)
    assert np.abs(a - b).max() < 5e-5
    assert M.melspectrogram(np.zeros(154480, np.float32)).shape == (773, 80)      # the reference fixture: 1 + 154480 // 200


def test_fixture_mel_of_the_reference_test_utterance_is_well_formed():
    """tests/golden/fixture_mel.npz (made by make_fixture_mel.py from the reference's tests/test_data/test.wav): F = 1 + n // 200
    = 773 frames for the 154 480 samples the reference's fixture has, values on the featuriser's [40/140, 1] range
    (auxilaries/mel_extractor.py:85-90: 20 log10(1e-5) = -100 dB is the floor), and the stored head of the utterance re-analysed
    by the host featuriser AND by the float64 oracle reproduces the fixture's first frames (those whose window ends inside the head)."""
    here = os.path.dirname(os.path.abspath(__file__))
    fx = np.load(os.path.join(here, 'golden', 'fixture_mel.npz'))
    facts = np.load(os.path.join(here, 'golden', 'ref_fixture_facts.npz'))
    mel, n = fx['mel'], int(fx['n_samples'])
    assert n == int(facts['test_wav/n']) == 154480 and mel.shape == (1 + n // 200, 80) == (773, 80) and mel.dtype == np.float32
    assert mel.min() >= 40.0 / 140.0 - 1e-6 and mel.max() <= 1.0 and mel.std() > 0.05
    head = fx['wav_head']
    assert head.shape == (2048,) and np.abs(head).max() <= 1.0
    # frame f covers samples [200 f - 400, 200 f + 400) (a window of 800 centred in the 2 048-point frame): frames 0..8 lie
    # inside the first 2 048 samples (frame 0 reaches into the reflect padding, which mirrors samples 1..400 of the head)
    for fn in (M.melspectrogram, lambda y: np.asarray(OM.melspectrogram(y), np.float32)):
        got = fn(head)[:9]
        assert np.abs(got - mel[:9]).max() <= 2e-5, float(np.abs(got - mel[:9]).max())
