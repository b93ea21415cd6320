"""Log-mel featuriser: the host featuriser of the product package and the oracle restatement
(oracle/mel_np.py, float64, explicit DFT, no shared code) are EACH held against closed-form analysis -- two
stationary tones and a unit impulse -- and against each other on noise.  The GPU twin of this test is
tests/test_gpu_iaf.py::test_device_mel_featuriser_against_analysis."""
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
