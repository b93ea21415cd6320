"""CPU oracle of the log-mel featuriser (TEST INFRASTRUCTURE, see oracle/__init__.py) and the closed-form
expectations the featurisers are checked against.

Restates auxilaries/mel_extractor.py:14-35 (parameters), :31-35 (melspectrogram), :65-90 (_stft ->
librosa.stft(n_fft=2048, hop=200, win=800), _linear_to_mel -> librosa.filters.mel(16000, 2048, 80, 125, 7600),
_amp_to_db, _normalize) of the reference.  librosa is not installed here: PARITY UNPINNED against librosa
itself; the published definitions are restated -- centred frames over a reflect-padded signal, a PERIODIC
Hann window of `win_length` samples zero-padded symmetrically to n_fft, magnitude, Slaney-scale triangular
filters with area normalisation -- and written differently from the product module on purpose (explicit DFT
matrix, per-filter loops), so that the two share no code.

`analytic_*` give what a featuriser must produce for signals whose spectrum is known in closed form:
a stationary tone (the window's DTFT at the bin offsets, no FFT involved) and a unit impulse (the window
sample at the impulse position, flat over frequency).
"""
import numpy as np

SR = 16000
N_FFT = 2048
HOP = 200
WIN = 800
N_MEL = 80
FMIN, FMAX = 125.0, 7600.0
MIN_LEVEL_DB = -140.0
MIN_AMP = 1e-5


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True): 0.5 - 0.5 cos(2 pi k / n)."""
    return np.array([0.5 - 0.5 * np.cos(2.0 * np.pi * k / n) for k in range(n)], np.float64)


def padded_window():
    w = np.zeros(N_FFT, np.float64)
    lp = (N_FFT - WIN) // 2
    w[lp:lp + WIN] = hann_periodic(WIN)
    return w


def slaney_hz_to_mel(f):
    f = float(f)
    if f < 1000.0:
        return 3.0 * f / 200.0
    return 15.0 + 27.0 * np.log(f / 1000.0) / np.log(6.4)


def slaney_mel_to_hz(m):
    m = float(m)
    if m < 15.0:
        return 200.0 * m / 3.0
    return 1000.0 * np.exp(np.log(6.4) * (m - 15.0) / 27.0)


def mel_basis():
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with htk=False, norm='slaney' (librosa < 0.8 default):
    80 triangles with corners equally spaced on the Slaney mel scale, each scaled by 2 / (f_hi - f_lo)."""
    lo, hi = slaney_hz_to_mel(FMIN), slaney_hz_to_mel(FMAX)
    corners = [slaney_mel_to_hz(lo + (hi - lo) * i / (N_MEL + 1)) for i in range(N_MEL + 2)]
    basis = np.zeros((N_MEL, N_FFT // 2 + 1), np.float64)
    for m in range(N_MEL):
        f0, f1, f2 = corners[m], corners[m + 1], corners[m + 2]
        for k in range(N_FFT // 2 + 1):
            f = k * SR / float(N_FFT)
            up = (f - f0) / (f1 - f0)
            down = (f2 - f) / (f2 - f1)
            basis[m, k] = max(0.0, min(up, down)) * 2.0 / (f2 - f0)
    return basis


_DFT = None


def stft_mag(y):
    """|librosa.stft(y, 2048, 200, 800)| with center=True, reflect padding: [1025, 1 + len(y)//200], float64,
    by an explicit DFT matrix (no FFT routine shared with the product)."""
    global _DFT
    y = np.asarray(y, np.float64)
    yp = np.pad(y, N_FFT // 2, mode='reflect')
    n_frames = 1 + (len(yp) - N_FFT) // HOP
    w = padded_window()
    if _DFT is None:
        k = np.arange(N_FFT // 2 + 1)[:, None]
        n = np.arange(N_FFT)[None, :]
        ang = -2.0 * np.pi * ((k * n) % N_FFT) / N_FFT
        _DFT = (np.cos(ang), np.sin(ang))
    out = np.zeros((N_FFT // 2 + 1, n_frames), np.float64)
    for t in range(n_frames):
        seg = yp[t * HOP:t * HOP + N_FFT] * w
        re, im = _DFT[0] @ seg, _DFT[1] @ seg
        out[:, t] = np.sqrt(re * re + im * im)
    return out


def db_normalise(S):
    db = 20.0 * np.log10(np.maximum(MIN_AMP, S))
    return np.clip((db - MIN_LEVEL_DB) / -MIN_LEVEL_DB, 0.0, 1.0)


def melspectrogram(y):
    """[frames, 80] float64."""
    return db_normalise(mel_basis() @ stft_mag(y)).T


# ------------------------------------------------------------------ closed forms ----
def hann_dtft_mag(omega):
    """|sum_{n<WIN} hann_periodic[n] e^{-j omega n}| from the three Dirichlet kernels of the Hann window
    (w[n] = 1/2 - 1/4 e^{j 2 pi n/N} - 1/4 e^{-j 2 pi n/N}); omega in radians per sample."""
    N = WIN

    def dirichlet(th):           # sum_{n<N} e^{-j th n}
        th = np.asarray(th, np.float64)
        small = np.abs(np.sin(th / 2.0)) < 1e-12
        num = np.sin(N * th / 2.0)
        den = np.where(small, 1.0, np.sin(th / 2.0))
        mag = np.where(small, float(N), num / den)
        return mag * np.exp(-1j * th * (N - 1) / 2.0)

    om = np.asarray(omega, np.float64)
    d = 2.0 * np.pi / N
    return np.abs(0.5 * dirichlet(om) - 0.25 * dirichlet(om - d) - 0.25 * dirichlet(om + d))


def analytic_tone_mag(freq_hz, amp):
    """Magnitude spectrum [1025] of an interior STFT frame of amp*sin(2 pi f t + phi): positive-frequency line only
    (the image at -f adds < 1e-4 relative for 200 Hz < f < 7800 Hz with this window)."""
    k = np.arange(N_FFT // 2 + 1)
    om = 2.0 * np.pi * (k / float(N_FFT) - freq_hz / float(SR))
    return 0.5 * amp * hann_dtft_mag(om)


def analytic_tone_mel(freq_hz, amp):
    return db_normalise(mel_basis() @ analytic_tone_mag(freq_hz, amp))


def analytic_impulse_mel(pos, n_samples, amp=1.0):
    """[frames, 80] for amp * delta[n - pos] (pos far from both ends): frame t sees the window sample at
    pos + 1024 - 200 t, the same magnitude in every bin, so every mel band gets window * amp * (row sum)."""
    w = padded_window()
    rows = mel_basis().sum(axis=1)
    n_frames = 1 + n_samples // HOP
    out = np.zeros((n_frames, N_MEL), np.float64)
    for t in range(n_frames):
        i = pos + N_FFT // 2 - t * HOP
        mag = amp * w[i] if 0 <= i < N_FFT else 0.0
        out[t] = db_normalise(rows * mag)
    return out
