"""Log-mel featuriser that feeds the generation path.

Restates the numpy/librosa branch of the reference's auxilaries/mel_extractor.py
(:14-35 parameters, :31-44 melspectrogram, :65-90 stft / mel basis / dB /
normalise) without librosa: STFT n_fft 2048, hop 200, periodic Hann window of 800
samples centred in the FFT frame, reflect-padded centred frames; Slaney-scale,
area-normalised triangular mel filterbank (80 bands, 125-7600 Hz) applied to the
MAGNITUDE; 20*log10(max(1e-5, .)); clip((S + 140) / 140, 0, 1); time-major output.
`preemphasis` and `ref_level_db` are declared by the reference but never applied.
`melspectrogram` / `batch_melspectrogram` are host numpy (the reference's own host-side form);
`batch_melspectrogram_device` is the same featuriser as a HIP kernel behind the C ABI (`wn_mel_spectrogram`,
csrc/wn_mel.hip), so that wav -> mel -> audio stays on the device in the CLIs.
"""
import numpy as np

SAMPLE_RATE = 16000
NUM_FREQ = 1025
NUM_MEL = 80
N_FFT = (NUM_FREQ - 1) * 2
FRAME_SHIFT = int(12.5 * SAMPLE_RATE / 1000.)
WIN_LENGTH = int(50 * SAMPLE_RATE / 1000.)
MIN_LEVEL_DB = -140
MEL_FMIN, MEL_FMAX = 125, 7600
MIN_AMP = 1e-5

_mel_basis = None


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_mel = 1000.0 / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, 1000.0 * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=NUM_MEL, fmin=MEL_FMIN, fmax=MEL_FMAX):
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


def stft(y, n_fft=N_FFT, hop=FRAME_SHIFT, win_length=WIN_LENGTH):
    y = np.asarray(y, np.float32)
    n = np.arange(win_length)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * n / win_length)          # periodic Hann
    lpad = (n_fft - win_length) // 2
    window = np.zeros(n_fft)
    window[lpad:lpad + win_length] = win
    yp = np.pad(y, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = yp[idx] * window[None, :]
    return np.fft.rfft(frames, axis=1).T.astype(np.complex64)      # [1025, frames]


def melspectrogram(y):
    global _mel_basis
    if _mel_basis is None:
        _mel_basis = mel_filterbank()
    S = np.dot(_mel_basis, np.abs(stft(y)))
    S = 20 * np.log10(np.maximum(MIN_AMP, S))
    NS = np.clip((S - MIN_LEVEL_DB) / -MIN_LEVEL_DB, 0, 1)
    return NS.T.astype(np.float32)


def batch_melspectrogram(y):
    assert len(y.shape) == 2
    return np.array([melspectrogram(y[b]) for b in range(y.shape[0])])


def batch_melspectrogram_device(y, device=None):
    """[B, L] float audio (numpy or torch) -> [B, 1 + L // 200, 80] float32 mel ON THE GPU, through the C ABI
    (`wn_mel_spectrogram`, csrc/wn_mel.hip: a hand-written HIP kernel, no FFT library).  Same definition as
    `melspectrogram` (centred reflect-padded frames, periodic Hann 800 in a 2048-point DFT, magnitude, Slaney mel,
    dB, normalise); differences are float32 rounding.  There is no CPU fallback: without libwnhip.so this raises."""
    import torch
    from .. import _lib
    dev = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
    y = torch.as_tensor(y, dtype=torch.float32, device=dev).contiguous()
    if y.dim() != 2:
        raise ValueError('batch_melspectrogram_device: expected [B, L] audio, got shape {}'.format(tuple(y.shape)))
    lib = _lib.load()
    B, L = int(y.shape[0]), int(y.shape[1])
    with torch.cuda.device(dev):
        out = torch.empty(B, int(lib.wn_mel_frames(L)), NUM_MEL, dtype=torch.float32, device=dev)
        _lib.check(lib.wn_mel_spectrogram(y.data_ptr(), B, L, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return out
