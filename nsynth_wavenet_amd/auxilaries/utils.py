"""Host-side audio helpers of the generation path.

numpy codecs used by the per-sample feedback of the reference's fastgen driver
(auxilaries/utils.py:89-105,125-139,162-169 of the reference) and its audio loader
(:55-69).  The device versions live in csrc/wn_codec.h.
"""
import os

import numpy as np
from scipy.io import wavfile


def shell_path(path):
    return os.path.abspath(os.path.expanduser(os.path.expandvars(path)))


def load_audio(path, sample_length=64000, sr=16000):
    """Mono float32 in [-1,1) at `sr` Hz, truncated to sample_length when > 0.
    (The reference uses librosa.load; librosa is not available here, so PCM wavs
    are read with scipy and resampled polyphase when their rate differs.)"""
    rate, data = wavfile.read(path)
    if data.dtype == np.int16:
        audio = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        audio = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        audio = (data.astype(np.float32) - 128.0) / 128.0
    else:
        audio = data.astype(np.float32)
    if audio.ndim > 1:
        audio = audio.mean(axis=1)
    if rate != sr:
        from scipy.signal import resample_poly
        from math import gcd
        g = gcd(int(rate), int(sr))
        audio = resample_poly(audio, sr // g, rate // g).astype(np.float32)
    if sample_length > 0:
        audio = audio[:sample_length]
    return audio


def mu_law_numpy(x, mu=255, int8=False):
    out = np.sign(x) * np.log(1 + mu * np.abs(x)) / np.log(1 + mu)
    out = np.floor(out * 128)
    return out.astype(np.int8) if int8 else out


def inv_mu_law_numpy(x, mu=255.0):
    x = np.array(x).astype(np.float32)
    out = (x + 0.5) * 2. / (mu + 1)
    out = np.sign(out) / mu * ((1 + mu) ** np.abs(out) - 1)
    return np.where(np.equal(x, 0), x, out)


def cast_quantize_numpy(x, quant_chann):
    """auxilaries/utils.py:162-164 of the reference: scale, then astype(int32) -- numpy TRUNCATES toward zero here
    (the TF twin cast_quantize floors, utils.py:153-154; the two differ for negative non-grid inputs, and the
    reference only ever feeds this helper values that already lie on the grid)."""
    return (np.asarray(x) * quant_chann / 2).astype(np.int32)


def inv_cast_quantize_numpy(x_quantized, quant_chann):
    return np.asarray(x_quantized).astype(np.float32) / (quant_chann / 2)
