"""Autoregressive generation driver -- drop-in for the reference's wavenet/fastgen.py:
`load_batch`, `save_batch`, `encode`, `synthesis` (fastgen.py:17-58,69-88,128-169).

The reference's per-sample python loop (one sess.run per audio sample, numpy
de-quantisation on the host) is replaced by wn_ar_generate, which keeps the whole
loop -- queues, sampling head, feedback -- on the device.
"""
import logging
import os

import numpy as np
import torch
from scipy.io import wavfile

from .. import config as cfg
from ..auxilaries import mel_extractor, utils


def get_ema_shadow_dict(var_names):
    """checkpoint key -> variable name (fastgen.py:12-14)."""
    return {'{}/ExponentialMovingAverage'.format(n): n for n in var_names}


def load_batch(files, sample_length=64000):
    """A list of .wav (audio) or .npy (arrays) files -> zero-padded batch array."""
    batch_data = []
    max_length = 0
    is_npy = (os.path.splitext(files[0])[1] == '.npy')
    for f in files:
        data = np.load(f) if is_npy else utils.load_audio(f, sample_length, sr=16000)
        batch_data.append(data)
        max_length = max(max_length, data.shape[0])
    for i, data in enumerate(batch_data):
        if data.shape[0] < max_length:
            padded = np.zeros((max_length,) + data.shape[1:], dtype=data.dtype)
            padded[:data.shape[0]] = data
            batch_data[i] = padded
    return np.stack(batch_data, axis=0)


def save_batch(batch_audio, batch_save_paths):
    for audio, name in zip(batch_audio, batch_save_paths):
        logging.info('Saving: %s' % name)
        wavfile.write(name, 16000, np.asarray(audio, np.float32))


def _teacher_engine(hparams, checkpoint_path):
    from . import parallelgen
    return parallelgen._engine_for(hparams, checkpoint_path, 'teacher')


def encode_mel(hparams, mel, checkpoint_path):
    """mel [B,F,80] -> conditioning [B, F*200, deconv_width] (Wavenet.deconv_stack)."""
    eng = _teacher_engine(hparams, checkpoint_path)
    return eng.deconv(mel).cpu().numpy()


def encode(hparams, wav_data, checkpoint_path):
    """Audio [B,L] (or [L]) -> mel -> deconv-stack encoding, as fastgen.py:69-88."""
    wav_data = np.asarray(wav_data)
    if wav_data.ndim == 1:
        wav_data = np.expand_dims(wav_data, 0)
    import torch
    mel_val = mel_extractor.batch_melspectrogram_device(wav_data) if torch.cuda.is_available() \
        else mel_extractor.batch_melspectrogram(wav_data)
    return encode_mel(hparams, mel_val, checkpoint_path)


def calculate_cond_vars(hparams, encoding, checkpoint_path):
    """fastgen.py:91-115: the per-layer conditioning projections of `encoding` [B, T, deconv_width] as numpy arrays."""
    eng = _teacher_engine(hparams, checkpoint_path)
    return {k: v.cpu().numpy() for k, v in eng.ar_cond_vars(np.asarray(encoding, np.float32)).items()}


def generate(hparams, mel_encoding, checkpoint_path, rnd=None, seed=None):
    eng = _teacher_engine(hparams, checkpoint_path)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    out = eng.ar_generate(mel_encoding, rnd=rnd, seed=seed)
    return out['wav'].cpu().numpy()


def synthesis(hparams, mel_encoding, save_paths, checkpoint_path):
    audio_batch = generate(hparams, mel_encoding, checkpoint_path)
    save_batch(audio_batch, save_paths)
