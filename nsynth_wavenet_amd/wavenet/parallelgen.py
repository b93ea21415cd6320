"""Parallel (IAF) generation driver -- drop-in for the reference's wavenet/parallelgen.py:
`load_parallelgen`, `synthesis(hparams, mel, save_paths, checkpoint_path)`.

The reference opens a TF graph + session, restores the EMA shadows and does a single
sess.run (parallelgen.py:22-51).  Here the "session" is a cached Engine and the
sess.run is one wn_iaf_generate call; the wall-clock 'Delay' log line is kept.
"""
import logging
import time

import numpy as np
import torch

from . import fastgen
from .. import config as cfg
from ..engine import Engine

_ENGINES = {}


def _engine_for(hparams, checkpoint_path, kind):
    hp = cfg.load_hparams(hparams)
    key = (kind, checkpoint_path, tuple(sorted((k, str(v)) for k, v in vars(hp).items())),
           torch.cuda.current_device() if torch.cuda.is_available() else -1)
    eng = _ENGINES.get(key)
    if eng is None:
        eng = Engine(hp, kind=kind)
        eng.load_checkpoint(checkpoint_path)
        _ENGINES[key] = eng
    return eng


def load_parallelgen(hparams, checkpoint_path):
    """Engine holding the restored student (replaces graph build + Saver.restore)."""
    return _engine_for(hparams, checkpoint_path, 'student')


def generate(hparams, mel, checkpoint_path, noise=None, seed=None):
    """mel [B,F,80] -> float32 numpy audio [B,T] (the fetched `x` of parallelgen.py:44-45)."""
    eng = load_parallelgen(hparams, checkpoint_path)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))       # the reference's draws are unseeded
    if torch.is_tensor(mel):                                # already featurised on the device
        mel_d = mel.to(device=eng.device, dtype=torch.float32).contiguous()
    else:
        mel_d = torch.as_tensor(np.ascontiguousarray(mel), dtype=torch.float32).to(eng.device)
    torch.cuda.synchronize(eng.device)
    start = time.time()
    out = eng.iaf_generate(mel_d, noise=noise, seed=seed, want=('wav',))
    audio = out['wav'].cpu().numpy()
    cost = time.time() - start
    wave_length = audio.shape[1] / 16000
    if wave_length > 0:
        logging.info('Target waveform length {:.5f}, Session run consume {:.5f} secs, Delay {:.2f}'.format(
            wave_length, cost, cost / wave_length))
    return audio


def synthesis(hparams, mel, save_paths, checkpoint_path):
    batch_size, length, num_mel = mel.shape
    assert len(save_paths) == batch_size
    audio = generate(hparams, mel, checkpoint_path)
    fastgen.save_batch(audio, save_paths)


def generate_async(hparams, mel, checkpoint_path, seed=None):
    """The GPU stage of a pipelined driver (cli.run): enqueue mel -> audio and the device-to-host copy of the result into a
    pinned buffer, return (pinned float32 tensor [B,T], event recorded behind the copy, enqueue time stamp) WITHOUT
    synchronising -- the caller's writer thread waits on the event while the next batch is already running."""
    eng = load_parallelgen(hparams, checkpoint_path)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    if torch.is_tensor(mel):
        mel_d = mel.to(device=eng.device, dtype=torch.float32).contiguous()
    else:
        mel_d = torch.as_tensor(np.ascontiguousarray(mel), dtype=torch.float32).to(eng.device, non_blocking=True)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    out = eng.iaf_generate(mel_d, seed=seed, want=('wav',), check_range=True)['wav']
    ev1.record()
    host = torch.empty(out.shape, dtype=torch.float32, pin_memory=True)
    host.copy_(out, non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    return host, done, (ev0, ev1)


# cli.run pipelines this driver: reader thread (files -> numpy) | GPU stage (generate_async) | writer thread (wav files)
synthesis.generate_async = generate_async
