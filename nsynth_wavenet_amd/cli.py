"""Shared body of the two generation CLIs (eval_parallel_wavenet.py, eval_wavenet.py).

Keeps the reference's command-line surface (eval_parallel_wavenet.py:70-92 /
eval_wavenet.py:72-94 there): --ckpt_dir --source_path --save_path --sample_length
--batch_size --npy_only --log --gpu_id; output files are gen_<basename>.wav, float32,
16 kHz.  `.wav` inputs go through the mel featuriser; `.npy` inputs are taken as
precomputed mels [frames, 80].  With torch.distributed.run the file list is sharded
over ranks (one process per GPU, no data-path collective).
"""
import glob
import json
import logging
import os
from argparse import ArgumentParser, Namespace

import numpy as np

from . import dist as wdist
from . import weights as wts
from .auxilaries import mel_extractor, utils


def build_parser(description):
    p = ArgumentParser(description=description)
    p.add_argument('--ckpt_dir', required=True, help='Directory with the checkpoint and its single *.json config.')
    p.add_argument('--source_path', required=True,
                   help='A .wav/.npy file or a directory of them (.wav preferred when both exist).')
    p.add_argument('--save_path', required=True, help='Output directory.')
    p.add_argument('--sample_length', default=-1, type=int, help='Max input length in samples (-1: whole file).')
    p.add_argument('--batch_size', default=1, type=int, help='Utterances per batch.')
    p.add_argument('--npy_only', default=False, type=bool, help='If True, use only .npy files.')
    p.add_argument('--log', default='INFO', help='DEBUG, INFO, WARN, ERROR, or FATAL.')
    p.add_argument('--gpu_id', default='0', help='GPU used for generation (ignored under torch.distributed.run).')
    return p


def resolve_model(ckpt_dir):
    if not os.path.isdir(ckpt_dir):
        raise AssertionError('{} is not a directory'.format(ckpt_dir))
    checkpoint_path = wts.latest_checkpoint(ckpt_dir)
    if checkpoint_path is None:
        raise AssertionError('no checkpoint found in {}'.format(ckpt_dir))
    jsons = glob.glob(os.path.join(ckpt_dir, '*.json'))
    if len(jsons) != 1:
        raise AssertionError('expected exactly one *.json in {}, found {}'.format(ckpt_dir, len(jsons)))
    with open(jsons[0], 'rt') as f:
        return Namespace(**json.load(f)), checkpoint_path


def list_sources(source_path, npy_only):
    if os.path.isdir(source_path):
        names = os.listdir(source_path)
        exts = {os.path.splitext(n)[1] for n in names}
        if '.wav' in exts:
            postfix = '.wav'
        elif '.npy' in exts:
            postfix = '.npy'
        else:
            raise RuntimeError('Folder must contain .wav or .npy files.')
        if npy_only:
            postfix = '.npy'
        return sorted(os.path.join(source_path, n) for n in names if n.lower().endswith(postfix))
    if source_path.lower().endswith(('.wav', '.npy')):
        return [source_path]
    return []


def mel_batch(batch_files, sample_length):
    """[B,F,80] float32 mel for a batch of .wav (featurised) or .npy (precomputed) files."""
    from .wavenet import fastgen
    data = fastgen.load_batch(batch_files, sample_length=sample_length)
    if batch_files[0].lower().endswith('.npy'):
        if data.ndim != 3:
            raise ValueError('.npy inputs must be mel arrays [frames, {}]'.format(mel_extractor.NUM_MEL))
        return data.astype(np.float32)
    import torch
    if torch.cuda.is_available():          # featurise on the device the generation runs on
        return mel_extractor.batch_melspectrogram_device(data)
    return mel_extractor.batch_melspectrogram(data)


def run(args, synth_fn):
    """synth_fn(hparams, mel [B,F,80], save_names, checkpoint_path)."""
    rank, world, local = wdist.env_rank_world()
    if world > 1:
        import torch
        wdist.init_process_group()           # nccl (= RCCL) with a GPU, gloo without (host-only tests)
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
    else:
        os.environ.setdefault('HIP_VISIBLE_DEVICES', str(args.gpu_id))
    logging.basicConfig(level=getattr(logging, str(args.log).upper().replace('WARN', 'WARNING').replace(
        'WARNINGING', 'WARNING').replace('FATAL', 'CRITICAL'), logging.INFO))
    source_path = utils.shell_path(args.source_path)
    ckpt_dir = utils.shell_path(args.ckpt_dir)
    save_path = utils.shell_path(args.save_path)
    if not os.path.exists(save_path):
        logging.info('save_path does not exist, make it.')
        os.makedirs(save_path, exist_ok=True)
    hparams, checkpoint_path = resolve_model(ckpt_dir)
    files = list_sources(source_path, args.npy_only)
    lo, hi = wdist.shard_range(len(files), rank, world)
    files = files[lo:hi]
    for start in range(0, len(files), args.batch_size):
        logging.info('generating batch {:d}'.format(start // args.batch_size))
        batch_files = files[start:start + args.batch_size]
        save_names = [os.path.join(save_path, 'gen_' + os.path.splitext(os.path.basename(f))[0] + '.wav')
                      for f in batch_files]
        synth_fn(hparams, mel_batch(batch_files, args.sample_length), save_names, checkpoint_path)
