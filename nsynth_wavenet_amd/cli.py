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
    p.add_argument('--serial', action='store_true',
                   help='(not in the reference) one batch at a time -- load, generate, write -- instead of the overlapped reader / GPU / writer stages')
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


def mel_batch(batch_files, sample_length, data=None):
    """[B,F,80] float32 mel for a batch of .wav (featurised) or .npy (precomputed) files; `data`: the batch array when a
    reader thread has loaded it already."""
    from .wavenet import fastgen
    if data is None:
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
    batches = []
    for start in range(0, len(files), args.batch_size):
        batch_files = files[start:start + args.batch_size]
        save_names = [os.path.join(save_path, 'gen_' + os.path.splitext(os.path.basename(f))[0] + '.wav')
                      for f in batch_files]
        batches.append((batch_files, save_names))
    gen_async = getattr(synth_fn, 'generate_async', None)
    import torch
    if gen_async is not None and torch.cuda.is_available() and not getattr(args, 'serial', False) and len(batches) > 1:
        return run_pipelined(args, hparams, checkpoint_path, batches, gen_async)
    for i, (batch_files, save_names) in enumerate(batches):
        logging.info('generating batch {:d}'.format(i))
        synth_fn(hparams, mel_batch(batch_files, args.sample_length), save_names, checkpoint_path)
    return None


def run_pipelined(args, hparams, checkpoint_path, batches, gen_async, depth=3):
    """The loop of eval_parallel_wavenet.py:52-69 (load batch -> mel -> sess.run -> write) as three overlapped stages: a reader
    thread loads and pads the next batches (fastgen.load_batch), this thread featurises on the device and enqueues the
    generation and the device-to-host copy without synchronising, a writer thread waits for each batch's event and writes its
    gen_<name>.wav files.  The parallel student needs ~1.2 ms of GPU time per 4.8 s utterance: run serially the drop-in CLI
    spends nearly all of its wall time in file I/O.  Returns {'files', 'gpu_ms'} (bench.py's `cli_e2e`)."""
    import queue
    import threading
    import torch
    from .wavenet import fastgen
    loaded, to_write = queue.Queue(maxsize=depth), queue.Queue(maxsize=depth)
    errors = []

    def reader():
        try:
            for batch_files, save_names in batches:
                loaded.put((batch_files, save_names, fastgen.load_batch(batch_files, sample_length=args.sample_length)))
        except BaseException as e:           # noqa: B902 -- handed to the main thread
            errors.append(e)
        finally:
            loaded.put(None)

    stats = {'files': 0, 'gpu_ms': 0.0}

    def writer():
        try:
            while True:
                item = to_write.get()
                if item is None:
                    return
                host, done, (ev0, ev1), save_names = item
                done.synchronize()
                stats['gpu_ms'] += ev0.elapsed_time(ev1)
                fastgen.save_batch(host.numpy(), save_names)
                stats['files'] += len(save_names)
        except BaseException as e:           # noqa: B902
            errors.append(e)

    tr, tw = threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)
    tr.start()
    tw.start()
    i = 0
    while True:
        item = loaded.get()
        if item is None or errors:
            break
        batch_files, save_names, data = item
        logging.info('generating batch {:d}'.format(i))
        host, done, evs = gen_async(hparams, mel_batch(batch_files, args.sample_length, data=data), checkpoint_path)
        while not errors:                     # (a writer that died must not leave this thread blocked on a full queue)
            try:
                to_write.put((host, done, evs, save_names), timeout=0.5)
                break
            except queue.Full:
                pass
        i += 1
    to_write.put(None)
    tw.join()
    if errors:
        raise errors[0]
    return stats
