"""Multi-GPU generation: utterances shard across ranks, nothing else is shared.

The reference has no multi-device generation (it picks one GPU through
CUDA_VISIBLE_DEVICES, eval_parallel_wavenet.py:12); every utterance of the batch is
independent in ParallelWavenet.feed_forward / Fastgen.sample, so the batch splits
contiguously over one process per GPU with NO collective on the data path.  The
only communication is a one-time broadcast of the weight blob from rank 0 over
RCCL/xGMI (backend "nccl"; "gloo" in the CPU tests) so that only rank 0 has to read
the checkpoint, and an optional gather of the generated audio to rank 0.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import weights as wts


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def init_process_group(backend=None):
    """One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(array, rank, world):
    lo, hi = shard_range(len(array), rank, world)
    return array[lo:hi]


def pack_weights(weights, hp, kind=None):
    """name->array dict -> one flat float32 vector in expected_variables order."""
    parts = [np.asarray(weights[name], np.float32).reshape(-1)
             for name, _ in wts.expected_variables(hp, kind)]
    return np.concatenate(parts) if parts else np.zeros([0], np.float32)


def unpack_weights(flat, hp, kind=None):
    out, o = {}, 0
    for name, shape in wts.expected_variables(hp, kind):
        n = int(np.prod(shape))
        out[name] = np.asarray(flat[o:o + n], np.float32).reshape(shape)
        o += n
    if o != len(flat):
        raise ValueError('weight blob has {} floats, the model needs {}'.format(len(flat), o))
    return out


def broadcast_weights(weights, hp, kind=None, src=0, device=None):
    """Rank `src` passes its weight dict (others pass None); every rank returns the dict.
    ONE broadcast of the packed blob (32 MB student / 159 MB teacher)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return weights
    n = sum(int(np.prod(s)) for _, s in wts.expected_variables(hp, kind))
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' \
            else torch.device('cpu')
    if dist.get_rank() == src:
        flat = torch.from_numpy(pack_weights(weights, hp, kind)).to(device)
        assert flat.numel() == n
    else:
        flat = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(flat, src=src)
    if dist.get_rank() == src:
        return weights
    return unpack_weights(flat.cpu().numpy(), hp, kind)


def gather_audio(local_audio, n_total, dst=0):
    """Collect the per-rank [b_i, T] results on rank `dst` in utterance order
    (only needed when one process writes every file).  Returns None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_audio
    world, rank = dist.get_world_size(), dist.get_rank()
    T = local_audio.shape[1]
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    mx = max(counts)
    pad = torch.zeros((mx, T), dtype=local_audio.dtype, device=local_audio.device)
    pad[:local_audio.shape[0]] = local_audio
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    if dist.get_backend() == 'nccl':
        allb = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(allb, pad)
        bufs = allb
    else:
        dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
