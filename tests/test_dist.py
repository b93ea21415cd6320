"""world_size-2 gloo tests (CPU) of the multi-GPU path: utterance sharding, the one-time
weight broadcast and the optional audio gather.  No data-path collective exists."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_json
from nsynth_wavenet_amd import config as cfg
from nsynth_wavenet_amd import dist as wdist
from nsynth_wavenet_amd import weights as wts


def test_shard_range_is_a_contiguous_partition():
    for n in (0, 1, 7, 8, 64, 65, 128):
        for world in (1, 2, 3, 8):
            spans = [wdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [wdist.shard_range(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]     # config 3: 64 over 8
    assert wdist.shard_range(128, 3, 8) == (48, 64)                                 # config 5: 128 over 8


def test_pack_unpack_round_trip():
    hp = cfg.load_hparams(dict(load_json('parallel_wavenet.json'), num_iaf_layers=[2, 1]))
    w = wts.synthetic_weights(hp, seed=5)
    flat = wdist.pack_weights(w, hp)
    assert flat.dtype == np.float32 and flat.size == sum(v.size for v in w.values())
    back = wdist.unpack_weights(flat, hp)
    assert all(np.array_equal(back[k], w[k]) for k in w)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, w_, _ = wdist.init_process_group('gloo')
    assert (r, w_) == (rank, world)
    hp = cfg.load_hparams(dict(load_json('parallel_wavenet.json'), num_iaf_layers=[1, 1]))
    weights = wts.synthetic_weights(hp, seed=11) if rank == 0 else None
    got = wdist.broadcast_weights(weights, hp, src=0)
    ref = wts.synthetic_weights(hp, seed=11)
    ok = all(np.array_equal(got[k], ref[k]) for k in ref)
    # shard 5 utterances over 2 ranks, "generate" (tag with the utterance id), gather on rank 0
    n = 5
    lo, hi = wdist.shard_range(n, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 16)
    allw = wdist.gather_audio(local, n, dst=0)
    if rank == 0:
        ok = ok and allw.shape == (n, 16) and torch.equal(allw[:, 0], torch.arange(n, dtype=torch.float32))
    else:
        ok = ok and allw is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_broadcast_and_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no rendezvous in the environment (the command the driver uses) must start
    its N ranks itself and print ONE JSON line from rank 0.  Driven here with the --stub step (torch-CPU, gloo)
    so that the launcher, the rendezvous on 127.0.0.1 and the timing protocol run on a machine without GPUs."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--stub'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['world_size_seen'] == 2 and rec['steps'] == 3 and rec['warmup'] == 1
    # under an existing rendezvous with a mismatching world size the script refuses instead of guessing
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub'], env=env2,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert bad.returncode != 0


def test_bench_traffic_figure_follows_the_source_hash(monkeypatch):
    """roofline.traffic is the committed PMC figure of the dominant kernel ONLY for the kernel sources it was measured
    on: with the tree's hash it is the layer kernel's own entry (not the head variant's, whatever template flags the
    kernel name carries), with any other hash it is None."""
    import json
    import bench
    from nsynth_wavenet_amd import build
    path = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), 'profiles', 'r02_pmc_summary_f16x3.json')
    d = json.load(open(path))
    assert 'iaf_layer_c_kernel' in d['kernels'] and 'iaf_layer_c_kernel<head>' in d['kernels']
    plain, head = d['kernels']['iaf_layer_c_kernel'], d['kernels']['iaf_layer_c_kernel<head>']
    assert 0.9 * 58982400 <= plain['hbm_bytes_per_launch'] <= 1.3 * 58982400      # 768 B/sample: no wasted re-reads
    assert head['hbm_bytes_per_launch'] != plain['hbm_bytes_per_launch']
    monkeypatch.setattr(build, 'source_hash', lambda: d['source_hash'])
    assert bench.pmc_traffic(1, 384, 'f16x3', True) == plain['hbm_bytes_per_launch']
    monkeypatch.setattr(build, 'source_hash', lambda: 'not-the-measured-sources')
    assert bench.pmc_traffic(1, 384, 'f16x3', True) is None
