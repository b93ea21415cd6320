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


def test_bench_traffic_figure_follows_the_source_hash(monkeypatch, tmp_path):
    """roofline.traffic is the committed PMC figure of the dominant kernel ONLY for the kernel sources it was measured
    on: with the measured hash it is that kernel's own entry (not a variant's), with any other hash -- or for a
    precision that has no summary of its own -- it is None."""
    import json
    import bench
    from nsynth_wavenet_amd import build
    dom = bench.DOMINANT_KERNEL
    summary = {'workload': {'batch_per_gpu': 1, 'frames': 384, 'samples': 76800}, 'source_hash': 'a' * 64,
               'kernels': {dom: {'hbm_bytes_per_launch': 61000000, 'mfma_util': 0.31}, dom + '<head>': {'hbm_bytes_per_launch': 99000000}},
               'kernel_us_per_call': {dom: 700.0, 'iaf_cond_h_kernel': 390.0}}
    os.makedirs(tmp_path / 'profiles')
    with open(tmp_path / 'profiles' / (bench.PROFILE_ROUND + '_pmc_summary_f16x3.json'), 'w') as f:
        json.dump(summary, f)
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(build, 'source_hash', lambda: 'a' * 64)
    got = bench.pmc_replay(1, 384, 'f16x3', True)
    assert got['traffic'] == 61000000 and got['mfma_util'] == 0.31 and got['kernel_us_per_call']['iaf_cond_h_kernel'] == 390.0
    assert bench.pmc_replay(8, 384, 'f16x3', True)['traffic'] is None           # another workload
    assert bench.pmc_replay(1, 384, 'f16x3-fused', True)['traffic'] is None     # no summary of its own: nothing borrowed
    monkeypatch.setattr(build, 'source_hash', lambda: 'not-the-measured-sources')
    stale = bench.pmc_replay(1, 384, 'f16x3', True)
    assert stale['traffic'] is None and stale['mfma_util'] is None and stale['kernel_us_per_call'] is None
    assert bench.clock_hz_of({'sclk': '(2105Mhz)', 'mclk': '(2000Mhz)'}) == 2105e6 and bench.clock_hz_of(None) is None
    # the committed summary of this round, when present, must describe the kernel bench.py calls dominant
    real = os.path.join(ROOT, 'profiles', bench.PROFILE_ROUND + '_pmc_summary_f16x3.json')
    if os.path.exists(real):
        d = json.load(open(real))
        assert dom in d['kernels'] and len(d.get('source_hash', '')) == 64
        assert d['kernels'][dom]['hbm_bytes_per_launch'] > 0


def test_bench_roofline_blocks_carry_the_same_three_fractions():
    """Every roofline block of the bench line holds frac_mfma_alg, frac_mfma_exec and frac_hbm_moved (so that blocks and
    rounds compare), an MFMA-bound block's `frac` is the ALGORITHMIC one, and the parts' blocks are built from the in-process
    part timing: checked on the arithmetic with a stub engine (no GPU)."""
    import bench
    from types import SimpleNamespace
    f = bench.three_fracs(2.5e15 * 1e-3, 8e12 * 1e-3 * 0.5, 1e-3)            # 1 ms: 2.5 PFLOP useful, 4 GB moved
    assert abs(f['frac_mfma_alg'] - 1.0) < 1e-12 and abs(f['frac_mfma_exec'] - 3.0) < 1e-12 and abs(f['frac_hbm_moved'] - 0.5) < 1e-12
    assert abs(f['frac_mfma_exec_of_sustained'] - 3.0 * 2500.0 / bench.SUSTAINED_F16_MFMA_TFLOPS) < 1e-9
    f32 = bench.three_fracs(157.3e12 * 1e-3, 0.0, 1e-3, 1, bench.PEAK_F32_MFMA_TFLOPS)
    assert abs(f32['frac_mfma_alg'] - 1.0) < 1e-12 and 'frac_mfma_exec_of_sustained' not in f32
    hp = SimpleNamespace(deconv_width=256, width=64, num_iaf_layers=[10, 10, 10, 30], use_share_deconv=True,
                         deconv_config=[[40, 10], [80, 20]], use_resize_conv=False)
    eng = SimpleNamespace(precision='f16x3')
    B, F, T = 1, 384, 76800
    parts = {'prologue_epilogue': 18.0, 'upsampler': 140.0, 'cond_gemm': 380.0, 'residual_stack': 660.0}
    out = bench.part_rooflines(eng, hp, B, F, T, parts, {'kernels': {'iaf_cond_h_kernel': {'hbm_bytes_per_launch': 1500000000, 'mfma_util': 0.6}},
                                                        'file': 'profiles/x.json'})
    c, d = out['roofline_cond'], out['roofline_deconv']
    assert abs(c['flop_per_call'] - 2.0 * 4096 * 256 * T) < 1 and c['traffic'] == 1500000000 and c['mfma_util'] == 0.6
    assert abs(c['frac'] - c['frac_mfma_alg']) < 1e-12 and abs(c['frac_mfma_exec'] - 3 * c['frac_mfma_alg']) < 1e-12
    assert abs(c['achieved'] - c['flop_per_call'] / 380e-6 / 1e12) < 1e-6 and c['bound'] == 'mfma'
    # upsampler: 2 * 256 * 4 taps * (80 * 3840 + 256 * 76800) MACs
    assert abs(d['flop_per_call'] - 2.0 * 256 * 4 * (80 * 3840 + 256 * 76800)) < 1 and d['us_per_call'] == 140.0
    for blk in (c, d):
        assert all(k in blk for k in ('frac_mfma_alg', 'frac_mfma_exec', 'frac_hbm_moved', 'algorithmic_TFLOPs', 'executed_TFLOPs', 'moved_GBps'))
    assert bench.HwmonSampler(0).summary() is None or True                     # (no GPU here: the sampler must not raise)


def _cli_worker(rank, world, port, src, dst, ckpt, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from argparse import Namespace
    from nsynth_wavenet_amd import cli
    seen = []

    def synth(hparams, mel, save_names, checkpoint_path):
        assert mel.ndim == 3 and mel.shape[2] == 80 and mel.shape[0] == len(save_names)
        seen.append((list(save_names), float(mel[:, 0, 0].sum())))
        for n in save_names:
            open(n, 'w').write('rank %d' % rank)

    cli.run(Namespace(ckpt_dir=ckpt, source_path=src, save_path=dst, sample_length=-1, batch_size=2, npy_only=False,
                      log='ERROR', gpu_id='0'), synth)
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_cli_shards_files_over_two_ranks(tmp_path):
    """cli.run under WORLD_SIZE=2 (gloo, no GPU): the sorted file list splits contiguously over the ranks, every rank
    batches ITS files by --batch_size and names the outputs gen_<basename>.wav (the reference's naming,
    eval_parallel_wavenet.py:52-69), nothing is generated twice and nothing is left out."""
    src, dst, ckpt = tmp_path / 'in', tmp_path / 'out', tmp_path / 'ckpt'
    for d in (src, ckpt):
        os.makedirs(d)
    for i in range(5):
        np.save(src / ('utt%d.npy' % i), np.full([7, 80], float(i), np.float32))
    import json
    json.dump(load_json('parallel_wavenet.json'), open(ckpt / 'parallel_wavenet.json', 'w'))
    hp = cfg.load_hparams(dict(load_json('parallel_wavenet.json'), num_iaf_layers=[1]))
    np.savez(ckpt / 'model.ckpt-7.npz', **wts.synthetic_weights(hp, seed=3))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, str(src), str(dst), str(ckpt), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    names = lambda r: [os.path.basename(n) for batch, _ in res[r] for n in batch]     # noqa: E731
    assert names(0) == ['gen_utt0.wav', 'gen_utt1.wav', 'gen_utt2.wav'] and names(1) == ['gen_utt3.wav', 'gen_utt4.wav']
    assert [len(b) for b, _ in res[0]] == [2, 1] and [len(b) for b, _ in res[1]] == [2]
    assert [s for _, s in res[0]] == [1.0, 2.0] and [s for _, s in res[1]] == [7.0]
    assert sorted(os.listdir(dst)) == ['gen_utt%d.wav' % i for i in range(5)]
    assert open(dst / 'gen_utt4.wav').read() == 'rank 1'
