"""Headline benchmark: 16 kHz audio samples/s of Parallel-WaveNet (IAF student)
generation on MI355X -- BASELINE.json's metric on its configs[1]
("parallel_wavenet.json IAF student gen, 1 MI355X, batch=1 synthetic 80-dim mel").

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (it re-executes
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`);
started under torch.distributed.run it uses the ranks it is given.  One process per GPU, RCCL.

A step = one pass of the whole generation hot path (noise draw, mel upsampler, conditioning GEMM,
4 IAF flows = 60 residual layers + 4 heads, clip/quantise) over one batch of synthetic mels already
resident in HBM.  Utterances are independent, so N ranks each generate their own batch with no
data-path collective (weak scaling); the only communication is the one-time weight broadcast from
rank 0 (untimed).  Rank 0 prints ONE JSON line.

Beside the contract's fields the line carries (N = 1): `roofline` (the dominant kernel, HIP events inside the library around
its launches), `roofline_cond` / `roofline_deconv` (conditioning GEMM, upsampler), `roofline_b8` (eight utterances per GPU),
`roofline_f32` (the reference's own arithmetic) -- each with the same three fractions `frac_mfma_alg`, `frac_mfma_exec`,
`frac_hbm_moved`; `kernel_us_per_call` (HIP events at the part boundaries, this process); `power` (package power, shader clock
and energy per sample of a sustained run: the workload runs at the part's power cap); `ar_b1`, `ar_b64`,
`teacher_forward` (BASELINE configs[3] and the teacher's full-sequence forward); `e2e_ms_per_step` (PCIe-inclusive call);
`cpu_baseline`.  `--no-extras` keeps only the contract's fields and `roofline`.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nsynth_wavenet_amd import build as wbuild           # noqa: E402
from nsynth_wavenet_amd import config as cfg            # noqa: E402
from nsynth_wavenet_amd import dist as wdist             # noqa: E402
from nsynth_wavenet_amd import weights as wts            # noqa: E402

# per generated sample (BASELINE.md section 2 / SURVEY section 8d)
LAYER_FLOP_PER_SAMPLE = 61440          # one residual layer: 2 * (12288 + 16384 + 2048) MAC
LAYER_BYTES_PER_SAMPLE = 1536          # read l 256 B + read mel_en 1024 B + write l 256 B
LAYER_BYTES_PER_SAMPLE_HOISTED = 768   # read l 256 B + read hoisted cond term 256 B + write l 256 B
PATH_FLOP_PER_SAMPLE = 4385280
PATH_BYTES_PER_SAMPLE = 98484
PEAK_F32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 peak
PEAK_F16_MFMA_TFLOPS = 2500.0          # dense fp16/bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0
# What the matrix pipes SUSTAIN on this part with operands that look like data: a bare v_mfma_f32_16x16x32_f16 loop on all
# 1 024 SIMDs settles at ~1.92 GHz / ~1 270 W (scripts/ubench/mfma_power.hip, profiles/r05_mfma_power_ubench.txt; zeros: 2.38 GHz).
# Reported BESIDE the fractions of the nominal peak, never instead of them.
SUSTAINED_F16_MFMA_TFLOPS = 1657.0
BARRIER_KW = {}
RED_DEV = None
PROFILE_ROUND = 'r06'                  # profiles/<round>_pmc_summary_*.json hold the PMC passes of this round's kernels
DOMINANT_KERNEL = 'iaf_group_kernel'    # layer groups at one / two utterances; 'iaf_layer_c_kernel' when every layer is a launch
GROUP_LAYERS = 5                       # residual layers per launch of the group kernel (one half of a dilation cycle)
# One 16-sample block of one residual layer on a gfx950 SIMD: 84 x v_mfma_f32_16x16x32_f16 = 1344 cycles of the matrix
# pipe.  Round 3 added the epilogue's VALU time to this "floor" on the strength of a micro-benchmark that said the two
# pipes serialise; round 4's hand-placed streams (scripts/ubench/issue_overlap.hip, profiles/r04_issue_overlap_ubench.txt)
# show that VALU work DOES issue beside a running MFMA, within a wave and between waves, so the floor of the block is the
# matrix pipe alone; the epilogue arithmetic beside a saturated pipe needs ~1.7 k cycles of issue and is the longer pole
# of a perfectly overlapped layer.
BLOCK_LAYER_MFMA_CYCLES = 84 * 16


def pmc_replay(B, F, precision='f16x3', hoisted=False, kernel=None):
    """(HBM bytes per launch, MFMA utilisation, per-kernel microseconds of a call) of the dominant layer kernel from the committed rocprofv3 PMC passes
    (profiles/r0*_pmc_summary*.json; FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE).
    PMC counters cannot be collected from inside this process, so this is the value of the profiled
    run of the SAME command -- and only when that profile was taken from the kernel sources this
    process runs: a summary carries the hash of csrc/ + include/ (build.source_hash) it was measured
    on; a summary without a hash, or with another one, is stale and gives None."""
    none = {'traffic': None, 'mfma_util': None, 'kernel_us_per_call': None, 'file': None}
    if precision in ('f32', 'f32-hoisted') and hoisted:
        names, kernel = [PROFILE_ROUND + '_pmc_summary_f32.json'], kernel or 'iaf_layer_kernel<hoist>'
    elif precision.startswith('f32'):
        names, kernel = ['r01_pmc_summary.json'], 'iaf_layer_kernel'
    elif precision in ('f16x3', 'f16x3-hoisted') and hoisted:
        names = [PROFILE_ROUND + '_pmc_summary_f16x3.json', PROFILE_ROUND + '_pmc_summary_f16x3_batch8.json']
        kernel = kernel or DOMINANT_KERNEL
    elif precision in ('f16x3', 'f16x3-fused') and not hoisted:
        names, kernel = [PROFILE_ROUND + '_pmc_summary_f16x3_fused.json'], 'iaf_layer_h_kernel'
    else:
        return none                      # no PMC summary of its own: never borrow another kernel's figure
    have = wbuild.source_hash()
    for name in names:
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                d = json.load(f)
            w = d['workload']
            if (w['batch_per_gpu'], w['frames']) == (B, F) and d.get('source_hash') == have:
                k = d['kernels'][kernel]
                return {'traffic': k['hbm_bytes_per_launch'], 'mfma_util': k.get('mfma_util'), 'kernels': d['kernels'],
                        'kernel_us_per_call': d.get('kernel_us_per_call') or None, 'file': 'profiles/' + name}
        except (OSError, KeyError, ValueError):
            pass
    return {'traffic': None, 'mfma_util': None, 'kernel_us_per_call': None, 'file': None}


def cpu_baseline(hp_dict, frames, budget_s=25.0):
    """Reference-shaped CPU port (oracle/torch_ref.py, torch-CPU fp32, all host cores) on
    the same config-2 utterance.  The reference's TF CPU path cannot run (no TensorFlow)."""
    from oracle import wavenet_np as O
    from oracle.torch_ref import StudentRef
    hp = O.HP(hp_dict)
    ncpu = os.cpu_count() or 1
    w = O.synth_weights(hp, 'student', seed=1234, init='tf')
    ref = StudentRef(w, hp)
    # these are small convolutions (64 channels): oneDNN stops scaling well before a
    # many-core host is full, so probe a short utterance and keep the fastest thread count
    probe_mel = np.random.RandomState(1).uniform(0, 1, [1, 24, 80]).astype(np.float32)
    probe_noise = np.zeros([1, O.iaf_length(24, hp)], np.float32)
    best, cores = None, 1
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        ref.parallelgen(probe_mel, probe_noise)
        t0 = time.time()
        ref.parallelgen(probe_mel, probe_noise)
        dt = time.time() - t0
        if best is None or dt < best:
            best, cores = dt, nt
    torch.set_num_threads(cores)
    if best * (frames / 24.0) > 8.0:                 # keep the whole leg to ~10-30 s of CPU work
        frames = 96
    mel = np.random.RandomState(12345).uniform(0, 1, [1, frames, 80]).astype(np.float32)
    T = O.iaf_length(frames, hp)
    u = np.random.RandomState(12346).uniform(1e-5, 1 - 1e-5, [1, T])
    noise = O.logistic_from_uniform(u, np.float32)
    t0 = time.time()
    ref.parallelgen(mel, noise)                     # warm-up
    first = time.time() - t0
    times = []
    while len(times) < 5 and (sum(times) + first) < budget_s:
        t0 = time.time()
        ref.parallelgen(mel, noise)
        times.append(time.time() - t0)
    med = float(np.median(times)) if times else first
    return {'value': T / med, 'unit': 'samples/s', 'cores': cores, 'host_cpus': ncpu, 'kind': 'port',
            'sample': 'torch-CPU fp32 restatement (oracle/torch_ref.py), 1 utterance F={} T={} batch 1, '
                      'warm-up + median of {} runs; the reference TF path cannot run here'.format(
                          frames, T, max(len(times), 1)),
            'x_realtime': T / med / 16000.0}


def teacher_extras(dev, ar_samples):
    """`ar_b1`, `ar_b64`, `teacher_forward`: BASELINE.json configs[3] (wavenet_mol.json fastgen) at one and at 64 utterances and
    Wavenet.feed_forward at 4.8 s, timed like bench_aux.py does (synthetic conditioning, random-init weights, Philox
    sampling on the device).  One sample step of the AR path is a chain of dependent launches; its roofline is the
    weight stream per step."""
    from nsynth_wavenet_amd.engine import Engine
    with open(os.path.join(ROOT, 'config_jsons', 'wavenet_mol.json')) as f:
        hp = cfg.load_hparams(json.load(f))
    eng = Engine(hp, kind='teacher', device=dev).load_weights(wts.synthetic_weights(hp, 'teacher', seed=1, init='unit'))
    W, S, Cd = hp.width, hp.skip_width, hp.deconv_width
    G = cfg.teacher_gate_width(hp)
    OW = cfg.teacher_out_width(hp)
    rs = np.random.RandomState(0)
    out = {}
    # weights streamed once per step (fp32): gate + composite, res/skip, head
    wbytes = 4.0 * (hp.num_layers * (G * (3 * W + Cd) + (W + S) * (G // 2)) + (hp.num_layers - 1) * G * (G // 2) +
                    S * W + S * (S + Cd) + OW * S)
    mac_step = wbytes / 4.0
    for B, key in ((1, 'ar_b1'), (64, 'ar_b64'), (256, 'ar_b256')):
        Tn = ar_samples if B <= 64 else max(160, ar_samples // 4)
        enc = torch.as_tensor((rs.standard_normal([B, Tn, Cd]) * 0.1).astype(np.float32)).to(dev)
        eng.ar_generate(enc[:, :64], None, seed=1, use_graph=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        o = eng.ar_generate(enc, None, seed=2, use_graph=False)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(o['wav']).all())
        us_step = dt / Tn * 1e6
        r = {'workload': 'BASELINE.json configs[3]: wavenet_mol.json autoregressive fastgen, {} utterance(s) x {} samples'.format(B, Tn),
             'samples_per_sec': B * Tn / dt, 'us_per_sample_step': us_step, 'x_realtime_per_utterance': Tn / dt / 16000.0,
             'x_realtime_aggregate': B * Tn / dt / 16000.0,
             'launches_per_step': (hp.num_layers + 4) if B < 4 else (2 * hp.num_layers + 5), 'dtype': 'f32',
             'bound': 'hbm', 'achieved': wbytes / (us_step * 1e-6) / 1e9, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
             'frac': wbytes / (us_step * 1e-6) / 1e9 / PEAK_HBM_GBPS, 'traffic': None,
             'note': 'weight bytes streamed per sample step / step time; the step is a chain of dependent launches '
                     '(launch-latency bound, DESIGN.md 3.4)'}
        r.update(three_fracs(2.0 * mac_step * B, wbytes, us_step * 1e-6, 1, PEAK_F32_MFMA_TFLOPS))
        out[key] = r
    # Round 6: independent utterance groups as chains on their own streams against ONE handle (Engine.fork per host thread,
    # plain launches).  The single-stream timeline has a kernel resident 94 % of the time (profiles/r06_ar_streams_timeline.txt):
    # the step's kernels are latency-bound themselves, the GPU does not idle between them -- two chains overlap for half of
    # the span and buy ~1.2x; the batch is the lever (ar_b256 above: 3x ar_b64; 1 024 utterances: 5x, profiles/r06_ar_batch_and_streams.txt).
    try:
        import threading
        B, G, Tn = 64, 2, ar_samples
        enc = torch.as_tensor((rs.standard_normal([B, Tn, Cd]) * 0.1).astype(np.float32)).to(dev)
        parts = [enc[B * g // G: B * (g + 1) // G].contiguous() for g in range(G)]
        forks = [eng.fork() for _ in range(G)]
        streams = [torch.cuda.Stream(dev) for _ in range(G)]

        def work(g, n, seed):
            with torch.cuda.stream(streams[g]):
                forks[g].ar_generate(parts[g][:, :n], None, seed=seed + g, use_graph=False)

        def run_all(n, seed):
            ths = [threading.Thread(target=work, args=(g, n, seed)) for g in range(G)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            torch.cuda.synchronize(dev)
        run_all(64, 1)
        t0 = time.perf_counter()
        run_all(Tn, 7)
        dt = time.perf_counter() - t0
        out['ar_b64_s2'] = {'workload': 'as ar_b64, the 64 utterances as {} independent groups: {} host threads, one stream and one queue state each, ONE shared handle, plain launches'.format(G, G),
                            'samples_per_sec': B * Tn / dt, 'us_per_sample_step': dt / Tn * 1e6, 'x_realtime_aggregate': B * Tn / dt / 16000.0,
                            'vs_ar_b64': (B * Tn / dt) / out['ar_b64']['samples_per_sec'],
                            'note': 'profiles/r06_ar_streams_timeline.txt: kernels of the two chains overlap for 0.53 of the span, each as long as alone'}
        for f_ in forks:
            f_.close()
    except Exception as e:                   # a labelled extra: never fail the bench over it
        out['ar_b64_s2'] = {'error': repr(e)}
    F = 384
    T = F * cfg.frame_shift(hp)
    mel = torch.as_tensor(rs.uniform(0, 1, [1, F, 80]).astype(np.float32)).to(dev)
    wav = torch.as_tensor(rs.uniform(-1, 1, [1, T]).astype(np.float32)).to(dev)
    for _ in range(2):
        o = eng.teacher_forward(wav, mel)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        o = eng.teacher_forward(wav, mel)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / 5
    assert bool(torch.isfinite(o).all())
    mac = hp.num_layers * (G * (3 * W + Cd) + (W + S) * (G // 2)) + S * W + S * (S + Cd) + OW * S
    flop = 2.0 * mac * T
    # activations a layer must move: l (3 taps read once when cached, written once), m, s read-modify-write, enc read
    moved = 4.0 * T * (hp.num_layers * (2 * W + G // 2 + 2 * S + Cd)) + 4.0 * mac
    r = {'workload': 'wavenet_mol.json Wavenet.feed_forward, 1 utterance of {} frames = {} samples'.format(F, T),
         'samples_per_sec': T / dt, 'ms_per_call': dt * 1e3, 'x_realtime': T / dt / 16000.0,
         'dtype': 'split-fp16 (3 fp16 MFMAs per product), fp32 accumulate', 'bound': 'mfma', 'achieved': flop / dt / 1e12,
         'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': flop / dt / 1e12 / PEAK_F16_MFMA_TFLOPS, 'traffic': None}
    r.update(three_fracs(flop, moved, dt))
    out['teacher_forward'] = r
    eng.close()
    return out


def cli_e2e(hp_dict, weights, n_files=256, batch=8, frames=384):
    """End-to-end figure of the drop-in CLI (eval_parallel_wavenet.py:52-69: a directory of wavs in, gen_<name>.wav out):
    `n_files` synthetic int16 utterances of 4.8 s on a tmpfs, a checkpoint directory in the reference's layout (single *.json,
    `checkpoint` state file, EMA-keyed tensors), then cli.run -- serial (load, mel, generate, D2H, write per batch, what the
    reference's loop does) and with the reader / GPU / writer stages overlapped.  files/s, x real time, and the share of the
    wall time the GPU was generating (HIP events around the generate calls)."""
    import shutil
    import tempfile
    from argparse import Namespace
    from scipy.io import wavfile
    from nsynth_wavenet_amd import cli
    from nsynth_wavenet_amd.wavenet import parallelgen
    base = '/dev/shm' if os.path.isdir('/dev/shm') else None
    root = tempfile.mkdtemp(prefix='wn_cli_e2e_', dir=base)
    try:
        src, ck = os.path.join(root, 'wavs'), os.path.join(root, 'ckpt')
        os.makedirs(src)
        os.makedirs(ck)
        n = frames * 200 - 100                                 # 1 + n // 200 = `frames` mel frames
        rs = np.random.RandomState(5)
        t = np.arange(n) / 16000.0
        for i in range(n_files):
            y = 0.3 * np.sin(2 * np.pi * (110.0 + 3.0 * i) * t) + 0.05 * rs.standard_normal(n)
            wavfile.write(os.path.join(src, 'utt_%04d.wav' % i), 16000, (np.clip(y, -1, 1) * 32767).astype(np.int16))
        wts.save_checkpoint(os.path.join(ck, 'model.ckpt-1'), weights, cfg.load_hparams(hp_dict))
        with open(os.path.join(ck, 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "model.ckpt-1"\n')
        with open(os.path.join(ck, 'parallel_wavenet.json'), 'w') as f:
            json.dump(hp_dict, f)
        out = {}
        T = None
        for mode in ('serial', 'pipelined'):
            dst = os.path.join(root, 'out_' + mode)
            args = Namespace(ckpt_dir=ck, source_path=src, save_path=dst, sample_length=-1, batch_size=batch, npy_only=False,
                             log='ERROR', gpu_id=os.environ.get('HIP_VISIBLE_DEVICES', '0'), serial=(mode == 'serial'))
            if mode == 'serial':                               # engine creation + checkpoint load outside the timed loops
                parallelgen.load_parallelgen(*cli.resolve_model(ck))
                warm = Namespace(**dict(vars(args), source_path=os.path.join(src, 'utt_0000.wav'), save_path=os.path.join(root, 'warm')))
                cli.run(warm, parallelgen.synthesis)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st = cli.run(args, parallelgen.synthesis)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            names = sorted(os.listdir(dst))
            assert len(names) == n_files and names[0] == 'gen_utt_0000.wav', (mode, len(names))
            sr, a = wavfile.read(os.path.join(dst, names[-1]))
            T = int(a.shape[0])
            assert sr == 16000 and a.dtype == np.float32 and T == (frames * 200 // 512) * 512
            out[mode] = {'seconds': dt, 'files_per_sec': n_files / dt, 'x_realtime': n_files * T / 16000.0 / dt,
                         'gpu_busy_frac': (st['gpu_ms'] * 1e-3 / dt) if st else None}
        out.update({'files': n_files, 'batch_size': batch, 'samples_per_file': T, 'storage': 'tmpfs' if base else 'tmp dir',
                    'note': 'eval_parallel_wavenet.py end to end on this host: int16 wav in, device mel, generate, float32 wav out; '
                            'serial = the reference\'s loop (eval_parallel_wavenet.py:52-69), pipelined = reader thread | GPU stage | '
                            'writer thread (cli.run_pipelined); gpu_busy_frac = HIP-event time of the generate calls over wall time'})
        return out
    finally:
        shutil.rmtree(root, ignore_errors=True)


F16X2_LIBS = (('three terms (the shipped arithmetic)', 'libwnhip.so'),
              ('two terms in the conditioning GEMM', 'libwnhip_f16x2c.so'),
              ('two terms in the conditioning GEMM, the residual stack and the heads', 'libwnhip_f16x2.so'))


def f16x2_probe():
    """Child process of the `roofline_f16x2` extra (WN_LIB_PATH names the library): distance from the REFERENCE CODE's vectors
    (tests/golden/ref_float_full.npz: configs[1] at full size; ref_float.npz: the small student cases incl. the unit-gain one on
    a +-28 range) and sustained time / power / energy per call, as one JSON line."""
    from nsynth_wavenet_amd.engine import Engine
    gold = os.path.join(ROOT, 'tests', 'golden')
    RF = np.load(os.path.join(gold, 'ref_float_full.npz'))
    cfgd = json.loads(str(RF['full/in_cfg_json']))
    hp = cfg.load_hparams(cfgd)
    w = wts.synthetic_weights(hp, seed=int(RF['full/in_seed']), init=str(RF['full/in_init']))
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    eng = Engine(cfgd, device=dev).load_weights(w)
    u = np.random.RandomState(12346).uniform(1e-5, 1 - 1e-5, [1, 76800]).astype(np.float32).astype(np.float64)
    noise = (np.log(u) - np.log(1.0 - u)).astype(np.float32)
    mel = torch.from_numpy(RF['full/in_mel']).to(dev)
    out = eng.iaf_generate(mel, noise, want=('x', 'idx'))
    x_ref = RF['full/x_f64']
    res = {'full_size_max_abs_err': float(np.abs(out['x'].cpu().numpy() - x_ref).max()), 'full_size_range': float(np.abs(x_ref).max()),
           'full_size_index_flips': int((out['idx'].cpu().numpy().astype(np.int64) != RF['full/idx_i16'].astype(np.int64)).sum())}
    R = np.load(os.path.join(gold, 'ref_float.npz'))
    small = {}
    for tag in ('iaf_logistic_tf', 'iaf_logistic_unit', 'iaf_gauss_perflow', 'iaf_mulaw'):
        g = np.load(os.path.join(gold, tag + '.npz'))
        c2 = json.loads(str(g['cfg_json']))
        e2 = Engine(c2, device=dev).load_weights(wts.synthetic_weights(cfg.load_hparams(c2), seed=int(g['seed']), init=str(g['init'])))
        x = e2.iaf_generate(g['mel'], R[tag + '/rand_input_f64'].astype(np.float32), want=('x',), check_range=False)['x'].cpu().numpy()
        ref = R[tag + '/x_f64']
        small[tag] = {'max_abs_err': float(np.abs(x - ref).max()), 'range': float(np.abs(ref).max())}
        e2.close()
    res['small_cases'] = small
    for i in range(40):
        eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
    torch.cuda.synchronize()
    pw = measure_power(eng, mel, 0, 0, seconds=1.5)
    if pw:
        res.update({'ms_per_call_sustained': pw['ms_per_step_sustained'], 'avg_W': pw['avg_W'], 'avg_sclk_MHz': pw['avg_sclk_MHz'],
                    'J_per_call': pw['J_per_step']})
    else:
        t0 = time.perf_counter()
        for i in range(200):
            eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
        torch.cuda.synchronize()
        res['ms_per_call_sustained'] = (time.perf_counter() - t0) / 200 * 1e3
    eng.close()
    print(json.dumps(res), flush=True)


def roofline_f16x2():
    """`roofline_f16x2`: what the unspent error budget would buy -- NARROWER THAN THE REFERENCE, NOT THE METRIC.  The
    contract allows 1e-3 max-abs; the shipped split-fp16 arithmetic (three fp16 MFMAs per product, 22-bit operands) is at
    3e-7.  Two measurement builds drop the term with the activations' lo halves (wh.xh + wl.xh; build.py: build_f16x2); each
    is run in a child process through WN_LIB_PATH and held to the vectors the reference's own code produced."""
    out = {'label': 'narrower than the reference: not the metric', 'variants': []}
    lib_dir = os.path.join(ROOT, 'nsynth_wavenet_amd', 'lib')
    for what, name in F16X2_LIBS:
        path = os.path.join(lib_dir, name)
        if not os.path.exists(path):
            out['variants'].append({'arithmetic': what, 'library': name, 'error': 'not built (python -m nsynth_wavenet_amd.build --f16x2)'})
            continue
        env = dict(os.environ, WN_LIB_PATH=path)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--f16x2-probe'], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=300)
            line = [ln for ln in r.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
            d = json.loads(line[-1]) if line else {'error': r.stderr.decode(errors='replace')[-400:]}
        except Exception as e:               # a labelled extra: never fail the bench over it
            d = {'error': repr(e)}
        d.update({'arithmetic': what, 'library': name})
        out['variants'].append(d)
    base = out['variants'][0]
    for v in out['variants'][1:]:
        if 'ms_per_call_sustained' in v and 'ms_per_call_sustained' in base:
            v['time_vs_shipped'] = v['ms_per_call_sustained'] / base['ms_per_call_sustained']
            if 'J_per_call' in v and 'J_per_call' in base:
                v['energy_vs_shipped'] = v['J_per_call'] / base['J_per_call']
    return out


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    """`python bench.py --gpus N` without a rendezvous in the environment: run N ranks of this script
    on this node under torch.distributed.run and pass their output (rank 0's JSON line) through."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def gpu_clocks(index):
    """Shader / memory clock of GPU `index` as rocm-smi reports them right now (MHz strings), or None.  Recorded
    before and after the timed region instead of lengthening the warm-up the command asked for."""
    try:
        out = subprocess.run(['rocm-smi', '-d', str(index), '--showclocks', '--json'], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, timeout=20).stdout.decode(errors='replace')
        d = json.loads(out[out.index('{'):])
        card = next(iter(d.values()))
        pick = lambda key: next((str(v) for k, v in card.items() if key in k.lower()), None)   # noqa: E731
        return {'sclk': pick('sclk'), 'mclk': pick('mclk')}
    except Exception:                    # measurement garnish only: never fail the bench over it
        return None


def clock_hz_of(clocks):
    """Shader clock in Hz from a gpu_clocks() record ('(2100Mhz)' style strings), or None."""
    import re
    try:
        m = re.search(r'(\d+(?:\.\d+)?)\s*[Mm][Hh]z', str(clocks['sclk']))
        return float(m.group(1)) * 1e6 if m else None
    except (TypeError, KeyError):
        return None


def settle_clocks(step, cap_s=0.3, tol=0.005, batch=10):
    """Untimed calls until the GPU has left its idle clocks: batches of `batch` calls, each closed by a synchronize, until
    the means of two consecutive batches agree within `tol` (0.5 %), at most `cap_s` seconds.  From an idle GPU the first
    ten calls take 1.3-1.6 ms, calls 11-20 1.21 ms, calls 21-50 1.13-1.16 ms, then 1.155 ms steady
    (profiles/r05_clock_ramp.txt): a 5 + 20-step protocol started cold times that DVFS transient, 6 % below the kernels'
    own rate on every box.  This is clock warm-up, not skipped work -- every timed step still runs the whole call --
    and the line says how many calls it took (`ramp_steps`) and why it stopped (`ramp_reason`)."""
    t_start = time.perf_counter()
    prev, n = None, 0
    while True:
        t0 = time.perf_counter()
        for i in range(batch):
            step(200000 + n + i)
        torch.cuda.synchronize()
        now = time.perf_counter()
        n += batch
        mean = (now - t0) / batch
        if prev is not None and abs(mean - prev) <= tol * prev:
            return n, 'settled: two consecutive {}-call means within {:.1f} % ({:.4f} / {:.4f} ms)'.format(batch, tol * 100, prev * 1e3, mean * 1e3)
        if now - t_start >= cap_s:
            return n, 'cap of {:.1f} s reached (last {}-call means {:.4f} / {:.4f} ms)'.format(cap_s, batch, (prev or mean) * 1e3, mean * 1e3)
        prev = mean


def measure(eng, mel, steps, warmup, rank, world, local, dev, events_every, ramp=-1):
    """Clock settle (or `ramp` fixed extra steps) + W untimed steps, then K timed steps between barrier + synchronize
    fences; MAX over ranks."""
    def step(i):
        # asynchronous calls: the fp16 range guard accumulates in the workspace and is read ONCE behind the timed
        # region (eng.check_range() in main: raises if any of these calls overflowed); a per-call read-back would put a
        # host round trip of ~80 us between the calls
        return eng.iaf_generate(mel, None, seed=1000 * rank + i, want=('wav',), check_range=False)['wav']

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(**BARRIER_KW)
            torch.cuda.synchronize(dev)

    # --ramp-steps: -1 (default) = settle_clocks() above; N >= 0 = exactly N extra untimed steps (0: the bare protocol)
    if ramp < 0:
        ramp_n, ramp_why = settle_clocks(step)
    else:
        for i in range(ramp):
            step(100000 + i)
        ramp_n, ramp_why = ramp, 'fixed by --ramp-steps'
    for i in range(warmup):
        step(i)
    fence()
    eng.profile_begin()
    t0 = time.perf_counter()
    wav = None
    for i in range(steps):
        # the event pairs around the layer kernels cost the stream a bubble each: sample them
        # (not in the first step of a period: step 0 starts on an idle GPU that catches up with the host's launches)
        n_ev = max(1, min(events_every, steps))
        eng.profile_pause(i % n_ev != n_ev // 2)
        wav = step(warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    layer_ms, layer_launches = eng.profile_end()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=RED_DEV or dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, layer_ms, layer_launches, wav, (ramp_n, ramp_why)


def three_fracs(alg_flop, moved_bytes, seconds, mfma_per_product=3, peak_tf=PEAK_F16_MFMA_TFLOPS):
    """The three fractions EVERY roofline block of this line carries, every round, so that blocks and rounds compare:
    frac_mfma_alg   useful (algorithmic) FLOP/s over the dense MFMA peak of the arithmetic's instruction,
    frac_mfma_exec  executed MFMA FLOP/s (x3 for the split-fp16 contraction: three fp16 MFMAs per product) over it,
    frac_hbm_moved  bytes the launch(es) must move in THIS design (not SURVEY 8(d)'s per-layer model) over 8 TB/s."""
    alg_tf = alg_flop / seconds / 1e12
    out = {'frac_mfma_alg': alg_tf / peak_tf, 'frac_mfma_exec': mfma_per_product * alg_tf / peak_tf,
           'frac_hbm_moved': moved_bytes / seconds / 1e9 / PEAK_HBM_GBPS,
           'algorithmic_TFLOPs': alg_tf, 'executed_TFLOPs': mfma_per_product * alg_tf, 'moved_GBps': moved_bytes / seconds / 1e9}
    if peak_tf == PEAK_F16_MFMA_TFLOPS:
        out['frac_mfma_exec_of_sustained'] = mfma_per_product * alg_tf / SUSTAINED_F16_MFMA_TFLOPS   # of the bare-MFMA-loop rate at the power-limited clock
    return out


def part_rooflines(eng, hp, B, F, T, part_us, pm):
    """`roofline_cond` / `roofline_deconv`: the conditioning GEMM and the upsampler from the in-process part timing
    (HIP events at the part boundaries, wn_profile_parts_*), with traffic / mfma_util of the committed PMC pass when it was
    taken on these kernel sources."""
    out = {}
    f32 = eng.precision.startswith('f32')
    npp, peak = (1, PEAK_F32_MFMA_TFLOPS) if f32 else (3, PEAK_F16_MFMA_TFLOPS)
    kern = pm.get('kernels') or {}

    def pmc_of(names):
        tr = [kern[n]['hbm_bytes_per_launch'] for n in names if n in kern and kern[n].get('hbm_bytes_per_launch') is not None]
        mu = {n: kern[n].get('mfma_util') for n in names if n in kern and kern[n].get('mfma_util') is not None}
        return (sum(tr) if tr else None), (mu or None)

    Cd, W = hp.deconv_width, hp.width
    rows = W * (sum(hp.num_iaf_layers) + len(hp.num_iaf_layers))        # one 64-row block per layer and per flow head
    stacks = 1 if getattr(hp, 'use_share_deconv', False) else len(hp.num_iaf_layers)
    if part_us.get('cond_gemm'):
        sec = part_us['cond_gemm'] * 1e-6
        flop = 2.0 * rows * Cd * B * T
        moved = 4.0 * rows * B * T + stacks * 4.0 * Cd * B * T + 4.0 * rows * Cd        # C written, enc read once per stack, weights
        cond_k = 'gemm_f32_kernel<cond>' if f32 else 'iaf_cond_h_kernel'
        tr, mu = pmc_of([cond_k])
        r = {'kernel': ('gemm_f32_kernel<8, false> (wn_iaf_f.hip)' if f32 else 'iaf_cond_h_kernel (wn_iaf_c.hip)') + ': C[{} x T] = Wcond[{} x {}] . enc, all layers and heads of a deconv stack in '
                       'one GEMM, written in the accumulator layout of the layer kernels'.format(rows, rows, Cd),
             'bound': 'mfma', 'achieved': flop / sec / 1e12, 'peak': peak, 'unit': 'TFLOP/s', 'frac': flop / sec / 1e12 / peak,
             'us_per_call': part_us['cond_gemm'], 'launches_per_call': stacks, 'flop_per_call': flop,
             'algorithmic_bytes_per_call': moved, 'traffic': tr, 'mfma_util': mu and mu.get(cond_k),
             'pmc_file': pm.get('file'),
             'note': 'co-bound: its output alone (4 B x {} rows per sample) is {:.2f} GB per call, {:.0f} us at the ~5 TB/s '
                     'this part sustains for writes'.format(rows, 4.0 * rows * B * T / 1e9, 4.0 * rows * B * T / 5e12 * 1e6)}
        r.update(three_fracs(flop, moved, sec, npp, peak))
        out['roofline_cond'] = r
    if part_us.get('upsampler'):
        sec = part_us['upsampler'] * 1e-6
        flop, moved, L, cin = 0.0, 0.0, F, 80
        nl = len(hp.deconv_config)
        for j, (fl, st) in enumerate(hp.deconv_config):
            Lout = L * st
            taps = fl // st if not getattr(hp, 'use_resize_conv', False) else (fl - 1 + st - 1) // st + 1
            flop += 2.0 * Cd * taps * cin * B * Lout
            # input read once, weights once, output written once; intermediate layers still go through a phase-major fp32
            # buffer (write + read), the last one writes its G4 words from the GEMM (deconv_pg_kernel, round 5)
            moved += 4.0 * cin * B * L + 4.0 * Cd * taps * st * cin + 4.0 * Cd * B * Lout + (2 * 4.0 * Cd * B * Lout if j + 1 < nl else 0.0)
            L, cin = Lout, Cd
        flop *= stacks
        moved *= stacks
        names = ['gemm_f32_kernel<deconv>', 'deconv_mfma_kernel', 'deconv_interleave_kernel', 'mel_to_cm_kernel'] if f32 else \
            ['deconv_mfma_h_kernel<false>', 'deconv_pg_kernel', 'deconv_mfma_hs_kernel<4>', 'deconv_interleave_g4_kernel', 'mel_to_split_kernel']
        tr, mu = pmc_of(names)
        r = {'kernel': 'upsampler (wn_deconv.hip): {} transposed-conv layers as per-phase split-fp16 GEMMs (deconv_mfma_h[s]_kernel) + '
                       'phase interleave / fp16 split (deconv_interleave_g4_kernel)'.format(nl),
             'bound': 'mfma', 'achieved': flop / sec / 1e12, 'peak': peak, 'unit': 'TFLOP/s', 'frac': flop / sec / 1e12 / peak,
             'us_per_call': part_us['upsampler'], 'launches_per_call': stacks * 2 * nl, 'flop_per_call': flop,
             'algorithmic_bytes_per_call': moved, 'traffic': tr, 'mfma_util': mu, 'pmc_file': pm.get('file')}
        r.update(three_fracs(flop, moved, sec, npp, peak))
        out['roofline_deconv'] = r
    return out


class HwmonSampler:
    """Package power (W) and shader clock (MHz) of one GPU from its hwmon node, sampled on a thread beside a run.  The part
    runs this workload AT ITS POWER CAP (1 400 W): what the kernels can reach is set by energy per sample, and the clock the
    firmware grants is an output of the measurement, not a constant (profiles/r05_power_per_part.txt)."""

    def __init__(self, index, period_s=0.005):
        import glob
        import threading
        self.period = period_s
        self.p, self.f = [], []
        self.dir = None
        try:
            pr = torch.cuda.get_device_properties(index)
            bdf = '{:04x}:{:02x}:{:02x}.0'.format(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            hw = glob.glob('/sys/bus/pci/devices/{}/hwmon/hwmon*'.format(bdf))
            self.dir = hw[0] if hw else None
        except Exception:                    # measurement garnish only
            self.dir = None
        self._on = False
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return int(f.read().strip())

    def _run(self):
        while self._on:
            try:
                self.p.append(self._read('power1_input') * 1e-6)
                self.f.append(self._read('freq1_input') * 1e-6)
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def __enter__(self):
        if self.dir:
            self._on = True
            self._th.start()
        return self

    def __exit__(self, *exc):
        if self._on:
            self._on = False
            self._th.join()

    def summary(self):
        if not self.p:
            return None
        cap = None
        try:
            cap = self._read('power1_cap') * 1e-6
        except (OSError, ValueError):
            pass
        return {'avg_W': float(np.mean(self.p)), 'max_W': float(np.max(self.p)), 'cap_W': cap, 'avg_sclk_MHz': float(np.mean(self.f)),
                'samples': len(self.p)}


def measure_power(eng, mel, rank, local, seconds=1.5):
    """Sustained run of the same step with the package power and the shader clock sampled beside it: J per step, and how
    much of the power cap the workload takes.  Behind the timed region; asynchronous calls like the timed ones."""
    def burst(n):
        for i in range(n):
            eng.iaf_generate(mel, None, seed=9000 + 1000 * rank + i, want=('wav',), check_range=False)
        torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:       # reach the steady operating point first
        burst(20)
    n = 0
    with HwmonSampler(local) as hs:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            burst(20)
            n += 20
        dt = time.perf_counter() - t0
    sm = hs.summary()
    if not sm:
        return None
    sm.update({'ms_per_step_sustained': dt / n * 1e3, 'J_per_step': sm['avg_W'] * dt / n, 'steps': n,
               'frac_of_power_cap': sm['avg_W'] / sm['cap_W'] if sm['cap_W'] else None,
               'note': 'hwmon power1_input / freq1_input of this GPU sampled every 5 ms beside a sustained run of the step; '
                       'at the cap the step time is energy per step over (cap - static power): profiles/r05_power_per_part.txt'})
    return sm


def measure_parts(eng, mel, rank, calls=10):
    """Per-part microseconds of a generate call, measured in THIS process: HIP events at the part boundaries of `calls`
    calls behind the timed region (each event costs the stream a few microseconds: the parts sum to slightly more than
    the unprofiled call)."""
    for i in range(2):
        eng.iaf_generate(mel, None, seed=4000 + i, want=('wav',), check_range=False)
    torch.cuda.synchronize()
    eng.profile_parts_begin()
    for i in range(calls):
        eng.iaf_generate(mel, None, seed=5000 + 1000 * rank + i, want=('wav',), check_range=False)
    ms, n = eng.profile_parts_end()
    return {k: v * 1e3 / max(n, 1) for k, v in ms.items()}, n


def roofline_of(eng, B, F, T, layer_ms, layer_launches, clock_hz=None):
    """Roofline record of the dominant kernel (the single-layer launches bracketed by HIP events inside the
    library, on the stream they are launched on)."""
    avg_layer_s = layer_ms * 1e-3 / max(layer_launches, 1)
    flops_per_launch = LAYER_FLOP_PER_SAMPLE * B * T
    bytes_per_launch = LAYER_BYTES_PER_SAMPLE * B * T
    achieved_tf = flops_per_launch / avg_layer_s / 1e12
    achieved_gbps = bytes_per_launch / avg_layer_s / 1e9
    hoisted = eng.iaf_cond_hoisted(B, F)
    kernel_key = None
    f32 = eng.precision.startswith('f32')
    if f32 and hoisted:
        # fp32 form, round 6: the conditioning 1x1s in one fp32 GEMM per deconv stack (gemm_f32_kernel, `roofline_cond`),
        # the per-layer launch keeps the dilated conv, the gate and the residual 1x1 -- K = 192 + 32 instead of 448 + 32
        kernel_key = 'iaf_layer_kernel<hoist>'
        flops_per_launch = (LAYER_FLOP_PER_SAMPLE - 2 * 16384) * B * T
        bytes_per_launch = LAYER_BYTES_PER_SAMPLE_HOISTED * B * T
        achieved_tf = flops_per_launch / avg_layer_s / 1e12
        achieved_gbps = bytes_per_launch / avg_layer_s / 1e9
        roof = {'kernel': 'iaf_layer_kernel<HOIST> (wn_iaf.hip: dilated conv + gate + residual 1x1 on the hoisted fp32 conditioning term, fp32 MFMA)',
                'bound': 'mfma', 'achieved': achieved_tf, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved_tf / PEAK_F32_MFMA_TFLOPS,
                'hbm_view': {'algorithmic_GBps': achieved_gbps, 'peak_GBps': PEAK_HBM_GBPS},
                'quantisation_note': 'one utterance = 4 800 blocks of 16 samples on 1 024 SIMDs = 4.69 per SIMD: a launch runs '
                                     '5 rounds, 0.94 of the peak before any other loss'}
    elif hoisted and eng.iaf_layer_groups(B, F):
        # Layer groups: one launch = GROUP_LAYERS residual layers of every sample, the residual stream in LDS; the event
        # pairs bracket every group launch of the call.  What binds the launch is the issue of a CU's SIMDs -- the matrix
        # pipe and, beside it, the VALU -- not HBM: `bound` = "mfma", `achieved` = the fp16 MFMA rate the launch executes
        # (three MFMAs per product of the split-fp16 contraction, halo recompute not counted) against the 2.5 PFLOP/s
        # dense peak, i.e. its matrix-pipe utilisation; `mfma_util` is the same quantity from the PMC pass
        # (SQ_VALU_MFMA_BUSY_CYCLES, halo included).  `hbm_view`: the bytes such a launch has to move in this design (read l
        # 256 B, GROUP_LAYERS hoisted terms of 256 B, write l 256 B per sample) against the 8 TB/s; SURVEY 8(d)'s
        # layer-granular model (1536 B per sample and layer) is given beside it.
        nl = GROUP_LAYERS
        bytes_per_launch = (256 + 256 * nl + 256) * B * T
        flops_per_launch = (LAYER_FLOP_PER_SAMPLE - 2 * 16384) * nl * B * T      # without the hoisted 1x1s (the GEMM's)
        achieved_gbps = bytes_per_launch / avg_layer_s / 1e9
        executed_tf = 3 * flops_per_launch / avg_layer_s / 1e12
        kernel_key = 'iaf_group_kernel'
        clk = clock_hz or 2.1e9
        mfma_floor_s = (B * T / 16) * nl * BLOCK_LAYER_MFMA_CYCLES / 1024 / clk
        roof = {'kernel': 'iaf_group_kernel (wn_iaf_g.hip: {} residual layers per launch on hoisted conditioning, l resident '
                          'in LDS, causal halo recomputed; natural and decimated groups alternate)'.format(nl),
                'bound': 'mfma', 'achieved': executed_tf / 3, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': executed_tf / 3 / PEAK_F16_MFMA_TFLOPS,
                'achieved_note': 'achieved / frac: ALGORITHMIC FLOP/s of the launch (halo recompute and the x3 of the split-fp16 '
                                 'contraction not counted) over the dense fp16 MFMA peak; the matrix-pipe rate the launch '
                                 'executes is executed_TFLOPs / frac_mfma_exec (round 4 printed that one under `frac`)',
                'bound_note': 'SIMD issue per CU: matrix pipe + the VALU work beside it (gate, fp16 split / join); not HBM',
                'layers_per_launch': nl,
                'hbm_view': {'bytes_per_launch': bytes_per_launch, 'GBps': achieved_gbps, 'frac': achieved_gbps / PEAK_HBM_GBPS},
                'survey_8d_view': {'bytes_per_launch': LAYER_BYTES_PER_SAMPLE * nl * B * T,
                                   'GBps': LAYER_BYTES_PER_SAMPLE * nl * B * T / avg_layer_s / 1e9,
                                   'frac': LAYER_BYTES_PER_SAMPLE * nl * B * T / avg_layer_s / 1e9 / PEAK_HBM_GBPS,
                                   'note': '1536 B per sample and layer (each layer reads l and enc, writes l): the traffic '
                                           'this launch replaces'},
                'pipe_view': {'mfma_floor_us': mfma_floor_s * 1e6, 'frac_of_mfma_floor': mfma_floor_s / avg_layer_s,
                              'clock_GHz': clk / 1e9,
                              'note': 'per 16-sample block and layer a SIMD needs 84 MFMAs = 1344 cycles of the matrix '
                                      'pipe; floor = that on 1024 SIMDs at the recorded shader clock, without halo. '
                                      'VALU work issues beside the MFMAs on gfx950 (profiles/r04_issue_overlap_ubench.txt), '
                                      'so this -- not MFMA + VALU -- is the floor'}}
    elif hoisted:
        kernel_key = 'iaf_layer_c_kernel'
        flops_per_launch = (LAYER_FLOP_PER_SAMPLE - 2 * 16384) * B * T
        # conditioning 1x1s hoisted into one GEMM per deconv stack (the default): the layer
        # kernel streams l in/out and the projected term, 768 B/sample; the event pairs bracket the
        # single-layer launches only (36 of the 60 layers; the other 24 run two per launch)
        bytes_per_launch = LAYER_BYTES_PER_SAMPLE_HOISTED * B * T
        achieved_gbps = bytes_per_launch / avg_layer_s / 1e9
        roof = {'kernel': 'iaf_layer_c_kernel (dilated conv + gate + residual 1x1 on hoisted conditioning, split-fp16 MFMA)',
                'bound': 'hbm', 'achieved': achieved_gbps, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                'frac': achieved_gbps / PEAK_HBM_GBPS,
                'note': '768 B/sample/layer in this kernel + 256 B/sample/layer written by iaf_cond_h_kernel '
                        '(vs 1536 B/sample/layer of the fused layer kernel)'}
    elif eng.precision.startswith('f16x3'):
        # split-fp16 operands on the fp16 MFMA: 3 MFMAs per product -> the matrix pipe needs
        # 3*61440 fp16-FLOP/sample at a 2.5 PFLOP/s peak (0.07 ns) vs 1536 B/sample at 8 TB/s
        # (0.19 ns): the kernel is HBM-bound
        roof = {'kernel': 'iaf_layer_h_kernel (fused dilated conv + cond 1x1 + gate + residual 1x1, split-fp16 MFMA)',
                'bound': 'hbm', 'achieved': achieved_gbps, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                'frac': achieved_gbps / PEAK_HBM_GBPS,
                'mfma_view': {'executed_fp16_TFLOPs': 3 * achieved_tf, 'peak_TFLOPs': PEAK_F16_MFMA_TFLOPS,
                              'algorithmic_TFLOPs': achieved_tf}}
    else:
        roof = {'kernel': 'iaf_layer_kernel (fused dilated conv + cond 1x1 + gate + residual 1x1, fp32 MFMA)',
                'bound': 'mfma', 'achieved': achieved_tf, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved_tf / PEAK_F32_MFMA_TFLOPS,
                'hbm_view': {'algorithmic_GBps': achieved_gbps, 'peak_GBps': PEAK_HBM_GBPS}}
    pm = pmc_replay(B, F, eng.precision, hoisted, kernel_key)
    roof.update(three_fracs(flops_per_launch, bytes_per_launch, avg_layer_s, 1 if f32 else 3,
                            PEAK_F32_MFMA_TFLOPS if f32 else PEAK_F16_MFMA_TFLOPS))
    roof.update({'traffic': pm['traffic'], 'mfma_util': pm['mfma_util'], 'pmc_file': pm['file'],
                 'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC pass of this command on these kernel sources, '
                                 'profiles/; null when no such pass is committed)',
                 'algorithmic_bytes_per_launch': bytes_per_launch, 'flop_per_launch': flops_per_launch,
                 'avg_launch_us': avg_layer_s * 1e6, 'launches': layer_launches})
    if pm['kernel_us_per_call']:
        roof['kernel_us_per_call_rocprof'] = pm['kernel_us_per_call']     # rocprofv3 kernel trace of this command on these sources (replayed)
    return roof


def stub_main(args, rank, world, local):
    """--stub: the launcher / rendezvous / timing protocol with a torch-CPU step and the gloo backend
    (exercised by tests/test_dist.py on machines without a GPU).  Not a measurement."""
    if world > 1:
        wdist.init_process_group('gloo')
    x = torch.ones(64, 64)
    for _ in range(args.warmup):
        x = torch.tanh(x @ x * 1e-2)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = torch.tanh(x @ x * 1e-2)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    seen = world
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ws = torch.ones(1)
        dist.all_reduce(ws)
        seen = int(ws.item())
    if rank == 0:
        print(json.dumps({'metric': 'stub', 'value': args.steps * world / elapsed, 'unit': 'steps/s', 'n_gpus': world,
                          'world_size_seen': seen, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': elapsed / args.steps * 1e3, 'data': 'stub'}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch-per-gpu', type=int, default=1)
    ap.add_argument('--frames', type=int, default=384, help='mel frames per utterance (384 -> 76800 samples = 4.8 s)')
    ap.add_argument('--config', default=os.path.join(ROOT, 'config_jsons', 'parallel_wavenet.json'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip everything behind the timed region: part timing, power sampling, the AR / teacher figures, the PCIe-inclusive call, the 8-utterance and fp32 blocks')
    ap.add_argument('--layer-events-every', type=int, default=20,
                    help='record the HIP-event pairs around the layer kernels in every n-th timed step')
    ap.add_argument('--precision', default=None, choices=['f16x3', 'f16x3-fused', 'f16x3-hoisted', 'f32', 'f32-fused', 'f32-hoisted'],
                    help='IAF contraction arithmetic (default: f16x3 = split-fp16 on the fp16 MFMA)')
    ap.add_argument('--ar-samples', type=int, default=1600,
                    help='extras: generated samples per utterance of the autoregressive runs (configs[3]; 1600 = 0.1 s)')
    ap.add_argument('--ramp-steps', type=int, default=-1,
                    help='untimed steps before the --warmup steps: -1 (default) = until two consecutive 10-call means agree within '
                         '0.5 %% (at most 0.3 s: the GPU leaves its idle clocks); N >= 0 = exactly N (0 = the bare protocol)')
    ap.add_argument('--stub', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--f16x2-probe', action='store_true', help=argparse.SUPPRESS)   # child process of the roofline_f16x2 extra
    ap.add_argument('--same-gpu', action='store_true', help=argparse.SUPPRESS)   # N ranks on GPU 0 over gloo: code-path check on a one-GPU box
    args = ap.parse_args()

    if args.f16x2_probe:
        return f16x2_probe()
    rank, world, local = wdist.env_rank_world()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
    if args.stub:
        return stub_main(args, rank, world, local)
    from nsynth_wavenet_amd.engine import Engine
    if args.same_gpu:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        wdist.init_process_group('gloo' if args.same_gpu else 'nccl')
    dev = torch.device('cuda', local)
    global BARRIER_KW, RED_DEV
    BARRIER_KW = {} if args.same_gpu else {'device_ids': [local]}
    RED_DEV = torch.device('cpu') if args.same_gpu else None

    with open(args.config) as f:
        hp_dict = json.load(f)
    hp = cfg.load_hparams(hp_dict)
    # random-init weights of the named architecture (no checkpoints offline); rank 0
    # builds them, every other rank receives them in one RCCL broadcast over xGMI
    weights = wts.synthetic_weights(hp, 'student', seed=1234, init='tf') if rank == 0 else None
    weights = wdist.broadcast_weights(weights, hp, 'student', src=0, device=RED_DEV or dev)
    eng = Engine(hp, kind='student', device=dev, precision=args.precision).load_weights(weights)

    B, F = args.batch_per_gpu, args.frames
    T = eng.iaf_length(F)
    mel_host = np.random.RandomState(12345 + rank).uniform(0, 1, [B, F, 80]).astype(np.float32)
    mel = torch.from_numpy(mel_host).to(dev)
    clocks_before = gpu_clocks(local) if rank == 0 else None
    elapsed, layer_ms, layer_launches, wav, ramp_info = measure(eng, mel, args.steps, args.warmup, rank, world, local, dev,
                                                                args.layer_events_every, ramp=args.ramp_steps)
    clocks_after = gpu_clocks(local) if rank == 0 else None
    assert wav.shape == (B, T) and bool(torch.isfinite(wav).all())
    eng.check_range()                     # raises if a split-fp16 operand left the fp16 range during the timed calls
    seen = world
    if world > 1:                         # the world size the collective actually spans, not the environment's word
        ws = torch.ones(1, device=RED_DEV or dev)
        dist.all_reduce(ws)
        seen = int(ws.item())

    rec = None
    if rank == 0:
        total_samples = world * B * T * args.steps
        value = total_samples / elapsed
        roof = roofline_of(eng, B, F, T, layer_ms, layer_launches, clock_hz_of(clocks_after))
        # the whole call against SURVEY.md 8(d)'s layer-granular traffic model (98 484 B per generated sample; its
        # "60 % of the HBM roofline" is 48.7 M samples/s per GPU) -- beside the dominant kernel's own figures
        gbps = PATH_BYTES_PER_SAMPLE * (total_samples / world) / elapsed / 1e9
        roof['path_8d_view'] = {'bytes_per_sample': PATH_BYTES_PER_SAMPLE, 'GBps': gbps, 'frac': gbps / 8000.0,
                                'note': 'whole generate call, SURVEY 8(d) accounting (every layer reads l and enc and '
                                        'writes l once); the shipped launch structure moves fewer bytes than this model'}
        dtype = 'f32' if eng.precision.startswith('f32') else \
            'split-fp16: activations stored as fp16 hi+lo pairs (32 bits per value, 22-bit significand, fp16 exponent ' \
            'range), contractions as 3 fp16 MFMAs per product with fp32 accumulate; conditioning term and outputs fp32'
        rec = {
            'metric': '16 kHz audio samples/sec, parallel-WaveNet (IAF student) generation',
            'value': value,
            'unit': 'samples/s',
            'n_gpus': world,
            'world_size_seen': seen,
            'steps': args.steps,
            'warmup': args.warmup,
            'ramp_steps': ramp_info[0],          # untimed calls before the warm-up steps: clock settle (bench.py: settle_clocks)
            'ramp_reason': ramp_info[1],
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': dtype,
            'data': 'synthetic',
            'clocks': {'before': clocks_before, 'after': clocks_after, 'source': 'rocm-smi --showclocks, GPU of rank 0'},
            'config': {
                'workload': 'BASELINE.json configs[1]: parallel_wavenet.json IAF student generation, '
                            'synthetic 80-dim mel, 16 kHz, {} utterance(s) of {} frames = {} samples per GPU per step'
                            .format(B, F, T),
                'batch_per_gpu': B, 'frames': F, 'samples_per_utterance': T,
                'x_realtime_per_gpu': value / world / 16000.0,
                'samples_per_sec_per_gpu': value / world,
                'weights': 'random-init (TF initialisers), seed 1234; noise drawn on device (Philox)',
                'precision': eng.precision,
                'parallelism': 'utterance-sharded x{} (no data-path collective)'.format(world),
                'path_gflop_per_step_per_gpu': PATH_FLOP_PER_SAMPLE * B * T / 1e9,
                'path_achieved_tflops': PATH_FLOP_PER_SAMPLE * (total_samples / world) / elapsed / 1e12,
                'path_algorithmic_GBps': PATH_BYTES_PER_SAMPLE * (total_samples / world) / elapsed / 1e9,
            },
            'roofline': roof,
        }
    if rank == 0 and world == 1 and not args.no_extras:
        # Where the call's time goes, measured in THIS process (HIP events at the part boundaries of ten more calls) --
        # not replayed from a profile -- and the roofline blocks of the two parts that are not the dominant kernel.
        # (the sustained run first: the parts are then timed at the operating point the firmware settles on, not in the
        # clock transient behind a 25-step run from an idle GPU)
        pw = measure_power(eng, mel, rank, local)
        eng.check_range()
        if pw:
            pw['uJ_per_sample'] = pw['J_per_step'] / (B * T) * 1e6
            rec['power'] = pw
        part_us, n_parts = measure_parts(eng, mel, rank, calls=20)
        eng.check_range()
        rec['kernel_us_per_call'] = dict(part_us, calls=n_parts, sum_us=sum(part_us.values()),
                                         source='HIP events at the part boundaries inside the library (wn_profile_parts_*), this '
                                                'process, behind the timed region and the sustained run; every event costs the '
                                                'stream a few microseconds: the parts sum to slightly more than a call')
        pm_all = pmc_replay(B, F, eng.precision, eng.iaf_cond_hoisted(B, F), rec['roofline'].get('kernel', '').split(' ')[0])
        rec.update(part_rooflines(eng, hp, B, F, T, part_us, pm_all))
    if world == 1 and not args.no_extras:
        # (0) the other BASELINE.json configs that fit one GPU, driver-timed in the same line: configs[3] (wavenet_mol.json
        #     autoregressive fastgen, one utterance) and its batched form, and the teacher's full-sequence forward
        rec.update(teacher_extras(dev, args.ar_samples))
        # (1) PCIe-inclusive call: pageable numpy mel in -> H2D -> generate -> D2H -> numpy wav out, the
        #     path of parallelgen.synthesis.  Reported beside the resident figure, never as `value`.
        n_e2e = max(3, min(args.steps, 20))
        for i in range(2):
            eng.iaf_generate(mel_host, None, seed=i, want=('wav',))['wav'].cpu().numpy()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(n_e2e):
            out = eng.iaf_generate(mel_host, None, seed=7000 + i, want=('wav',))['wav'].cpu().numpy()
        rec['e2e_ms_per_step'] = (time.perf_counter() - t0) / n_e2e * 1e3
        rec['e2e_note'] = 'host numpy mel -> H2D -> generate -> D2H -> host numpy wav ({} calls); `value` is the ' \
                          'HBM-resident rate'.format(n_e2e)
        assert out.shape == (B, T)
        # (1a) the unspent error budget, priced (a labelled extra: narrower than the reference, never the metric)
        rec['roofline_f16x2'] = roofline_f16x2()
        # (1b) the drop-in CLI end to end (files in, files out)
        try:
            rec['cli_e2e'] = cli_e2e(hp_dict, weights)
        except Exception as e:               # a figure beside the metric: never fail the bench over the host's file system
            rec['cli_e2e'] = {'error': repr(e)}
        # (2) the same dominant kernel with the GPU filled: 8 utterances per GPU (BASELINE configs[2]'s per-GPU share)
        if B != 8:
            mel8 = torch.from_numpy(np.random.RandomState(777).uniform(0, 1, [8, F, 80]).astype(np.float32)).to(dev)
            n8 = max(3, min(args.steps, 20))
            el8, lms8, ll8, wav8, _ = measure(eng, mel8, n8, 2, rank, world, local, dev, 2)
            eng.check_range()
            r8 = roofline_of(eng, 8, F, T, lms8, ll8, clock_hz_of(clocks_after))
            r8.update({'batch_per_gpu': 8, 'steps': n8, 'ms_per_step': el8 / n8 * 1e3,
                       'samples_per_sec': 8 * T * n8 / el8})
            rec['roofline_b8'] = r8
        # (3) the same workload in the reference's own arithmetic: fp32 MFMA (v_mfma_f32_16x16x4_f32) instead of the
        #     split-fp16 contraction -- driver-timed beside the headline figure
        if not eng.precision.startswith('f32'):
            eng32 = Engine(hp, kind='student', device=dev, precision='f32').load_weights(weights)
            n32 = max(3, min(args.steps, 20))
            el32, lms32, ll32, wav32, _ = measure(eng32, mel, n32, 2, rank, world, local, dev, 2)
            assert bool(torch.isfinite(wav32).all())
            r32 = roofline_of(eng32, B, F, T, lms32, ll32, clock_hz_of(clocks_after))
            path_tf = PATH_FLOP_PER_SAMPLE * B * T * n32 / el32 / 1e12
            r32.update({'precision': 'f32', 'steps': n32, 'ms_per_step': el32 / n32 * 1e3, 'samples_per_sec': B * T * n32 / el32,
                        'path_achieved_tflops': path_tf, 'path_frac_of_f32_mfma_peak': path_tf / PEAK_F32_MFMA_TFLOPS,
                        'form': 'hoisted: conditioning in one fp32 GEMM per deconv stack (gemm_f32_kernel), the upsampler\'s last layer as a '
                                'frame-axis fp32 GEMM, per-layer launches on Q4 rows, the flow head in the last layer\'s epilogue (round 6)'
                                if eng32.iaf_cond_hoisted(B, F) else 'fused: one kernel per layer reads enc itself'})
            pu32, n_p32 = measure_parts(eng32, mel, rank, calls=10)
            r32['kernel_us_per_call'] = dict(pu32, calls=n_p32, sum_us=sum(pu32.values()))
            pm32 = pmc_replay(B, F, eng32.precision, eng32.iaf_cond_hoisted(B, F), 'iaf_layer_kernel<hoist>')
            for k_, v_ in part_rooflines(eng32, hp, B, F, T, pu32, pm32).items():
                r32[k_] = v_
            pw32 = measure_power(eng32, mel, rank, local, seconds=1.0)
            if pw32:
                r32['power'] = {k_: pw32[k_] for k_ in ('avg_W', 'max_W', 'cap_W', 'avg_sclk_MHz', 'ms_per_step_sustained', 'J_per_step')}
            # ... and at eight utterances per GPU (BASELINE configs[2]'s share): the per-launch start-up of the 60 layer launches
            # amortises, the 4.69 -> 5 tile rounds of one utterance become 37.5 -> 38
            try:
                mel8f = torch.from_numpy(np.random.RandomState(778).uniform(0, 1, [8, F, 80]).astype(np.float32)).to(dev)
                el8f, _, _, w8f, _ = measure(eng32, mel8f, 5, 2, rank, world, local, dev, 1 << 30)
                tf8 = PATH_FLOP_PER_SAMPLE * 8 * T * 5 / el8f / 1e12
                r32['b8'] = {'batch_per_gpu': 8, 'steps': 5, 'ms_per_step': el8f / 5 * 1e3, 'samples_per_sec': 8 * T * 5 / el8f,
                             'path_achieved_tflops': tf8, 'path_frac_of_f32_mfma_peak': tf8 / PEAK_F32_MFMA_TFLOPS}
                assert bool(torch.isfinite(w8f).all())
            except Exception as e:           # a labelled extra
                r32['b8'] = {'error': repr(e)}
            rec['roofline_f32'] = r32
            eng32.close()
    if world > 1 and not args.no_extras:
        # The per-GPU shares of BASELINE configs[2] (64 utterances over 8 GPUs = 8 per GPU, this model) and configs[4]
        # (parallel_wavenet_gauss.json as shipped, 128 over 8 = 16 per GPU): timed on every rank, MAX over ranks,
        # aggregate = all ranks' samples over that time.  Weak scaling like the headline figure.
        def share(engine, nb, seed, steps):
            melb = torch.from_numpy(np.random.RandomState(seed + rank).uniform(0, 1, [nb, F, 80]).astype(np.float32)).to(dev)
            el, _, _, w, _ = measure(engine, melb, steps, 2, rank, world, local, dev, 1 << 30)
            engine.check_range()          # the timed calls are asynchronous: one question behind them (raises on overflow)
            assert w.shape == (nb, T) and bool(torch.isfinite(w).all())
            return {'batch_per_gpu': nb, 'utterances': nb * world, 'steps': steps, 'ms_per_step': el / steps * 1e3,
                    'samples_per_sec': world * nb * T * steps / el, 'x_realtime': world * nb * T * steps / el / 16000.0}
        n_s = max(3, min(args.steps, 10))
        c2 = share(eng, 8, 777, n_s)
        with open(os.path.join(ROOT, 'config_jsons', 'parallel_wavenet_gauss.json')) as f:
            hp4 = cfg.load_hparams(json.load(f))
        w4 = wts.synthetic_weights(hp4, 'student', seed=1234, init='tf') if rank == 0 else None
        w4 = wdist.broadcast_weights(w4, hp4, 'student', src=0, device=RED_DEV or dev)
        eng4 = Engine(hp4, kind='student', device=dev, precision=args.precision).load_weights(w4)
        c4 = share(eng4, 16, 888, n_s)
        eng4.close()
        if rank == 0:
            c2['workload'] = 'BASELINE.json configs[2] share: parallel_wavenet.json, 8 utterances of {} frames per GPU'.format(F)
            c4['workload'] = 'BASELINE.json configs[4] share: parallel_wavenet_gauss.json as shipped (private deconv ' \
                             'stacks), 16 utterances of {} frames per GPU'.format(F)
            rec['config2_share'] = c2
            rec['config4_share'] = c4
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            rec['cpu_baseline'] = cpu_baseline(hp_dict, F)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier(**BARRIER_KW)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
