"""Measurements of the secondary paths, one JSON line each (the driver contract is bench.py; this file
keeps the AR and teacher-forward numbers quoted in DESIGN.md / BASELINE.md reproducible).

    python bench_aux.py --workload ar      [--batch 1] [--samples 1600]   BASELINE.json configs[3]: fastgen
    python bench_aux.py --workload teacher [--batch 1] [--frames 384]     teacher full-sequence forward

Synthetic conditioning / mel and random-init weights of wavenet_mol.json (width 512, 30 layers, MoL-10).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from nsynth_wavenet_amd import config as cfg              # noqa: E402
from nsynth_wavenet_amd import weights as wts            # noqa: E402
from nsynth_wavenet_amd.engine import Engine             # noqa: E402

PEAK_HBM_GBPS = 8000.0
PEAK_F16_MFMA_TFLOPS = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', required=True, choices=['ar', 'teacher'])
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--samples', type=int, default=1600, help='AR: generated samples per utterance (0.1 s)')
    ap.add_argument('--frames', type=int, default=384, help='teacher: mel frames (384 -> 76800 samples)')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--graph', action='store_true', help='AR: hipGraph replay instead of plain launches')
    ap.add_argument('--streams', type=int, default=1,
                    help='AR: cut the batch into this many utterance groups, each an independent chain on its own stream (one shared handle)')
    ap.add_argument('--threads', action='store_true',
                    help='AR with --streams G: G host threads, each a fork of the engine (own queue state, own stream) issuing PLAIN launches, instead of one thread replaying hipGraphs')
    ap.add_argument('--config', default=os.path.join(ROOT, 'config_jsons', 'wavenet_mol.json'))
    args = ap.parse_args()
    with open(args.config) as f:
        hp = cfg.load_hparams(json.load(f))
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    eng = Engine(hp, kind='teacher', device=dev).load_weights(wts.synthetic_weights(hp, 'teacher', seed=1, init='unit'))
    W, S, Cd = hp.width, hp.skip_width, hp.deconv_width
    G = cfg.teacher_gate_width(hp)
    B = args.batch
    rs = np.random.RandomState(0)
    if args.workload == 'ar':
        Tn = args.samples
        enc = torch.as_tensor((rs.standard_normal([B, Tn, Cd]) * 0.1).astype(np.float32)).to(dev)
        if args.threads and args.streams > 1:
            import threading
            G = args.streams
            parts = [enc[B * g // G: B * (g + 1) // G].contiguous() for g in range(G)]
            forks = [eng.fork() for _ in range(G)]
            streams = [torch.cuda.Stream(dev) for _ in range(G)]
            outs = [None] * G

            def work(g, seed):
                with torch.cuda.stream(streams[g]):
                    outs[g] = forks[g].ar_generate(parts[g], None, seed=seed + g, use_graph=args.graph)

            def run_all(seed):
                ths = [threading.Thread(target=work, args=(g, seed)) for g in range(G)]
                for t in ths:
                    t.start()
                for t in ths:
                    t.join()
                torch.cuda.synchronize(dev)
            run_all(1)
            t0 = time.perf_counter()
            for i in range(args.steps):
                run_all(100 + 10 * i)
            dt = (time.perf_counter() - t0) / args.steps
            out = {'wav': torch.cat([o['wav'] for o in outs], 0)}
        else:
            eng.ar_generate(enc, None, seed=1, use_graph=args.graph, streams=args.streams)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(args.steps):
                out = eng.ar_generate(enc, None, seed=2 + i, use_graph=args.graph, streams=args.streams)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / args.steps
        assert bool(torch.isfinite(out['wav']).all())
        us_step = dt / Tn * 1e6
        # weights streamed once per step (fp32): gate + composite (batch >= 1), res/skip, head
        wbytes = 4.0 * (hp.num_layers * (G * (3 * W + Cd) + (W + S) * (G // 2)) + (hp.num_layers - 1) * G * (G // 2) +
                        S * W + S * (S + Cd) + cfg.teacher_out_width(hp) * S)
        rec = {'metric': '16 kHz audio samples/sec, autoregressive WaveNet (fastgen) generation', 'value': B * Tn / dt,
               'unit': 'samples/s', 'n_gpus': 1, 'steps': args.steps, 'ms_per_step': dt * 1e3, 'higher_is_better': True,
               'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'BASELINE.json configs[3]: wavenet_mol.json fastgen, {} utterance(s) x {} samples, '
                                      'Philox sampling on device'.format(B, Tn),
                          'us_per_sample_step': us_step, 'x_realtime_per_utterance': Tn / dt / 16000.0,
                          'launches_per_step': (hp.num_layers + 4) if B < 4 else (2 * hp.num_layers + 5),
                          'streams': args.streams, 'utterances_per_stream': B / float(args.streams),
                          'host_threads': args.streams if args.threads else 1,
                          'submission': 'plain launches' if (args.threads and not args.graph) or not (args.graph or args.streams > 1) else 'hipGraph replay (16 steps per graph, captured per call)'},
               'roofline': {'bound': 'hbm', 'achieved': wbytes / (us_step * 1e-6) / 1e9, 'peak': PEAK_HBM_GBPS,
                            'unit': 'GB/s', 'frac': wbytes / (us_step * 1e-6) / 1e9 / PEAK_HBM_GBPS, 'traffic': None,
                            'note': 'weight bytes streamed per step / step time; the step is a chain of dependent '
                                    'launches (launch-latency bound, DESIGN.md 3.4)'}}
    else:
        F = args.frames
        T = F * cfg.frame_shift(hp)
        mel = torch.as_tensor(rs.uniform(0, 1, [B, F, 80]).astype(np.float32)).to(dev)
        wav = torch.as_tensor(rs.uniform(-1, 1, [B, T]).astype(np.float32)).to(dev)
        for _ in range(2):
            out = eng.teacher_forward(wav, mel)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = eng.teacher_forward(wav, mel)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / args.steps
        assert bool(torch.isfinite(out).all())
        mac = hp.num_layers * (G * (3 * W + Cd) + (W + S) * (G // 2)) + S * W + S * (S + Cd) + cfg.teacher_out_width(hp) * S
        tf = 3 * 2.0 * mac * B * T / dt / 1e12                     # three fp16 MFMAs per product
        rec = {'metric': '16 kHz audio samples/sec, WaveNet teacher full-sequence forward', 'value': B * T / dt,
               'unit': 'samples/s', 'n_gpus': 1, 'steps': args.steps, 'ms_per_step': dt * 1e3, 'higher_is_better': True,
               'dtype': 'f32 storage; contractions as split-fp16 (3 fp16 MFMAs per product), fp32 accumulate',
               'data': 'synthetic',
               'config': {'workload': 'wavenet_mol.json Wavenet.feed_forward, {} utterance(s) of {} frames = {} samples'
                          .format(B, F, T), 'x_realtime': B * T / dt / 16000.0},
               'roofline': {'bound': 'mfma', 'achieved': tf, 'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': tf / PEAK_F16_MFMA_TFLOPS, 'traffic': None,
                            'note': 'executed fp16 MFMA FLOPs of the layer GEMMs / wall time of the whole call'}}
    print(json.dumps(rec), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
