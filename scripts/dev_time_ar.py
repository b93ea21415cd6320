"""Time the full-size AR path (dev tool)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nsynth_wavenet_amd import weights as wts, config as cfg
from nsynth_wavenet_amd.engine import Engine
d = json.load(open(os.path.join(ROOT, 'config_jsons', 'wavenet_mol.json')))
hp = cfg.load_hparams(d)
eng = Engine(d).load_weights(wts.synthetic_weights(hp, seed=1))
Tn = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
for B in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1,8').split(',')]:
    enc = torch.randn(B, Tn, 256, device='cuda') * 0.1
    eng.ar_generate(enc, None, seed=1)
    torch.cuda.synchronize()
    t = time.time()
    eng.ar_generate(enc, None, seed=2)
    torch.cuda.synchronize()
    dt = time.time() - t
    print('AR full-size B=%d: %.1f us/step, %.0f samples/s total (%.2fx RT per utterance)' % (B, dt / Tn * 1e6, B * Tn / dt, Tn / dt / 16000))
