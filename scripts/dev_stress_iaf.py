"""Stress shapes for the IAF path: f16x3 vs f32 GPU paths + invariants (dev tool)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nsynth_wavenet_amd import weights as wts, config as cfg
from nsynth_wavenet_amd.engine import Engine
for name in ('parallel_wavenet.json', 'parallel_wavenet_gauss.json'):
    d = json.load(open(os.path.join(ROOT, 'config_jsons', name)))
    hp = cfg.load_hparams(d)
    w = wts.synthetic_weights(hp, seed=1234)
    e16 = Engine(d, precision='f16x3').load_weights(w)
    e32 = Engine(d, precision='f32').load_weights(w)
    for (B, F) in ((8, 400), (3, 1000), (16, 384), (1, 773), (5, 37)):
        mel = torch.rand(B, F, 80, device='cuda')
        a = e16.iaf_generate(mel, None, seed=3, want=('x', 'wav', 'rand_input', 'mean_tot', 'scale_tot'))
        b = e32.iaf_generate(mel, a['rand_input'], want=('x',))
        torch.cuda.synchronize()
        x = a['x'].double(); r = a['rand_input'].double()
        k2 = (x - (r * a['scale_tot'].double() + a['mean_tot'].double())).abs().max().item()
        t0 = time.time(); e16.iaf_generate(mel, None, seed=4); torch.cuda.synchronize(); dt = time.time() - t0
        print(name[:22], 'B', B, 'F', F, 'T', x.shape[1], 'f16x3 vs f32 maxdiff %.2e' % (a['x'] - b['x']).abs().max().item(),
              'K2 %.1e' % k2, 'finite', bool(torch.isfinite(a['x']).all()), '%.1f Msamples/s' % (B * x.shape[1] / dt / 1e6))
    e16.close(); e32.close()
