"""Run-to-run reproducibility of every IAF execution form (dev tool): the kernels have no
atomics and a fixed summation order, so repeated calls must be bitwise identical."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nsynth_wavenet_amd import weights as wts, config as cfg
from nsynth_wavenet_amd.engine import Engine
d = json.load(open(os.path.join(ROOT, 'config_jsons', 'parallel_wavenet.json')))
hp = cfg.load_hparams(d)
w = wts.synthetic_weights(hp, seed=1234)
bad = 0
for prec in ('f16x3-fused', 'f16x3-hoisted', 'f32'):
    eng = Engine(d, precision=prec).load_weights(w)
    for (B, F) in ((1, 384), (4, 200), (9, 61)):
        mel = torch.rand(B, F, 80, device='cuda')
        ref = eng.iaf_generate(mel, None, seed=7, want=('x',))['x'].clone()
        n = 0
        for it in range(25):
            # interleave another shape so that the workspace is re-used with different contents
            eng.iaf_generate(torch.rand(2, 30 + it, 80, device='cuda'), None, seed=it)
            x = eng.iaf_generate(mel, None, seed=7, want=('x',))['x']
            n += int(not torch.equal(x, ref))
        bad += n
        print(prec, B, F, 'mismatching repeats:', n, 'finite', bool(torch.isfinite(ref).all()))
    eng.close()
print('TOTAL mismatches', bad)
sys.exit(1 if bad else 0)
