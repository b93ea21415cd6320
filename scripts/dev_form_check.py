"""dev: repeated calls of one form with CHANGING inputs (same or different shapes) against the default."""
import json, os, sys
os.environ.setdefault('WN_UNVERIFIED_FORMS', '1')     # the hoisted-resident form is withheld
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd import config as cfg, weights as wts
from nsynth_wavenet_amd.engine import Engine
form = sys.argv[1]
d = json.load(open('config_jsons/parallel_wavenet.json'))
hp = cfg.load_hparams(d)
w = wts.synthetic_weights(hp, seed=7, init='unit')
a, b = Engine(d, precision='f16x3').load_weights(w), Engine(d, precision=form).load_weights(w)
for arg in sys.argv[2:]:
    B, F = (int(v) for v in arg.split('x'))
    mel = torch.rand(B, F, 80, device='cuda')
    ra = a.iaf_generate(mel, None, seed=int(torch.randint(1 << 30, (1,))), want=('x', 'rand_input'))
    rb = b.iaf_generate(mel, ra['rand_input'], want=('x',))
    diff = (ra['x'] - rb['x']).abs()
    bad = (diff > 2e-5 * max(1.0, float(ra['x'].abs().max()))).nonzero()
    print('B=%d F=%d T=%d maxdiff %.3e nbad %d first %s' % (B, F, ra['x'].shape[1], float(diff.max()), len(bad), bad[0].tolist() if len(bad) else None), flush=True)
