"""dev: the fuzz tool's marginal cases (unit-gain weights) -- are the forms deterministic, and how far apart relative to |x|?"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nsynth_wavenet_amd import weights as wts, config as cfg
from nsynth_wavenet_amd.engine import Engine
d = json.load(open(os.path.join(ROOT, 'config_jsons', 'parallel_wavenet.json')))
hp = cfg.load_hparams(d)
wseed = int(sys.argv[1]) if len(sys.argv) > 1 else 402157
w = wts.synthetic_weights(hp, seed=wseed, init='unit')
FORMS = ('f16x3', 'f16x3-fused', 'f16x3-hoisted', 'f32', 'f32-fused')
engs = {p: Engine(d, precision=p).load_weights(w) for p in FORMS}
rs = np.random.RandomState(5)
for (B, F) in ((12, 54), (6, 122), (10, 394), (5, 448), (6, 181), (3, 226)):
    mel = torch.rand(B, F, 80, device='cuda')
    a = engs['f16x3'].iaf_generate(mel, None, seed=B * 1000 + F, want=('x', 'rand_input'))
    scale = max(1.0, float(a['x'].abs().max()))
    xs = {}
    for p in FORMS:
        runs = [engs[p].iaf_generate(mel, a['rand_input'], want=('x',))['x'].clone() for _ in range(3)]
        det = all(torch.equal(runs[0], r) for r in runs[1:])
        xs[p] = runs[0].double()
        if not det:
            print('  NOT deterministic:', p, [float((runs[0] - r).abs().max()) for r in runs[1:]])
    ref = xs['f32']
    print('B=%d F=%d scale %.1f  | rel. to hoisted fp32: ' % (B, F, scale) +
          '  '.join('%s %.2e' % (p, float((xs[p] - ref).abs().max()) / scale) for p in FORMS if p != 'f32'), flush=True)
