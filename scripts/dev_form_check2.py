"""dev: an execution form vs the default for several flow structures and lengths; prints the bad blocks."""
import json, os, sys
os.environ.setdefault('WN_UNVERIFIED_FORMS', '1')     # the hoisted-resident form is withheld
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd import config as cfg, weights as wts
from nsynth_wavenet_amd.engine import Engine
form = sys.argv[1]
for nl in ([13], [14], [16], [18], [20]):
    for F in (246,):
        d = dict(json.load(open('config_jsons/parallel_wavenet.json')), num_iaf_layers=nl)
        hp = cfg.load_hparams(d)
        w = wts.synthetic_weights(hp, seed=7, init='unit')
        a, b = Engine(d, precision='f16x3').load_weights(w), Engine(d, precision=form).load_weights(w)
        mel = torch.rand(1, F, 80, device='cuda')
        ra = a.iaf_generate(mel, None, seed=1, want=('x', 'rand_input'))
        rb = b.iaf_generate(mel, ra['rand_input'], want=('x',))
        rb2 = b.iaf_generate(mel, ra['rand_input'], want=('x',))
        print('  repeat identical:', bool(torch.equal(rb['x'], rb2['x'])), 'second-call maxdiff %.3e' % float((ra['x'] - rb2['x']).abs().max()))
        diff = (ra['x'] - rb['x']).abs()[0]
        T = diff.shape[0]
        bad = (diff > 2e-5 * max(1.0, float(ra['x'].abs().max()))).nonzero().flatten().cpu().numpy()
        blocks = np.unique(bad // 16)
        print('layers', nl, 'F', F, 'T', T, 'maxdiff %.3e' % float(diff.max()), 'bad samples', len(bad), 'bad blocks', len(blocks),
              'first blocks', blocks[:8].tolist(), flush=True)
        a.close(); b.close()
