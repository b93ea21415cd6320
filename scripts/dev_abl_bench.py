"""dev: time one library build (WN_LIB_PATH) on configs[1] without caring whether its results are right -- for
ablation builds (-DGK_ABL=...) whose outputs are wrong by construction.  Prints ms per call and the average duration
of the bracketed residual-stack launches (HIP events inside the library).
    WN_LIB_PATH=vlibs/lib_<tag>.so python scripts/dev_abl_bench.py [--batch B] [--steps K]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd.engine import Engine      # noqa: E402
from oracle import wavenet_np as O                # noqa: E402  (synthetic weights only)

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--frames', type=int, default=384)
ap.add_argument('--steps', type=int, default=100)
ap.add_argument('--tag', default=os.environ.get('WN_LIB_PATH', 'default'))
a = ap.parse_args()
cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
w = O.synth_weights(O.HP(cfgd), 'student', seed=1234, init='tf')
eng = Engine(cfgd).load_weights(w)
mel = torch.from_numpy(np.random.RandomState(12345).uniform(0, 1, [a.batch, a.frames, 80]).astype(np.float32)).cuda()
for i in range(10):
    eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
torch.cuda.synchronize()
eng.profile_begin()
t0 = time.perf_counter()
for i in range(a.steps):
    eng.profile_pause(i % 10 != 5)
    eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, n = eng.profile_end()
parts = ''
if hasattr(eng, 'profile_parts_begin') and hasattr(eng.lib, 'wn_profile_parts_begin'):
    eng.profile_parts_begin()
    for i in range(10):
        eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
    pms, pn = eng.profile_parts_end()
    parts = '   parts us/call: ' + ' '.join('{} {:.1f}'.format(k[:5], v * 1e3 / max(pn, 1)) for k, v in pms.items())
# a digest of one call on fixed inputs: equal digests = bit-identical results (arithmetic-preserving variants)
import hashlib
x = eng.iaf_generate(mel, None, seed=4242, want=('x',), check_range=False)['x']
dig = hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:10]
print('{:24s} B={} {:8.4f} ms/call   residual-stack launches: {:7.2f} us avg over {}{}   x digest {} finite {}'.format(
    os.path.basename(a.tag), a.batch, dt / a.steps * 1e3, ms * 1e3 / max(n, 1), n, parts, dig, bool(torch.isfinite(x).all())), flush=True)
