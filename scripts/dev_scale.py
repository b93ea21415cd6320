import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np
from nsynth_wavenet_amd.engine import Engine
from oracle import wavenet_np as O
cfgd = dict(json.load(open('config_jsons/parallel_wavenet.json')), num_iaf_layers=[1])
hp = O.HP(cfgd)
w = O.scale_probe_weights(hp)
F = 384
for ng in ('0', '1'):
    os.environ['WN_NO_GROUPS'] = ng
    eng = Engine(cfgd, precision='f16x3').load_weights(w)
    T = eng.iaf_length(F)
    mel = np.zeros([2, F, 80], np.float32)
    z = np.random.RandomState(94107).standard_normal([2, T]).astype(np.float32)
    z[1] *= 12.0
    out = eng.iaf_generate(mel, z, want=('scale_tot', 'mean_tot', 'x'))
    s = out['scale_tot'].cpu().numpy().astype(np.float64)
    p = np.concatenate([np.zeros([2, 1]), z[:, :-1].astype(np.float64)], axis=1)
    want = O.scale_log_scale(p)[0]
    d = np.abs(s - want)
    i = np.unravel_index(d.argmax(), d.shape)
    print('no_groups', ng, 'max diff', d.max(), 'at', i, 'p', p[i], 's', s[i], 'want', want[i], 'tol', np.abs(np.maximum(want, 1.0)).max() * 4e-7 + 1e-9)
    rel = d / (2e-6 * np.maximum(np.abs(p), 1.0))
    print('   second criterion max ratio', rel.max(), 'count bad', (rel > 1).sum(), 'first bad idx', np.argwhere(rel > 1)[:5].tolist())
    eng.close()
