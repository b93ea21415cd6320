import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nsynth_wavenet_amd.engine import Engine
from oracle import wavenet_np as O
cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
hp = O.HP(cfgd)
w = O.synth_weights(hp, 'student', seed=1234, init='tf')
eng = Engine(cfgd, precision='f16x3').load_weights(w)
os.environ['WN_GROUPS'] = '1'
for B, F in ((1, 11), (2, 35)):
    T = O.iaf_length(F, hp)
    mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(np.random.RandomState(2).uniform(1e-5, 1 - 1e-5, [B, T]), np.float32)
    ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
    for rep in range(4):
        out = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'), check_range=False)
        a = out['x'].cpu().numpy()
        d = np.abs(a - ref['x'])
        bad = np.argwhere(d > 1e-3)
        if len(bad) == 0:
            print(B, F, rep, 'ok', d.max()); continue
        for b in range(B):
            tb = bad[bad[:, 0] == b][:, 1]
            if len(tb) == 0: continue
            res = sorted(set((tb % 32).tolist()))
            firsts = {r: int(tb[tb % 32 == r].min()) for r in res}
            print(B, F, rep, 'b', b, 'nbad', len(tb), 'first', int(tb.min()), 'last', int(tb.max()), 'residues', res[:40], 'first per residue', dict(list(firsts.items())[:8]), flush=True)
            t0 = int(tb.min())
            print('   d around first', np.round(d[b, t0 - 2:t0 + 40], 4).tolist())
