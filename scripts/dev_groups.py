import os, sys, json, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nsynth_wavenet_amd.engine import Engine
from oracle import wavenet_np as O
cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
hp = O.HP(cfgd)
w = O.synth_weights(hp, 'student', seed=1234, init='tf')
eng = Engine(cfgd, precision='f16x3').load_weights(w)
for B, F in ((1, 11), (2, 35), (1, 384)):
    T = O.iaf_length(F, hp)
    mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80]).astype(np.float32)
    noise = O.logistic_from_uniform(np.random.RandomState(2).uniform(1e-5, 1 - 1e-5, [B, T]), np.float32)
    eng.set_layer_groups(True)
    a = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'))
    a = {k: v.cpu().numpy() for k, v in a.items()}
    eng.set_layer_groups(False)
    b = eng.iaf_generate(mel, noise, want=('x', 'mean_tot', 'scale_tot'))
    b = {k: v.cpu().numpy() for k, v in b.items()}
    eng.set_layer_groups(None)
    for k in a:
        d = np.abs(a[k] - b[k])
        print(B, F, T, k, 'max diff', d.max(), 'finite', np.isfinite(a[k]).all(), 'first bad', (np.argwhere(d > 1e-4)[:3].tolist() if d.max() > 1e-4 else None), 'fallbacks', eng.range_fallbacks)
    if F <= 35:
        ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
        print('  vs oracle', np.abs(a['x'] - ref['x']).max())
