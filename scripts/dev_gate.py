import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from nsynth_wavenet_amd.engine import Engine
from oracle import wavenet_np as O
cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
hp = O.HP(cfgd)
w = O.synth_weights(hp, 'student', seed=1234, init='tf')
for prec in ('f16x3', 'f16x3-fused'):
    eng = Engine(cfgd, precision=prec).load_weights(w)
    for B, F in ((1, 11), (2, 35)):
        T = O.iaf_length(F, hp)
        mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80]).astype(np.float32)
        noise = O.logistic_from_uniform(np.random.RandomState(2).uniform(1e-5, 1 - 1e-5, [B, T]), np.float32)
        ref = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
        for mode in ('groups', 'nogroups'):
            eng.set_layer_groups(mode == 'groups')
            errs = []
            for rep in range(3):
                a = eng.iaf_generate(mel, noise, want=('x',), check_range=False)['x'].cpu().numpy()
                d = np.abs(a - ref['x'])
                errs.append((float(d.max()), np.argwhere(d > 1e-3)[:2].tolist()))
            print(prec, B, F, mode, errs, flush=True)
