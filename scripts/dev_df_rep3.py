"""dev: each of several repeated calls of a build against the saved output of a reference build (WN_REF_SAVE=1 saves)"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd.engine import Engine
from oracle import wavenet_np as O
for layers in ([1], [5], [10]):
    cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
    cfgd['num_iaf_layers'] = layers
    w = O.synth_weights(O.HP(cfgd), 'student', seed=1234, init='tf')
    eng = Engine(cfgd).load_weights(w)
    mel = torch.from_numpy(np.random.RandomState(12345).uniform(0, 1, [1, 16, 80]).astype(np.float32)).cuda()
    xs = [eng.iaf_generate(mel, None, seed=7, want=('x',), check_range=False)['x'].cpu().numpy()[0] for _ in range(5)]
    f = '/tmp/df_ref_%d.npy' % sum(layers)
    if os.environ.get('WN_REF_SAVE'):
        np.save(f, xs[0])
        print('saved', layers, [float(np.abs(x - xs[0]).max()) for x in xs])
        continue
    ref = np.load(f)
    print(os.path.basename(os.environ.get('WN_LIB_PATH', 'default')), layers, 'per call: max |x - ref| and number of 16-blocks off:',
          ['%.1e/%d' % (np.abs(x - ref).max(), (np.abs(x - ref).reshape(-1, 16) > 1e-6).any(1).sum()) for x in xs])
