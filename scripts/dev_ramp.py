"""dev: per-call duration of the first N generate calls on an idle GPU (the clock / power transient behind the driver's
25-step protocol).  python scripts/dev_ramp.py"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd.engine import Engine
from oracle import wavenet_np as O
cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
w = O.synth_weights(O.HP(cfgd), 'student', seed=1234, init='tf')
eng = Engine(cfgd).load_weights(w)
mel = torch.from_numpy(np.random.RandomState(12345).uniform(0, 1, [1, 384, 80]).astype(np.float32)).cuda()
eng.iaf_generate(mel, None, seed=0, want=('wav',), check_range=False)     # code objects loaded, workspace allocated
torch.cuda.synchronize()
for idle in (2.0, 0.2):
    time.sleep(idle)
    N = 120
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
    print('after %.1f s idle: ms per call, calls 1-10:' % idle, ' '.join('%.3f' % x for x in d[:10]))
    for lo in (10, 20, 30, 40, 60, 80, 100):
        print('   calls %3d-%3d: mean %.4f ms' % (lo + 1, lo + 10, float(np.mean(d[lo:lo + 10]))))
    print('   driver window (calls 6-25): %.4f ms;  calls 61-120: %.4f ms' % (float(np.mean(d[5:25])), float(np.mean(d[60:]))))
eng.close()
