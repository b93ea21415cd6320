"""CPU stand-in for the reference's fastgen loop at full width (numpy restatement), dev tool."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import wavenet_np as O
d = json.load(open(os.path.join(ROOT, 'config_jsons', 'wavenet_mol.json')))
hp = O.HP(d)
w = O.synth_weights(hp, 'teacher', seed=1, init='unit')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(0)
enc = (rs.standard_normal([1, n, 256]) * 0.1).astype(np.float32)
fg = O.Fastgen(w, hp, 1, np.float32)
rnd = rs.uniform(1e-5, 1 - 1e-5, [n, 1, fg.n_rand()]).astype(np.float32)
t = time.time()
O.fastgen_synthesis(enc, rnd, w, hp, np.float32)
dt = time.time() - t
print('CPU numpy fastgen, width 512 x 30 layers, 1 utterance: %.0f us/step = %.0f samples/s (%.3fx RT), cores available %d' % (
    dt / n * 1e6, n / dt, n / dt / 16000, os.cpu_count()))
