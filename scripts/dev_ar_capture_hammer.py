"""dev: wn_ar_generate with graphs on while ANOTHER thread issues device-wide synchronises (ROCm 7.2 invalidates the capture): every
call must still equal the serial call -- the capture is ended, retried, replaced by plain launches.  WN_AR_DEBUG=1 shows which."""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import wavenet_np as O
from nsynth_wavenet_amd.engine import Engine
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ar_mol.npz'))
cfgd = json.loads(str(g['cfg_json']))
hp = O.HP(cfgd)
eng = Engine(cfgd).load_weights(O.synth_weights(hp, 'teacher', seed=1234, init='unit'))
enc, rnd = g['enc'], g['rnd']
eng._set_ar_graph(True)
serial = {k: v.clone() for k, v in eng.ar_generate(enc, rnd, want_out=True, use_graph=True).items()}
torch.cuda.synchronize()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = refused = syncs = 0
for r in range(rounds):
    stop = threading.Event()
    def hammer():
        global refused, syncs
        while not stop.is_set():
            try:
                torch.cuda.synchronize()
                syncs += 1
            except RuntimeError:
                refused += 1
    th = threading.Thread(target=hammer)
    th.start()
    f = eng.fork()
    st = torch.cuda.Stream()
    got = []
    try:
        with torch.cuda.stream(st):
            for _ in range(6):
                got.append(f.ar_generate(enc, rnd, want_out=True, use_graph=True))
        st.synchronize()
    except Exception as e:           # noqa
        bad += 1
        print('round', r, 'call FAILED:', str(e)[:200])
    stop.set()
    th.join()
    f.close()
    for o in got:
        for k in ('idx', 'wav', 'out_params'):
            if not torch.equal(o[k], serial[k]):
                bad += 1
                print('round', r, 'result differs in', k)
print('%d rounds, %d bad, %d device synchronises done beside them, %d refused by the runtime' % (rounds, bad, syncs, refused))
