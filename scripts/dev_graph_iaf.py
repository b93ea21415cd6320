"""Does replaying wn_iaf_generate from a HIP graph shrink the inter-kernel gaps? (dev tool)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nsynth_wavenet_amd import weights as wts, config as cfg
from nsynth_wavenet_amd.engine import Engine
d = json.load(open(os.path.join(ROOT, 'config_jsons', 'parallel_wavenet.json')))
hp = cfg.load_hparams(d)
eng = Engine(d).load_weights(wts.synthetic_weights(hp, seed=1234))
mel = torch.rand(1, 384, 80, device='cuda')
for i in range(3):
    ref = eng.iaf_generate(mel, None, seed=5, want=('wav',))['wav'].clone()
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
print('eager  %.3f ms' % timeit(lambda: eng.iaf_generate(mel, None, seed=5, want=('wav',))))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    eng.iaf_generate(mel, None, seed=5, want=('wav',))
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = eng.iaf_generate(mel, None, seed=5, want=('wav',))['wav']
g.replay(); torch.cuda.synchronize()
print('graph output identical:', bool(torch.equal(out, ref)))
print('graph  %.3f ms' % timeit(g.replay))
