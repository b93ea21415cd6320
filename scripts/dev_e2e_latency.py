import json, os, sys, time, tempfile
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
from nsynth_wavenet_amd import weights as wts, config as cfg
from nsynth_wavenet_amd.wavenet import parallelgen
d = json.load(open('/root/repo/config_jsons/parallel_wavenet.json'))
hp = cfg.load_hparams(d)
tmp = tempfile.mkdtemp()
ck = wts.save_checkpoint(os.path.join(tmp, 'model.ckpt-1'), wts.synthetic_weights(hp, seed=1234), hp)
mel = np.random.RandomState(0).uniform(0, 1, [1, 384, 80]).astype(np.float32)
for i in range(3): a = parallelgen.generate(hp, mel, ck)
t = time.time(); n = 50
for i in range(n): a = parallelgen.generate(hp, mel, ck)
print('parallelgen.generate (H2D mel + generate + D2H wav), numpy in/out: %.3f ms per 4.8 s utterance' % ((time.time() - t) / n * 1e3))
eng = parallelgen.load_parallelgen(hp, ck)
melg = torch.as_tensor(mel).cuda()
torch.cuda.synchronize(); t = time.time()
for i in range(n): w = eng.iaf_generate(melg, None, seed=i, want=('wav',))['wav']
torch.cuda.synchronize(); print('engine.iaf_generate resident: %.3f ms' % ((time.time() - t) / n * 1e3))
t = time.time()
for i in range(n): w = eng.iaf_generate(melg, None, seed=i, want=('wav',))['wav']; torch.cuda.synchronize()
print('engine.iaf_generate + sync each call: %.3f ms' % ((time.time() - t) / n * 1e3))
