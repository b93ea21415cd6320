"""dev: per-layer time stamps of one workgroup of the segment-resident kernel (WN_SRF_DEBUG=<workgroup>)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd import config as cfg, weights as wts
from nsynth_wavenet_amd.engine import Engine
hp = cfg.load_hparams(json.load(open('config_jsons/parallel_wavenet.json')))
eng = Engine(hp, kind='student', precision=os.environ.get('WN_DEV_PRECISION', 'f16x3-resident')).load_weights(wts.synthetic_weights(hp, 'student', seed=1234, init='tf'))
mel = torch.from_numpy(np.random.RandomState(1).uniform(0, 1, [1, 384, 80]).astype(np.float32)).cuda()
os.environ.pop('WN_SRF_DEBUG', None); os.environ.pop('WN_RES_DEBUG', None)
for i in range(3):
    eng.iaf_generate(mel, None, seed=i)
torch.cuda.synchronize()
os.environ['WN_SRF_DEBUG'] = os.environ['WN_RES_DEBUG'] = sys.argv[1] if len(sys.argv) > 1 else '100'
eng.iaf_generate(mel, None, seed=9)
torch.cuda.synchronize()
