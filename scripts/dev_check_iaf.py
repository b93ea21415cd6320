"""Ad-hoc GPU check of the IAF path against the numpy oracle (dev tool)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import wavenet_np as O
from nsynth_wavenet_amd.engine import Engine

cfgd = json.load(open(os.path.join(ROOT, 'config_jsons', 'parallel_wavenet.json')))
hp = O.HP(cfgd)
for init in ('tf', 'unit'):
    w = O.synth_weights(hp, 'student', init=init)
    eng = Engine(cfgd).load_weights(w)
    for (B, F) in ((1, 8), (2, 11)):
        mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80]).astype(np.float32)
        T = O.iaf_length(F, hp)
        u = np.random.RandomState(2).uniform(1e-5, 1 - 1e-5, [B, T])
        noise = O.logistic_from_uniform(u, np.float32)
        # deconv alone
        enc = eng.deconv(mel).cpu().numpy()
        enc_o = O.deconv_stack(mel, w, hp, 'iaf_share', np.float64)
        print(init, B, F, 'deconv maxdiff', np.abs(enc - enc_o).max(), 'absmax', np.abs(enc_o).max())
        out = eng.iaf_generate(mel, noise, want=('wav', 'idx', 'x', 'mean_tot', 'scale_tot'))
        torch.cuda.synchronize()
        ff = O.iaf_feed_forward(mel, noise, w, hp, np.float64)
        wav_o, idx_o = O.clip_quant_scale(ff['x'], 65536, False, np.float64)
        for k, ref in (('x', ff['x']), ('mean_tot', ff['mean_tot']), ('scale_tot', ff['scale_tot']), ('wav', wav_o)):
            print('   ', k, 'maxdiff', np.abs(out[k].cpu().numpy() - ref).max(), 'ref absmax', np.abs(ref).max())
        print('    idx mismatches', (out['idx'].cpu().numpy() != idx_o).sum(), 'of', idx_o.size)
    eng.close()
# timing at config 2
w = O.synth_weights(hp, 'student', init='tf')
eng = Engine(cfgd).load_weights(w)
mel = np.random.RandomState(12345).uniform(0, 1, [1, 384, 80]).astype(np.float32)
melg = torch.as_tensor(mel).cuda()
for i in range(3):
    eng.iaf_generate(melg, None, seed=i)
torch.cuda.synchronize()
t = time.time()
n = 10
for i in range(n):
    eng.iaf_generate(melg, None, seed=i)
torch.cuda.synchronize()
dt = (time.time() - t) / n
print('config2: %.3f ms per utterance, %.2f M samples/s, %.0fx RT' % (dt * 1e3, 76800 / dt / 1e6, 76800 / dt / 16000))
