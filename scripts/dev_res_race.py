"""dev: find the (layer, block) where two runs of the hoisted-resident form first differ (race hunting)."""
import json, os, sys
os.environ.setdefault('WN_UNVERIFIED_FORMS', '1')     # the hoisted-resident form is withheld
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd import config as cfg, weights as wts
from nsynth_wavenet_amd.engine import Engine
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = dict(json.load(open('config_jsons/parallel_wavenet.json')), num_iaf_layers=[NL])
hp = cfg.load_hparams(d)
w = wts.synthetic_weights(hp, seed=7, init='unit')
b = Engine(d, precision='f16x3-hoisted-resident').load_weights(w)
F = int(sys.argv[2]) if len(sys.argv) > 2 else 246
mel = torch.rand(1, F, 80, device='cuda')
T = b.iaf_length(F); TE = F * 200; RS = 1024 + T
noise = torch.randn(1, T, device='cuda')
print('T', T, 'blocks', T // 16, 'base', T // 16 // 256, 'rem', T // 16 % 256)
al = lambda n: (n + 255) // 256 * 256
enc_bytes = al((256 * TE + 64) * 4)
buf_words = 64 * RS
def snap():
    ws = b._ws
    return ws[enc_bytes:enc_bytes + (NL + 1) * buf_words * 4].view(torch.int32).view(NL + 1, 16, RS, 4).clone()
ref = None
for it in range(12):
    x = b.iaf_generate(mel, noise, want=('x',))['x']
    torch.cuda.synchronize()
    s = snap()
    if ref is None:
        ref, refx = s, x.clone()
        continue
    if torch.equal(x, refx):
        continue
    for j in range(NL + 1):
        neq = (s[j] != ref[j]).any(dim=0).any(dim=1)      # per column
        if bool(neq.any()):
            cols = neq.nonzero().flatten() - 1024
            blk = (cols // 16).unique()
            print('iter', it, 'first differing buffer', j, '(output of layer %d, dilation %d)' % (j - 1, 2 ** ((j - 1) % 10)),
                  'blocks', blk[:10].tolist(), 'n', len(blk), 'first block', blk[0].item())
            # which rows / words differ in the first block
            c0 = int(cols[0]) + 1024
            dif = (s[j][:, c0:c0 + 16] != ref[j][:, c0:c0 + 16])
            # decode the split-fp16 words of the first differing column: hi/lo half pairs -> floats
            def dec(t):
                hi = t[:8].contiguous().view(torch.float16).float().view(8, 4, 2)
                lo = t[8:].contiguous().view(torch.float16).float().view(8, 4, 2)
                return (hi + lo)
            va, vb = dec(s[j][:, c0]), dec(ref[j][:, c0])
            dd = (va - vb)
            print('   group rows x slot x half differences:', dd.nonzero().tolist()[:8], 'values', va[dd != 0][:8].tolist(), 'vs', vb[dd != 0][:8].tolist())
            cols16 = [float((dec(s[j][:, c0 + k]) - dec(ref[j][:, c0 + k])).abs().max()) for k in range(16)]
            print('   |difference| per column of the block:', ['%.4f' % v for v in cols16])
            lprev_a = torch.stack([dec(s[j - 1][:, c0 + k])[3, 0, 0] for k in range(16)]) if j > 0 else None
            oa = torch.stack([dec(s[j][:, c0 + k]).flatten()[dd.flatten().nonzero()[0, 0]] for k in range(16)])
            ob = torch.stack([dec(ref[j][:, c0 + k]).flatten()[dd.flatten().nonzero()[0, 0]] for k in range(16)])
            la = torch.stack([dec(s[j - 1][:, c0 + k]).flatten()[dd.flatten().nonzero()[0, 0]] for k in range(16)])
            print('   this run - skip input :', ['%.4f' % v for v in (oa - la).tolist()])
            print('   first run - skip input:', ['%.4f' % v for v in (ob - la).tolist()])
            print('   rows differing', dif.any(dim=1).any(dim=1).nonzero().flatten().tolist(), 'cols in block', dif.any(dim=0).any(dim=1).nonzero().flatten().tolist())
            break
    break
else:
    print('no difference in 12 runs')
