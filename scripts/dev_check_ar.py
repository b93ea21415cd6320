"""Ad-hoc GPU check of the AR path against the numpy oracle (dev tool)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import wavenet_np as O
from nsynth_wavenet_amd.engine import Engine

for name in ('wavenet_mol', 'wavenet_ce', 'wavenet_gauss'):
    d = json.load(open(os.path.join(ROOT, 'config_jsons', name + '.json')))
    d.update(width=128, skip_width=64, deconv_width=64, num_layers=7, num_stages=3, deconv_config=[[8, 2], [12, 4]])
    if name == 'wavenet_ce':
        d['double_gate_width'] = False
    hp = O.HP(d)
    w = O.synth_weights(hp, 'teacher', init='unit')
    eng = Engine(d).load_weights(w)
    B, F = 3, 6
    mel = np.random.RandomState(1).uniform(0, 1, [B, F, 80]).astype(np.float32)
    enc = eng.deconv(mel).cpu().numpy()
    enc_o = O.deconv_stack(mel, w, hp, '', np.float64)
    Tn = enc_o.shape[1]
    print(name, 'deconv maxdiff', np.abs(enc - enc_o).max(), 'Tn', Tn)
    fg = O.Fastgen(w, hp, B, np.float32)
    nr = fg.n_rand()
    rs = np.random.RandomState(5)
    rnd = rs.standard_normal([Tn, B, nr]) if hp.loss_type == 'gauss' else rs.uniform(1e-5, 1 - 1e-5, [Tn, B, nr])
    rnd = rnd.astype(np.float32)
    # teacher forced (K1) : out_params vs oracle full-sequence forward
    wavf = np.random.RandomState(3).uniform(-1, 1, [B, Tn]).astype(np.float32)
    out = eng.ar_generate(enc_o.astype(np.float32), rnd, forced_wav=wavf, want_out=True)
    torch.cuda.synchronize()
    ws = O.encode_signal(wavf, hp, np.float64)
    ref = O.teacher_feed_forward(ws, enc_o, w, hp, np.float64)
    print('   K1 forced out_params maxdiff', np.abs(out['out_params'].cpu().numpy() - ref).max(), 'ref absmax', np.abs(ref).max())
    # free running vs oracle (fp32)
    out = eng.ar_generate(enc_o.astype(np.float32), rnd, want_out=True)
    torch.cuda.synchronize()
    wav_o, idx_o, outs_o = O.fastgen_synthesis(enc_o.astype(np.float32), rnd, w, hp, np.float32, return_out=True)
    gi = out['idx'].cpu().numpy()
    print('   free-run idx mismatches', (gi != idx_o).sum(), 'of', idx_o.size, 'max |didx|', np.abs(gi - idx_o).max(),
          'first mismatch step', (np.argwhere(gi != idx_o)[:, 1].min() if (gi != idx_o).any() else -1))
    # sampler given the GPU's own out_params
    gop = out['out_params'].cpu().numpy()
    mism = 0
    for t in range(Tn):
        q = fg.sample_from(gop[:, t], rnd[t])
        mism += (q != gi[:, t]).sum()
    print('   sampler-on-GPU-logits mismatches', mism)
    # step API
    st = eng.ar_new_state(B)
    a = np.zeros([B], np.float32)
    fg2 = O.Fastgen(w, hp, B, np.float32)
    md = 0
    for t in range(10):
        s, op = eng.ar_step(st, a, enc_o[:, t].astype(np.float32), rnd[t], want_out=True)
        op_o = fg2.out_params(a.reshape(B, 1), enc_o[:, t].astype(np.float32))
        md = max(md, np.abs(op.cpu().numpy() - op_o).max())
        a = fg2.dequant(s.cpu().numpy()).astype(np.float32)
    print('   step API out_params maxdiff', md)
    eng.close()

# full-size timing
d = json.load(open(os.path.join(ROOT, 'config_jsons', 'wavenet_mol.json')))
hp = O.HP(d)
w = O.synth_weights(hp, 'teacher', init='tf')
eng = Engine(d).load_weights(w)
for B in (1, 8):
    Tn = 1600
    enc = torch.randn(B, Tn, 256, device='cuda') * 0.1
    eng.ar_generate(enc, None, seed=1)
    torch.cuda.synchronize()
    t = time.time()
    eng.ar_generate(enc, None, seed=2)
    torch.cuda.synchronize()
    dt = time.time() - t
    print('AR full-size B=%d: %.1f us/step, %.0f samples/s total (%.2fx RT per utterance)' % (B, dt / Tn * 1e6, B * Tn / dt, Tn / dt / 16000))
