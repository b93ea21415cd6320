"""Randomised cross-checks of the execution forms against each other on the GPU (dev tool; run through
gpurun: `python tests/tools/fuzz_gpu.py [seconds]`).  Student: auto / fused / hoisted (layer groups forced on and off) / fp32 on random
(batch, frames); teacher: GEMV step vs batched step vs full-sequence forward on random (batch, length)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nsynth_wavenet_amd import weights as wts, config as cfg
from nsynth_wavenet_amd.engine import Engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 90.0
rs = np.random.RandomState(int(time.time()) & 0xffff)
bad = 0
t_end = time.time() + budget / 2
d = json.load(open(os.path.join(ROOT, 'config_jsons', 'parallel_wavenet.json')))
hp = cfg.load_hparams(d)
init = os.environ.get('WN_FUZZ_INIT') or ('unit' if rs.rand() < 0.5 else 'tf')
wseed = int(rs.randint(1 << 20))
print('student weights: init', init, 'seed', wseed, flush=True)
w = wts.synthetic_weights(hp, seed=wseed, init=init)
FORMS = ('f16x3-fused', 'f16x3-hoisted', 'f32', 'f32-fused')
engs = {p: Engine(d, precision=p).load_weights(w) for p in ('f16x3',) + FORMS}
os.environ['WN_DC_NO_PG'] = '1'           # (read once, in wn_create) the upsampler's last layer as phase-major GEMM + interleave
engs['f16x3-nopg'] = Engine(d, precision='f16x3').load_weights(w)
os.environ.pop('WN_DC_NO_PG')
FORMS = FORMS + ('f16x3-nopg',)
n = 0
while time.time() < t_end:
    B, F = int(rs.randint(1, 13)), int(rs.randint(3, 451))
    mel = torch.rand(B, F, 80, device='cuda')
    a = engs['f16x3'].iaf_generate(mel, None, seed=n, want=('x', 'rand_input', 'mean_tot', 'scale_tot'))
    if a['x'].shape[1] == 0:
        continue
    scale = max(1.0, float(a['x'].abs().max()))
    k2 = float((a['x'].double() - (a['rand_input'].double() * a['scale_tot'].double() + a['mean_tot'].double())).abs().max())
    # the forms share the split-fp16 arithmetic and differ in summation order only; 'f32' is plain fp32.  With
    # unit-variance weights the four flows amplify rounding by their scales (|x| up to 1e4), so a form is judged
    # against what fp32-vs-split rounding does on the same case: races show as >= 1e-3 * scale, rounding as ~1e-5
    errs = {}
    for p in FORMS:
        x = engs[p].iaf_generate(mel, a['rand_input'], want=('x',))['x']
        e = float((x - a['x']).abs().max())
        errs[p] = e if e == e else float('inf')
    # the hoisted form with the layer-group kernel forced on / off (the default picks by call size)
    for tag, mode in (('groups', True), ('per-layer', False)):
        engs['f16x3-hoisted'].set_layer_groups(mode)          # (the environment switches are read once, in wn_create)
        x = engs['f16x3-hoisted'].iaf_generate(mel, a['rand_input'], want=('x',))['x']
        engs['f16x3-hoisted'].set_layer_groups(None)
        e = float((x - a['x']).abs().max())
        errs[tag] = e if e == e else float('inf')
    # (round 6: 2e-5 * scale was marginal -- unit-gain weights amplify the forms' different summation orders to 2e-6 .. 8e-5 of
    # max|x| depending on the draw, scripts/dev_fuzz_repro.py: every form deterministic, pairwise 2e-6 .. 9e-6 on the cases that
    # tripped it.  A race moves samples by their own size; and every tenth case now runs each form twice, bit for bit.)
    tol = max(3e-4 * scale, 3.0 * errs['f32'])
    worst, ok = 0.0, errs['f32'] <= 3e-4 * scale
    for p in errs:
        if p != 'f32':
            worst = max(worst, errs[p])
            if not errs[p] <= tol:
                ok = False
                print('  form', p, 'differs by', errs[p], '(fp32 form: %.3g)' % errs['f32'], flush=True)
    if n % 10 == 0:
        for p in FORMS:
            x1 = engs[p].iaf_generate(mel, a['rand_input'], want=('x',))['x'].clone()
            x2 = engs[p].iaf_generate(mel, a['rand_input'], want=('x',))['x']
            if not torch.equal(x1, x2):
                ok = False
                print('  form', p, 'is NOT deterministic:', float((x1 - x2).abs().max()), flush=True)
    ok = ok and k2 <= 2e-6 * scale and bool(torch.isfinite(a['x']).all())
    bad += not ok
    n += 1
    if not ok or n % 10 == 0:
        print('student B=%d F=%d T=%d maxdiff %.2e K2 %.1e scale %.1f %s' % (B, F, a['x'].shape[1], worst, k2, scale, 'ok' if ok else 'BAD'), flush=True)
for e in engs.values():
    e.close()
print('student cases', n, 'bad', bad)

td = json.load(open(os.path.join(ROOT, 'config_jsons', 'wavenet_mol.json')))
td.update(dict(width=128, skip_width=64, deconv_width=64, num_layers=7, num_stages=3, deconv_config=[[8, 2], [12, 4]]))
thp = cfg.load_hparams(td)
tw = wts.synthetic_weights(thp, 'teacher', seed=3, init='unit')
t_end = time.time() + budget / 2
m = 0
while time.time() < t_end:
    B, F = int(rs.randint(1, 41)), int(rs.randint(2, 30))
    Tn = F * 8
    mel = rs.uniform(0, 1, [B, F, 80]).astype(np.float32)
    forced = rs.uniform(-1, 1, [B, Tn]).astype(np.float32)
    outs = {}
    for mode in ('gemv', 'mfma'):
        os.environ['WN_AR_MODE'] = mode
        eng = Engine(td).load_weights(tw)
        enc = eng.deconv(mel)
        outs[mode] = eng.ar_generate(enc, forced_wav=forced, want_out=True)['out_params'].clone()
        if mode == 'mfma':
            outs['fwd'] = eng.teacher_forward(forced, mel).clone()
        eng.close()
    os.environ.pop('WN_AR_MODE')
    sc = max(1.0, float(outs['fwd'].abs().max()))
    e1 = float((outs['gemv'] - outs['mfma']).abs().max()); e2 = float((outs['gemv'] - outs['fwd']).abs().max())
    ok = e1 <= 2e-5 * sc and e2 <= 5e-5 * sc
    bad += not ok
    m += 1
    if not ok or m % 5 == 0:
        print('teacher B=%d Tn=%d gemv-mfma %.2e gemv-forward %.2e %s' % (B, Tn, e1, e2, 'ok' if ok else 'BAD'), flush=True)
print('teacher cases', m, 'TOTAL bad', bad)
sys.exit(1 if bad else 0)
