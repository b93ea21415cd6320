"""dev: package power / shader clock beside sustained loops of the teacher paths (full-sequence forward, AR step at 1 and 64
utterances).  python scripts/dev_power_teacher.py"""
import glob, json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd import config as cfg, weights as wts
from nsynth_wavenet_amd.engine import Engine

p = torch.cuda.get_device_properties(0)
HW = glob.glob('/sys/bus/pci/devices/{:04x}:{:02x}:{:02x}.0/hwmon/hwmon*'.format(p.pci_domain_id, p.pci_bus_id, p.pci_device_id))[0]


def rd(n):
    return int(open(os.path.join(HW, n)).read())


class S(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True); self.on = True; self.p = []; self.f = []
    def run(self):
        while self.on:
            self.p.append(rd('power1_input') * 1e-6); self.f.append(rd('freq1_input') * 1e-6); time.sleep(0.02)


hp = cfg.load_hparams(json.load(open('config_jsons/wavenet_mol.json')))
eng = Engine(hp, kind='teacher').load_weights(wts.synthetic_weights(hp, 'teacher', seed=1, init='unit'))
rs = np.random.RandomState(0)
F = 384; T = F * cfg.frame_shift(hp)
mel = torch.as_tensor(rs.uniform(0, 1, [1, F, 80]).astype(np.float32)).cuda()
wav = torch.as_tensor(rs.uniform(-1, 1, [1, T]).astype(np.float32)).cuda()
encs = {B: torch.as_tensor((rs.standard_normal([B, 800, hp.deconv_width]) * 0.1).astype(np.float32)).cuda() for B in (1, 64)}
jobs = [('teacher_forward 4.8 s', lambda: eng.teacher_forward(wav, mel), T),
        ('ar 1 utterance x 800', lambda: eng.ar_generate(encs[1], None, seed=1), 800),
        ('ar 64 utterances x 800', lambda: eng.ar_generate(encs[64], None, seed=1), 64 * 800)]
for tag, fn, units in jobs:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()
    s = S(); s.start(); n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        fn(); n += 1; torch.cuda.synchronize()
    dt = time.perf_counter() - t0; s.on = False; s.join()
    print('{:26s} {:9.3f} ms/call  {:7.1f} W  {:6.0f} MHz  {:8.4f} J/call'.format(tag, dt / n * 1e3, np.mean(s.p), np.mean(s.f), np.mean(s.p) * dt / n), flush=True)
eng.close()
