"""dev: package power, shader clock and energy of every PART of a generate call (wn_profile_parts_only), sampled from the
GPU's hwmon nodes beside a sustained loop of that part.  configs[1] by default.
    python scripts/dev_power.py [--batch B] [--seconds S]"""
import argparse, glob, json, os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd.engine import Engine      # noqa: E402
from oracle import wavenet_np as O                # noqa: E402  (synthetic weights only)

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--frames', type=int, default=384)
ap.add_argument('--seconds', type=float, default=3.0)
ap.add_argument('--precision', default=None)
a = ap.parse_args()


def hwmon_of_gpu0():
    p = torch.cuda.get_device_properties(0)
    try:
        bdf = '{:04x}:{:02x}:{:02x}.0'.format(p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        hw = glob.glob('/sys/bus/pci/devices/{}/hwmon/hwmon*'.format(bdf))
        if hw:
            return hw[0]
    except AttributeError:
        pass
    cands = glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')
    return cands[0] if cands else None


HW = hwmon_of_gpu0()
print('hwmon', HW, flush=True)


def read_int(name):
    with open(os.path.join(HW, name)) as f:
        return int(f.read().strip())


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.on = True
        self.p, self.f = [], []

    def run(self):
        while self.on:
            try:
                self.p.append(read_int('power1_input') * 1e-6)
                self.f.append(read_int('freq1_input') * 1e-6)
            except OSError:
                pass
            time.sleep(0.02)


cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
w = O.synth_weights(O.HP(cfgd), 'student', seed=1234, init='tf')
eng = Engine(cfgd, precision=a.precision).load_weights(w)
mel = torch.from_numpy(np.random.RandomState(12345).uniform(0, 1, [a.batch, a.frames, 80]).astype(np.float32)).cuda()
T = eng.iaf_length(a.frames)
for i in range(20):
    eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
torch.cuda.synchronize()
rows = []
# wn_profile_parts_only is accepted only inside a parts session (it makes a call SKIP work and cannot be left armed); the
# session keeps at most 4096 part events, so the event bubbles end inside the first warm-up second
eng.profile_parts_begin()
for tag, mask in (('all', 15), ('upsampler', 2), ('cond_gemm', 4), ('residual_stack', 8), ('prologue_epilogue', 1), ('all_again', 15)):
    eng.profile_parts_only(15)
    for i in range(3):
        eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)      # buffers of a full call
    eng.profile_parts_only(mask)
    torch.cuda.synchronize()
    # warm: reach the steady operating point before sampling
    t_end = time.perf_counter() + 1.0
    while time.perf_counter() < t_end:
        for i in range(50):
            eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
        torch.cuda.synchronize()
    s = Sampler()
    s.start()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        for i in range(50):
            eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
        n += 50
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s.on = False
    s.join()
    P, Fq = float(np.mean(s.p)) if s.p else float('nan'), float(np.mean(s.f)) if s.f else float('nan')
    us = dt / n * 1e6
    rows.append((tag, us, P, Fq, P * us * 1e-6))
    print('{:18s} {:9.1f} us/call  {:7.1f} W  {:6.0f} MHz  {:7.4f} J/call   ({} samples)'.format(tag, us, P, Fq, P * us * 1e-6, len(s.p)), flush=True)
eng.profile_parts_end()
tot = sum(r[4] for r in rows if r[0] in ('upsampler', 'cond_gemm', 'residual_stack', 'prologue_epilogue'))
print('sum of the parts: {:.4f} J, {:.1f} us; whole call: {:.4f} J, {:.1f} us'.format(
    tot, sum(r[1] for r in rows if r[0] in ('upsampler', 'cond_gemm', 'residual_stack', 'prologue_epilogue')), rows[0][4], rows[0][1]))
eng.close()
