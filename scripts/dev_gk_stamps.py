"""dev: timeline of the group kernel from a -DGK_STAMPS build (WN_LIB_PATH=vlibs/lib_stamps.so) of csrc/wn_iaf_g.hip with
profiles/r05_group_kernel_stamps.patch applied (the instrumentation is not part of the shipped kernel).  The stamp buffer holds the
LAST group launch of a call (the head group of the last flow, a decimated one) -- run with --flows to cut the student short."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsynth_wavenet_amd.engine import Engine
from nsynth_wavenet_amd import _lib
from oracle import wavenet_np as O
which = sys.argv[1] if len(sys.argv) > 1 else 'dec'
cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
# one flow whose LAST launch is the kind we want: [5] -> one natural group that is first AND last; [10] -> nat + dec(last)
cfgd['num_iaf_layers'] = {'nat': [5], 'dec': [10], 'mid': [10, 30]}[which]
w = O.synth_weights(O.HP(cfgd), 'student', seed=1234, init='tf')
eng = Engine(cfgd).load_weights(w)
mel = torch.from_numpy(np.random.RandomState(12345).uniform(0, 1, [1, 384, 80]).astype(np.float32)).cuda()
for i in range(5):
    eng.iaf_generate(mel, None, seed=i, want=('wav',), check_range=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 2 * 32))()
lib = ctypes.CDLL(_lib.LIB_PATH)
assert lib.wn_dbg_gk_stamps(buf) == 0
a = np.array(buf[:], dtype=np.uint64).reshape(4, 4, 32).astype(np.int64)
names = ['entry', 'prologue issued', 'prologue landed', 'barrier']
for j in range(5):
    names += ['L%d K done' % j, 'L%d epilogue math done' % j, 'L%d barrier 1' % j, 'L%d outputs + image written' % j, 'L%d barrier 2' % j]
print('# %s group, last launch of the call; cycles since entry (s_memtime); rows = stamps, columns = 4 workgroups x waves 0, 4, 8, 11' % which)
r0 = a[:, :, 30].min()
print('%-28s' % 'start (us after first, realtime)', ' '.join('%8.2f' % ((a[g, wv, 30] - r0) / 100.0) for g in range(4) for wv in range(4)))
print('%-28s' % 'lifetime us (realtime)', ' '.join('%8.2f' % ((a[g, wv, 31] - a[g, wv, 30]) / 100.0) for g in range(4) for wv in range(4)))
for k in range(1, 29):
    if k >= len(names): break
    vals = [(a[g, wv, k] - a[g, wv, 0]) if a[g, wv, k] else -1 for g in range(4) for wv in range(4)]
    print('%-28s' % names[k], ' '.join('%8d' % v for v in vals))
print('%-28s' % 'end', ' '.join('%8d' % (a[g, wv, 29] - a[g, wv, 0]) for g in range(4) for wv in range(4)))
