"""dev: where a launch of the hoisted fp32 layer kernel spends its time -- s_memtime stamps of one workgroup (a library built
with -DWN_F32_STAMPS=<workgroup + 1>, loaded through WN_LIB_PATH).  Slots per wave: 0 entry, 1 first loads issued, 2 image
staged; per tile i: 3+4i tile top, 4+4i K loop done, 5+4i gate + residual 1x1 done, 6+4i stores issued."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsynth_wavenet_amd import _lib                           # noqa: E402
from nsynth_wavenet_amd.engine import Engine                  # noqa: E402
from oracle import wavenet_np as O                            # noqa: E402

cfgd = json.load(open(os.path.join(ROOT, 'config_jsons', 'parallel_wavenet.json')))
eng = Engine(cfgd, precision='f32').load_weights(O.synth_weights(O.HP(cfgd), 'student', seed=1234, init='tf'))
mel = torch.from_numpy(np.random.RandomState(12345).uniform(0, 1, [1, 384, 80]).astype(np.float32)).cuda()
for i in range(30):
    eng.iaf_generate(mel, None, seed=i, want=('wav',))
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_uint64 * 256)()
assert lib.wn_debug_f32_stamps(buf) == 0
st = np.array(buf, dtype=np.uint64).reshape(8, 32).astype(np.int64)
t0 = st[:, 0].min()
print('cycles relative to the earliest wave entry (s_memtime: 100 MHz ticks on this part if small, shader clocks if large)')
for w in range(8):
    r = st[w] - t0
    tiles = []
    for i in range(6):
        if st[w][3 + 4 * i] == 0 or 6 + 4 * i >= 32:
            break
        a, b, c, d = r[3 + 4 * i: 7 + 4 * i]
        tiles.append('top %6d K %5d epi %5d st %4d' % (a, b - a, c - b, d - c))
    print('wave %d: entry %5d loads-issued %5d staged %6d | %s' % (w, r[0], r[1], r[2], ' | '.join(tiles)))
eng.close()
