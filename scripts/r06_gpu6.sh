#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
timeout 900 python -m pytest tests/test_gpu_iaf.py::test_golden_vectors "tests/test_ref_float.py::test_engine_student_against_the_reference_code" tests/test_gpu_threads.py::test_one_student_handle_two_threads_two_streams -x -q -m gpu -k "f32 or threads" 2>&1 | tail -3
for bg in 0 1; do
  WN_F32_BG=$bg timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --precision f32 --layer-events-every 1000000 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WN_F32_BG=$bg: value %.2f M, %.3f ms, path frac of f32 peak %.3f' % (d['value']/1e6, d['ms_per_step'], d['config']['path_achieved_tflops']/157.3))"
done
WN_PRECISION=f32 timeout 200 python - <<'PY'
import json, numpy as np, torch, sys
sys.path.insert(0, '.')
from nsynth_wavenet_amd.engine import Engine
from oracle import wavenet_np as O
cfgd = json.load(open('config_jsons/parallel_wavenet.json'))
hp = O.HP(cfgd)
w = O.synth_weights(hp, 'student', seed=1234, init='tf')
import os
outs = {}
for bg in ('0', '1'):
    os.environ['WN_F32_BG'] = bg
    eng = Engine(cfgd, precision='f32').load_weights(w)
    mel = np.random.RandomState(1).uniform(0, 1, [2, 384, 80]).astype(np.float32)
    xs = [eng.iaf_generate(mel, None, seed=5, want=('x',))['x'].clone() for _ in range(6)]
    assert all(torch.equal(xs[0], x) for x in xs), 'not repeatable with WN_F32_BG=' + bg
    outs[bg] = xs[0]
    eng.close()
print('background GEMM on/off bit-identical:', bool(torch.equal(outs['0'], outs['1'])), ' repeatable: True')
PY
