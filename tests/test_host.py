"""CPU tests of the host-side logic and of the C-ABI library's surface (no compute calls)."""
import ctypes
import json
import os
import re
from argparse import Namespace

import numpy as np
import pytest

from conftest import ROOT, load_json
from nsynth_wavenet_amd import _lib
from nsynth_wavenet_amd import config as cfg
from nsynth_wavenet_amd import weights as wts


# the reference's own JSON (with its training-only keys) must load unchanged
REFERENCE_STYLE_STUDENT = {
    "lr_schedule": [[0, 1e-4], [90000, 6e-5]], "num_iters": 400000, "wave_length": 7680, "num_stages": 10,
    "num_iaf_layers": [10, 10, 10, 30], "filter_length": 3, "width": 64, "deconv_width": 256,
    "deconv_config": [[40, 10], [80, 20]], "use_mu_law": False, "loss_type": "logistic",
    "use_weight_norm": False, "use_resize_conv": False, "use_share_deconv": True, "use_teacher_deconv": False,
    "upsample_act": "leaky_relu", "num_samples": 100, "power_loss_factor": 1.0, "contrastive_loss_factor": 0.3}


def test_library_is_built_and_exports_every_declared_symbol():
    """The C ABI is the boundary: what include/wnhip.h declares (WN_API) and the library's dynamic symbol table are the SAME
    set -- nothing declared is missing, and no internal C++ helper, STL instantiation or per-unit marker leaks out
    (libwnhip.so is linked with -fvisibility=hidden and a version script, nsynth_wavenet_amd/build.py)."""
    import subprocess
    header = open(os.path.join(ROOT, 'include', 'wnhip.h')).read()
    declared = sorted(set(re.findall(r'^WN_API [^;(]*?\b(wn_[a-z_0-9]+)\s*\(', header, re.M)))
    assert declared == sorted(set(re.findall(r'\b(wn_[a-z_0-9]+)\s*\(', header))), 'a declaration without WN_API'
    assert 'wn_iaf_generate' in declared and 'wn_ar_generate' in declared and len(declared) >= 17
    assert os.path.exists(_lib.LIB_PATH), 'run python -m nsynth_wavenet_amd.build'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sorted(_lib.SYMBOLS) == declared
    nm = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in nm.splitlines() if ln.strip())
    assert exported == declared, ('exported but not declared: {}; declared but not exported: {}'.format(
        sorted(set(exported) - set(declared)), sorted(set(declared) - set(exported))))
    assert _lib.load().wn_abi_version() == 1


def test_compiled_kernels_are_free_of_the_hazards_hipcc_does_not_guard(tmp_path):
    """gfx950 loses lanes of a buffer_store_dwordx4 whose data register is overwritten by the VALU instruction behind
    it, a form hipcc's hazard recognizer exempts (profiles/r03_store_hazard.txt): the audit reads the compiler's own
    assembly of every kernel source and must come back empty; it must also SEE the pattern in a crafted listing."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import audit_store_hazard as audit
    crafted = tmp_path / 'k.s'
    crafted.write_text('_Zkernel:\n\tbuffer_store_dwordx4 v[48:51], v115, s[40:43], s75 offen\n.LBB0_34:\n'
                       '\tv_max_f32_e64 v48, |v84|, |v85|\n\ts_endpgm\n'
                       '_Zfenced:\n\tbuffer_store_dwordx4 v[48:51], v115, s[40:43], s75 offen\n\t;;#ASMSTART\n\ts_nop 1\n'
                       '\t;;#ASMEND\n\tv_max_f32_e64 v48, |v84|, |v85|\n\ts_endpgm\n')
    found = audit.audit(str(crafted))
    assert len(found) == 1 and found[0][0] == '_Zkernel'
    # second rule: an MFMA reading the result of a VALU instruction inside an asm statement with fewer than two wait
    # states between them (scripts/ubench/valu_to_mfma.hip: it gets the old register content)
    crafted.write_text('_Zearly:\n\t;;#ASMSTART\n\tv_fma_mixhi_f16 v87, v51, -1.0, v88 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND\n'
                       '\tv_mov_b32_e32 v88, v97\n\tv_mfma_f32_16x16x32_f16 v[100:103], v[80:83], v[84:87], v[60:63]\n\ts_endpgm\n'
                       '_Zpadded:\n\t;;#ASMSTART\n\tv_fma_mixhi_f16 v87, v51, -1.0, v88 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND\n'
                       '\t;;#ASMSTART\n\ts_nop 1\n\t;;#ASMEND\n\tv_mfma_f32_16x16x32_f16 v[100:103], v[80:83], v[84:87], v[60:63]\n\ts_endpgm\n'
                       '_Zone_state:\n\t;;#ASMSTART\n\tv_fma_mixhi_f16 v87, v51, -1.0, v88 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t;;#ASMEND\n'
                       '\ts_nop 0\n\tv_mfma_f32_16x16x32_f16 v[100:103], v[80:83], v[84:87], v[60:63]\n\ts_endpgm\n')
    found = audit.audit(str(crafted))
    assert sorted(f[0] for f in found) == ['_Zearly', '_Zone_state']
    # third rule: packed fp32 that takes the HIGH register of src1 for the low lane (tests/test_gpu_hazards.py)
    crafted.write_text('_Zhigh:\n\tv_pk_fma_f32 v[14:15], v[14:15], v[86:87], v[82:83] op_sel:[0,1,0]\n\ts_endpgm\n'
                       '_Zswapped:\n\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]\n\ts_endpgm\n'
                       '_Zlow:\n\tv_pk_fma_f32 v[14:15], v[14:15], v[70:71], v[82:83] op_sel_hi:[1,0,1]\n\ts_endpgm\n'
                       '_Zplain:\n\tv_pk_mul_f32 v[22:23], v[24:25], v[22:23]\n\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] neg_lo:[0,1] neg_hi:[0,1]\n\ts_endpgm\n')
    found = audit.audit(str(crafted))
    assert sorted(f[0] for f in found) == ['_Zhigh', '_Zswapped']
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'audit_store_hazard.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(' 0 finding(s)') >= 9


def test_build_refuses_objects_that_contain_a_hazard(tmp_path):
    """build() audits the DEVICE CODE of the objects it has just produced (llvm-objdump of the offload bundle) before it
    links them -- the gate is on what ships, not on a second compile.  Negative test: crafted gfx950 objects, assembled
    here with the ROCm assembler, each holding one of the three patterns, must be found by audit_object and must make
    build.audit_objects raise; their guarded twins and the shipped objects must pass."""
    import subprocess
    from nsynth_wavenet_amd import build as wnbuild, hazard_audit
    clang = hazard_audit.llvm_tool('clang')

    def assemble(name, body):
        src = tmp_path / (name + '.s')
        src.write_text('\t.text\n\t.globl k\n\t.type k,@function\nk:\n' + body + '\ts_endpgm\n')
        obj = tmp_path / (name + '.o')
        subprocess.run([clang, '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', str(src), '-o', str(obj)],
                       check=True, capture_output=True)
        return str(obj)

    bad = {
        'store': '\tbuffer_store_dwordx4 v[48:51], v115, s[40:43], s75 offen\n\tv_max_f32_e64 v48, |v84|, |v85|\n',
        'mfma': '\tv_add_f32_e32 v87, v51, v88\n\tv_mov_b32_e32 v88, v97\n\tv_mfma_f32_16x16x32_f16 v[100:103], v[80:83], v[84:87], v[60:63]\n',
        'pk': '\tv_pk_fma_f32 v[14:15], v[14:15], v[86:87], v[82:83] op_sel:[0,1,0]\n',
    }
    good = {
        'store_ok': '\tbuffer_store_dwordx4 v[48:51], v115, s[40:43], s75 offen\n\ts_nop 1\n\tv_max_f32_e64 v48, |v84|, |v85|\n',
        'mfma_ok': '\tv_add_f32_e32 v87, v51, v88\n\ts_nop 1\n\tv_mfma_f32_16x16x32_f16 v[100:103], v[80:83], v[84:87], v[60:63]\n',
        'pk_ok': '\tv_pk_fma_f32 v[14:15], v[14:15], v[70:71], v[82:83] op_sel_hi:[1,0,1]\n',
    }
    for name, body in bad.items():
        obj = assemble(name, body)
        assert len(hazard_audit.audit_object(obj)) == 1, name
        with pytest.raises(RuntimeError, match='hazard audit failed'):
            wnbuild.audit_objects([obj], verbose=False)
    ok = [assemble(name, body) for name, body in good.items()]
    wnbuild.audit_objects(ok, verbose=False)
    # the objects of the shipped library (a hipcc offload bundle each): clean, and really disassembled
    shipped = [os.path.join(wnbuild.LIB_DIR, os.path.splitext(s)[0] + '.o') for s in wnbuild.SOURCES]
    wnbuild.audit_objects(shipped, verbose=False)
    lst = hazard_audit.disassemble_object(os.path.join(wnbuild.LIB_DIR, 'wn_iaf_g.o'), str(tmp_path))
    assert open(lst).read().count('v_mfma_f32_16x16x32_f16') > 1000
    # the gate FAILS CLOSED: an object from which no gfx950 kernel code can be extracted is an error, not "0 findings" --
    # a host object (what a failed bundle extraction used to fall back to), a device ELF without a kernel, garbage
    host_c = tmp_path / 'h.c'
    host_c.write_text('int f(int x) { return x + 1; }\n')
    host_o = tmp_path / 'host_only.o'
    subprocess.run(['gcc', '-c', str(host_c), '-o', str(host_o)], check=True, capture_output=True)
    with pytest.raises(RuntimeError, match='no gfx950 device image'):
        hazard_audit.audit_object(str(host_o))
    assert hazard_audit.audit_object(str(host_o), allow_no_kernels=True) == []          # (wn_host.o: no offload bundle inside)
    empty_s = tmp_path / 'empty.s'
    empty_s.write_text('\t.text\n\t.globl d\nd:\n\ts_nop 0\n')
    empty_o = tmp_path / 'empty.o'
    subprocess.run([clang, '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', str(empty_s), '-o', str(empty_o)],
                   check=True, capture_output=True)
    with pytest.raises(RuntimeError, match='no kernel code'):
        hazard_audit.audit_object(str(empty_o))
    junk = tmp_path / 'junk.o'
    junk.write_bytes(b'not an object file at all')
    with pytest.raises(RuntimeError):
        hazard_audit.audit_object(str(junk))


def test_mel_entry_points_validate_arguments_before_touching_a_device():
    """wn_mel_frames is the reference's frame count (1 + n // 200, librosa centred frames, hop 12.5 ms);
    wn_mel_spectrogram refuses null pointers and signals that numpy.pad(reflect) would refuse (<= 1024 samples)."""
    lib = _lib.load()
    assert [lib.wn_mel_frames(n) for n in (0, 199, 200, 76800, 154480)] == [1, 1, 2, 385, 773]
    assert lib.wn_mel_frames(-1) == -1
    buf = (ctypes.c_float * 2048)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.wn_mel_spectrogram(None, 1, 2048, p, None) == -22
    assert lib.wn_mel_spectrogram(p, 0, 2048, p, None) == -22
    assert lib.wn_mel_spectrogram(p, 1, 1024, p, None) == -22
    assert b'1024' in lib.wn_last_error(None)


def test_wn_config_struct_matches_header_layout():
    # 7 scalars + 2*4 + 4 scalars + 8 + 7 scalars + 8 reserved = 42 int32
    assert ctypes.sizeof(_lib.WnConfig) == 4 * (7 + 4 + 4 + 4 + 8 + 7 + 8)


def test_create_without_gpu_fails_loudly_or_succeeds_on_gpu():
    import torch
    hp = cfg.load_hparams(REFERENCE_STYLE_STUDENT)
    c = cfg.to_wn_config(hp)
    h = ctypes.c_void_p(0)
    rc = _lib.load().wn_create(ctypes.byref(c), ctypes.byref(h))
    if torch.cuda.is_available():
        assert rc == 0
        _lib.load().wn_destroy(h)
    else:
        assert rc == -5 and b'no HIP device' in _lib.load().wn_last_error(None)
        from nsynth_wavenet_amd.engine import Engine
        with pytest.raises(RuntimeError):
            Engine(hp)


def test_invalid_configs_are_rejected_by_the_library():
    lib = _lib.load()
    for patch, frag in (({'filter_length': 5}, b'filter_length'), ({'width': 47}, b'width must be even'),
                        ({'deconv_config': [[40, 12], [80, 20]]}, b'deconv layer'),
                        ({'num_stages': 12}, b'num_stages'), ({'num_stages': 2}, b'num_stages must be >= 3'),
                        # the generic layer kernel keeps 1.5 * width * 256 B of a tile in LDS: 418 would need 160.5 KB
                        ({'width': 418}, b'must fit the 160 KB'), ({'width': 1024}, b'must fit the 160 KB'),
                        ({'use_resize_conv': True, 'deconv_config': [[80, 10], [80, 20]]}, b'resize_conv layer')):
        d = dict(REFERENCE_STYLE_STUDENT)
        d.update(patch)
        c = cfg.to_wn_config(cfg.load_hparams(d))
        h = ctypes.c_void_p(0)
        assert lib.wn_create(ctypes.byref(c), ctypes.byref(h)) == -22
        assert frag in lib.wn_last_error(None)
    # shapes the MFMA kernels are not specialised for are served by the generic kernels, not refused: the config check
    # passes and only the missing GPU stops wn_create on this machine (-5), exactly as for the shipped shape
    ok_rc = lib.wn_create(ctypes.byref(cfg.to_wn_config(cfg.load_hparams(REFERENCE_STYLE_STUDENT))), ctypes.byref(ctypes.c_void_p(0)))
    for patch in ({'width': 48}, {'num_stages': 5}, {'width': 128, 'deconv_width': 128}, {'width': 416}):
        c = cfg.to_wn_config(cfg.load_hparams(dict(REFERENCE_STYLE_STUDENT, **patch)))
        h = ctypes.c_void_p(0)
        rc = lib.wn_create(ctypes.byref(c), ctypes.byref(h))
        assert rc == ok_rc and rc in (0, -5)
        if rc == 0:
            lib.wn_destroy(h)
    assert cfg.to_wn_config(cfg.load_hparams(dict(REFERENCE_STYLE_STUDENT, use_resize_conv=True))).use_resize_conv == 1
    c = cfg.to_wn_config(cfg.load_hparams(REFERENCE_STYLE_STUDENT))
    c.reserved[3] = 1
    assert lib.wn_create(ctypes.byref(c), ctypes.byref(ctypes.c_void_p(0))) == -22 and b'reserved' in lib.wn_last_error(None)
    with pytest.raises(ValueError):
        cfg.to_wn_config(cfg.load_hparams(dict(REFERENCE_STYLE_STUDENT, use_teacher_deconv=True)))


def test_per_class_defaults():
    ce = cfg.load_hparams(load_json('wavenet_ce.json'))
    assert cfg.model_kind(ce) == 'teacher'
    assert cfg.teacher_gate_width(ce) == 1024          # double_gate_width defaults to True
    c = cfg.to_wn_config(ce)
    assert (c.gate_width, c.out_width, c.use_mu_law, c.upsample_act) == (1024, 256, 1, _lib.ACT['tanh'])
    mol = cfg.load_hparams(load_json('wavenet_mol.json'))
    c = cfg.to_wn_config(mol)
    assert (c.gate_width, c.out_width, c.mol_mix, c.loss_type) == (512, 30, 10, _lib.LOSS['mol'])
    st = cfg.load_hparams(load_json('parallel_wavenet.json'))
    c = cfg.to_wn_config(st)
    assert (c.kind, c.share_deconv, c.n_flows, list(c.iaf_layers)[:4]) == (0, 1, 4, [10, 10, 10, 30])
    g = cfg.load_hparams(load_json('parallel_wavenet_gauss.json'))
    c = cfg.to_wn_config(g)
    assert c.share_deconv == 0 and c.loss_type == _lib.LOSS['gauss']   # no use_share_deconv key -> private stacks
    assert cfg.iaf_length(st, 773) == 154112 and cfg.iaf_length(st, 2) == 0 and cfg.frame_shift(st) == 200


def test_expected_variables_and_synthetic_weights_match_oracle_generator():
    from oracle import wavenet_np as O
    for name, kind in (('parallel_wavenet.json', 'student'), ('parallel_wavenet_gauss.json', 'student'),
                       ('wavenet_mol.json', 'teacher')):
        d = load_json(name)
        if kind == 'teacher':
            d.update(width=64, skip_width=64, num_layers=3)
        hp = cfg.load_hparams(d)
        ev = wts.expected_variables(hp)
        w = wts.synthetic_weights(hp, seed=7, init='tf')
        wo = O.synth_weights(O.HP(d), kind, seed=7, init='tf')
        assert sorted(w) == sorted(wo) == sorted(n for n, _ in ev)
        for n, shape in ev:
            assert w[n].shape == tuple(shape) == wo[n].shape
    hp = cfg.load_hparams(load_json('parallel_wavenet.json'))
    w = wts.synthetic_weights(hp)
    assert w['iaf_1/out2_scale/biases'][0] == np.float32(-0.3) and w['iaf_1/out1/biases'].sum() == 0
    assert abs(w['iaf_4/dilated_conv_30/W'].std() - 0.05) < 2e-3
    wn = wts.expected_variables(cfg.load_hparams(dict(load_json('parallel_wavenet.json'), use_weight_norm=True)))
    names = [n for n, _ in wn]
    assert 'iaf_1/start_conv/W_V' in names and 'iaf_share/trans_conv_2/kernel_g' in names


def test_checkpoint_round_trip_with_ema_keys(tmp_path):
    d = load_json('parallel_wavenet.json')
    d['num_iaf_layers'] = [1, 1]
    hp = cfg.load_hparams(d)
    w = wts.synthetic_weights(hp, seed=3)
    p = wts.save_checkpoint(str(tmp_path / 'model.ckpt-12'), w, hp)
    blob = np.load(p)
    assert all(k.endswith(wts.EMA) for k in blob.files)                       # fastgen.py:12-14
    back = wts.load_checkpoint(str(tmp_path / 'model.ckpt-12'), hp)
    assert all(np.array_equal(back[k], w[k]) for k in w)
    wts.save_checkpoint(str(tmp_path / 'model.ckpt-300'), w, hp)
    assert wts.latest_checkpoint(str(tmp_path)).endswith('model.ckpt-300.npz')
    (tmp_path / 'checkpoint').write_text('model_checkpoint_path: "model.ckpt-12"\n')
    assert wts.latest_checkpoint(str(tmp_path)).endswith('model.ckpt-12')
    # teacher-owned deconv variables are stored under raw names (parallelgen.py:31-39)
    d2 = dict(d, use_share_deconv=False, use_teacher_deconv=True)
    hp2 = cfg.load_hparams(d2)
    p2 = wts.save_checkpoint(str(tmp_path / 'td'), w, hp2)
    keys = np.load(p2).files
    assert 'iaf_share/trans_conv_1/kernel' in keys and 'iaf_1/start_conv/W' + wts.EMA in keys
    assert np.array_equal(wts.load_checkpoint(p2, hp2)['iaf_share/trans_conv_1/kernel'],
                          w['iaf_share/trans_conv_1/kernel'])
    bad = {k: v for k, v in w.items() if 'out2_mean' not in k}
    p3 = wts.save_checkpoint(str(tmp_path / 'bad'), bad, hp)
    with pytest.raises(KeyError):
        wts.load_checkpoint(p3, hp)


def test_cli_surface_and_source_discovery(tmp_path):
    from nsynth_wavenet_amd import cli
    a = cli.build_parser('x').parse_args(['--ckpt_dir', 'c', '--source_path', 's', '--save_path', 'o'])
    assert (a.sample_length, a.batch_size, a.npy_only, a.log, a.gpu_id) == (-1, 1, False, 'INFO', '0')
    for n in ('b.wav', 'a.wav', 'c.npy', 'notes.txt'):
        (tmp_path / n).write_bytes(b'')
    assert [os.path.basename(f) for f in cli.list_sources(str(tmp_path), False)] == ['a.wav', 'b.wav']
    assert [os.path.basename(f) for f in cli.list_sources(str(tmp_path), True)] == ['c.npy']
    assert cli.list_sources(str(tmp_path / 'a.wav'), False) == [str(tmp_path / 'a.wav')]
    assert cli.list_sources(str(tmp_path / 'notes.txt'), False) == []
    empty = tmp_path / 'e'
    empty.mkdir()
    (empty / 'x.txt').write_bytes(b'')
    with pytest.raises(RuntimeError):
        cli.list_sources(str(empty), False)
    ck = tmp_path / 'ck'
    ck.mkdir()
    with pytest.raises(AssertionError):
        cli.resolve_model(str(ck))
    hp = cfg.load_hparams(dict(load_json('parallel_wavenet.json'), num_iaf_layers=[1]))
    wts.save_checkpoint(str(ck / 'model.ckpt-5'), wts.synthetic_weights(hp), hp)
    (ck / 'a.json').write_text(json.dumps(vars(hp)))
    hp_back, path = cli.resolve_model(str(ck))
    assert isinstance(hp_back, Namespace) and path.endswith('model.ckpt-5.npz')
    (ck / 'b.json').write_text('{}')
    with pytest.raises(AssertionError):
        cli.resolve_model(str(ck))


def test_load_batch_pads_and_save_batch_writes_float32_wav(tmp_path):
    from scipy.io import wavfile
    from nsynth_wavenet_amd.wavenet import fastgen
    a = (np.sin(np.arange(3000) / 10.) * 20000).astype(np.int16)
    b = (np.sin(np.arange(1800) / 7.) * 10000).astype(np.int16)
    wavfile.write(str(tmp_path / 'a.wav'), 16000, a)
    wavfile.write(str(tmp_path / 'b.wav'), 16000, b)
    batch = fastgen.load_batch([str(tmp_path / 'a.wav'), str(tmp_path / 'b.wav')], sample_length=-1)
    assert batch.shape == (2, 3000) and batch.dtype == np.float32
    assert np.all(batch[1, 1800:] == 0) and abs(batch[0, 5] - a[5] / 32768.0) < 1e-7
    assert fastgen.load_batch([str(tmp_path / 'a.wav')], sample_length=1000).shape == (1, 1000)
    np.save(str(tmp_path / 'm1.npy'), np.ones([5, 80], np.float32))
    np.save(str(tmp_path / 'm2.npy'), np.ones([3, 80], np.float32))
    mb = fastgen.load_batch([str(tmp_path / 'm1.npy'), str(tmp_path / 'm2.npy')])
    assert mb.shape == (2, 5, 80) and mb[1, 3:].sum() == 0
    out = np.linspace(-1, 1, 64, dtype=np.float32)[None]
    fastgen.save_batch(out, [str(tmp_path / 'gen_x.wav')])
    sr, back = wavfile.read(str(tmp_path / 'gen_x.wav'))
    assert sr == 16000 and back.dtype == np.float32 and np.array_equal(back, out[0])


def test_mel_featuriser_shape_range_and_tone():
    from nsynth_wavenet_amd.auxilaries import mel_extractor as M
    y = (0.1 * np.random.RandomState(0).standard_normal(154480)).astype(np.float32)
    m = M.melspectrogram(y)
    assert m.shape == (773, 80) and m.dtype == np.float32           # K4: 1 + 154480 // 200 frames
    assert m.min() >= 0 and m.max() <= 1
    assert np.allclose(M.melspectrogram(np.zeros(4000, np.float32)), 40.0 / 140.0, atol=1e-6)   # floor: 1e-5 -> -100 dB
    fb = M.mel_filterbank()
    assert fb.shape == (80, 1025) and np.all(fb >= 0) and np.all(fb.sum(axis=1) > 0)
    freqs = np.linspace(0, 8000, 1025)
    assert fb[:, freqs < 125].sum() == 0 and fb[:, freqs > 7600].sum() == 0
    t = np.arange(16000) / 16000.0
    lo = M.melspectrogram((0.5 * np.sin(2 * np.pi * 500 * t)).astype(np.float32))[40].argmax()
    hi = M.melspectrogram((0.5 * np.sin(2 * np.pi * 4000 * t)).astype(np.float32))[40].argmax()
    assert lo < hi
    assert M.batch_melspectrogram(np.stack([y[:4000], y[4000:8000]])).shape == (2, 21, 80)


def test_host_codecs_match_oracle():
    from oracle import wavenet_np as O
    from nsynth_wavenet_amd.auxilaries import utils
    q = np.arange(-128, 128)
    assert np.array_equal(utils.inv_mu_law_numpy(q), O.inv_mu_law(q))
    x = np.random.RandomState(0).uniform(-1, 1, 1000).astype(np.float32)
    assert np.array_equal(utils.mu_law_numpy(x), O.mu_law(x))
    # the reference's numpy helper TRUNCATES (auxilaries/utils.py:162-164: astype(np.int32) after the scaling), its
    # TF twin floors (:153-154): equal on the grid and for x >= 0, one step apart for negative off-grid values
    got = utils.cast_quantize_numpy(x, 65536)
    assert np.array_equal(got, np.trunc(x.astype(np.float64) * 32768).astype(np.int32))
    fl = O.cast_quantize(x, 65536)
    assert np.array_equal(got[x >= 0], fl[x >= 0]) and np.all((got - fl)[x < 0] >= 0) and np.all((got - fl) <= 1)
    grid = np.arange(-32768, 32768, 257, dtype=np.float32) / 32768
    assert np.array_equal(utils.cast_quantize_numpy(grid, 65536), O.cast_quantize(grid, 65536))


def test_run_all_eval_staging_and_sweep(tmp_path, monkeypatch):
    """Local mirror of the reference's run_all_eval.py:36-140: newest checkpoint + config staged with
    a `checkpoint` state file that latest_checkpoint honours, outputs under waves/<exp>-iter_N,
    staging directory removed, remote hosts rejected."""
    import run_all_eval as rae
    assert rae.get_last_model_prefix(['model.ckpt-5.index', 'model.ckpt-120.index', 'events.x', 'a.json']) == \
        ('model.ckpt-120', 120)
    assert rae.get_last_model_prefix(['model.ckpt-7.npz']) == ('model.ckpt-7', 7)
    with pytest.raises(FileNotFoundError):
        rae.get_last_model_prefix(['a.json'])
    exp = tmp_path / 'exp_a'
    exp.mkdir()
    for n in ('model.ckpt-10.npz', 'model.ckpt-200.npz', 'events.out.tfevents.1', 'pwn.json'):
        (exp / n).write_bytes(b'x')
    target = tmp_path / 'out-01_01_00'
    model_dir, wave_dir, it = rae.stage_experiment(str(exp), str(target))
    assert it == 200 and sorted(os.listdir(model_dir)) == ['checkpoint', 'model.ckpt-200.npz', 'pwn.json']
    assert wave_dir.endswith(os.path.join('waves', 'exp_a-iter_200')) and os.path.isdir(wave_dir)
    assert os.path.exists(os.path.join(str(target), 'exp_a', 'events.out.tfevents.1'))
    assert wts.latest_checkpoint(model_dir) == os.path.join(model_dir, 'model.ckpt-200')
    sweep = tmp_path / 'sweep.json'
    calls, real_syn_wave = [], rae.syn_wave
    monkeypatch.setattr(rae, 'syn_wave', lambda *a: calls.append(a) or 0)
    sweep.write_text(json.dumps({'hosts': [None], 'users': [''], 'passwords': [''], 'exp_dirs': [str(exp)],
                                 'eval_scripts': ['eval_parallel_wavenet.py']}))
    out, failed = rae.run_all(str(sweep), str(tmp_path), str(tmp_path / 'out'), '0', stamp='01_01_00')
    assert failed == [] and len(calls) == 1 and calls[0][0] == 'eval_parallel_wavenet.py'
    assert calls[0][1] == os.path.join(out, 'exp_a-model') and not os.path.exists(calls[0][1])
    sweep.write_text(json.dumps({'hosts': ['10.0.0.5'], 'exp_dirs': [str(exp)],
                                 'eval_scripts': ['eval_wavenet.py']}))
    with pytest.raises(ValueError):
        rae.run_all(str(sweep), str(tmp_path), str(tmp_path / 'out'), '0')
    with pytest.raises(ValueError):
        real_syn_wave('rm_rf.py', 'a', 'b', 'c', '0')


def test_precision_names_map_to_config_fields():
    hp = cfg.load_hparams(REFERENCE_STYLE_STUDENT)
    for name, (prec, cond) in {'f16x3': (0, 0), 'f16x3-fused': (0, 1), 'f16x3-hoisted': (0, 2), 'f32': (1, 0), 'f32-fused': (1, 1), 'f32-hoisted': (1, 2)}.items():
        c = cfg.to_wn_config(hp, precision=name)
        assert (c.precision, c.cond_mode, c.use_resize_conv) == (prec, cond, 0) and list(c.reserved) == [0] * 5
    with pytest.raises(ValueError):
        cfg.to_wn_config(hp, precision='bf16')
    lib = _lib.load()
    c = cfg.to_wn_config(hp)
    c.cond_mode = 7
    assert lib.wn_create(ctypes.byref(c), ctypes.byref(ctypes.c_void_p(0))) == -22
    assert b'conditioning mode' in lib.wn_last_error(None)
