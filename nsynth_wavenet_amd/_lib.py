"""ctypes binding of libwnhip.so (include/wnhip.h).  No fallback: if the HIP
library is missing or fails to load, importing the engine raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# WN_LIB_PATH: another build of the same library (A/B measurements of kernel variants on one box); never a fallback
LIB_PATH = os.environ.get('WN_LIB_PATH') or os.path.join(_HERE, 'lib', 'libwnhip.so')

WN_MAX_DECONV = 4
WN_MAX_FLOWS = 8
KIND_STUDENT, KIND_TEACHER = 0, 1
LOSS = {'ce': 0, 'mol': 1, 'gauss': 2, 'logistic': 3}
ACT = {'tanh': 0, 'relu': 1, 'leaky_relu': 2}
ERRNAMES = {-22: 'WN_EINVAL', -2: 'WN_ENOENT', -12: 'WN_ENOMEM', -5: 'WN_EIO', -1: 'WN_ESTATE', -34: 'WN_ERANGE'}
WN_ERANGE = -34
FORM_DEFAULT, FORM_F16X3, FORM_F32, FORM_F16X3_FUSED = -1, 0, 1, 2

# every symbol include/wnhip.h declares
SYMBOLS = ['wn_abi_version', 'wn_create', 'wn_set_weight', 'wn_finalize', 'wn_iaf_length',
           'wn_ar_length', 'wn_workspace_bytes', 'wn_deconv', 'wn_iaf_generate', 'wn_iaf_generate_form',
           'wn_iaf_workspace_bytes_form', 'wn_iaf_range_status', 'wn_iaf_range_reset',
           'wn_iaf_range_status_since_reset', 'wn_clip_quant',
           'wn_ar_n_rand', 'wn_ar_state_bytes', 'wn_ar_reset', 'wn_ar_step', 'wn_ar_generate', 'wn_ar_set_graph', 'wn_ar_cond_vars', 'wn_ar_cond_vars_floats',
           'wn_iaf_cond_hoisted', 'wn_iaf_layer_groups', 'wn_iaf_set_groups', 'wn_teacher_workspace_bytes', 'wn_teacher_forward', 'wn_teacher_log_prob', 'wn_profile_begin', 'wn_profile_pause', 'wn_profile_end', 'wn_profile_parts_begin', 'wn_profile_parts_end', 'wn_profile_parts_only', 'wn_mel_frames', 'wn_mel_spectrogram', 'wn_last_error', 'wn_destroy', 'wn_crc32c']


class WnConfig(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('n_mel', ctypes.c_int32), ('width', ctypes.c_int32),
                ('skip_width', ctypes.c_int32), ('gate_width', ctypes.c_int32),
                ('deconv_width', ctypes.c_int32), ('n_deconv', ctypes.c_int32),
                ('deconv_filter', ctypes.c_int32 * WN_MAX_DECONV),
                ('deconv_stride', ctypes.c_int32 * WN_MAX_DECONV),
                ('filter_length', ctypes.c_int32), ('num_stages', ctypes.c_int32),
                ('num_layers', ctypes.c_int32), ('n_flows', ctypes.c_int32),
                ('iaf_layers', ctypes.c_int32 * WN_MAX_FLOWS), ('use_mu_law', ctypes.c_int32),
                ('loss_type', ctypes.c_int32), ('mol_mix', ctypes.c_int32),
                ('out_width', ctypes.c_int32), ('share_deconv', ctypes.c_int32),
                ('use_weight_norm', ctypes.c_int32), ('upsample_act', ctypes.c_int32),
                ('precision', ctypes.c_int32), ('cond_mode', ctypes.c_int32),
                ('use_resize_conv', ctypes.c_int32), ('reserved', ctypes.c_int32 * 5)]


_lib = None


def load():
    """Load libwnhip.so once; raise (never fall back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libwnhip.so not found at {}: build it with `python -m nsynth_wavenet_amd.build` '
            '(there is no CPU fallback for the generation path)'.format(LIB_PATH))
    # PyTorch-ROCm ships its own HIP runtime; load it first so that libwnhip.so binds to the same
    # libamdhip64 instance torch uses (two runtimes in one process do not share devices or streams)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    vp, i32, i64, u64, sz = c.c_void_p, c.c_int, c.c_int64, c.c_uint64, c.c_size_t
    lib.wn_abi_version.restype = i32
    lib.wn_create.argtypes = [c.POINTER(WnConfig), c.POINTER(vp)]
    lib.wn_set_weight.argtypes = [vp, c.c_char_p, vp, c.POINTER(i64), i32]
    lib.wn_finalize.argtypes = [vp]
    lib.wn_iaf_length.argtypes = [vp, i32]
    lib.wn_iaf_length.restype = i64
    lib.wn_ar_length.argtypes = [vp, i32]
    lib.wn_ar_length.restype = i64
    lib.wn_workspace_bytes.argtypes = [vp, i32, i32]
    lib.wn_workspace_bytes.restype = sz
    lib.wn_deconv.argtypes = [vp, c.c_char_p, vp, i32, i32, vp, vp, sz, vp]
    lib.wn_iaf_generate.argtypes = [vp, vp, i32, i32, vp, u64, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.wn_iaf_generate_form.argtypes = [vp, i32, vp, i32, i32, vp, u64, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.wn_iaf_workspace_bytes_form.argtypes = [vp, i32, i32, i32]
    lib.wn_iaf_workspace_bytes_form.restype = sz
    lib.wn_iaf_range_status.argtypes = [vp, vp, vp]
    lib.wn_iaf_range_reset.argtypes = [vp, vp, vp]
    lib.wn_iaf_range_status_since_reset.argtypes = [vp, vp, vp]
    lib.wn_clip_quant.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.wn_ar_n_rand.argtypes = [vp]
    lib.wn_ar_state_bytes.argtypes = [vp, i32]
    lib.wn_ar_state_bytes.restype = sz
    lib.wn_ar_reset.argtypes = [vp, vp, i32, vp]
    lib.wn_ar_step.argtypes = [vp, vp, i32, vp, vp, vp, u64, vp, vp, vp]
    lib.wn_ar_generate.argtypes = [vp, vp, i32, i32, vp, u64, vp, vp, vp, vp, vp, sz, vp]
    lib.wn_ar_set_graph.argtypes = [vp, i32]
    lib.wn_ar_cond_vars_floats.argtypes = [vp, i32, i32]
    lib.wn_ar_cond_vars_floats.restype = sz
    lib.wn_ar_cond_vars.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.wn_iaf_cond_hoisted.argtypes = [vp, i32, i32]
    lib.wn_iaf_layer_groups.argtypes = [vp, i32, i32]
    lib.wn_iaf_set_groups.argtypes = [vp, i32]
    lib.wn_teacher_workspace_bytes.argtypes = [vp, i32, i32, i64]
    lib.wn_teacher_workspace_bytes.restype = sz
    lib.wn_teacher_forward.argtypes = [vp, vp, vp, i32, i32, i64, vp, vp, sz, vp]
    lib.wn_teacher_log_prob.argtypes = [vp, vp, vp, i32, i64, vp, vp]
    lib.wn_profile_begin.argtypes = [vp]
    lib.wn_profile_pause.argtypes = [vp, c.c_int]
    lib.wn_profile_end.argtypes = [vp, c.POINTER(c.c_double), c.POINTER(i64)]
    lib.wn_profile_parts_begin.argtypes = [vp]
    lib.wn_profile_parts_only.argtypes = [vp, i32]
    lib.wn_profile_parts_end.argtypes = [vp, c.POINTER(c.c_double), c.POINTER(i64)]
    lib.wn_mel_frames.argtypes = [i64]
    lib.wn_mel_frames.restype = i64
    lib.wn_mel_spectrogram.argtypes = [vp, c.c_int, i64, vp, vp]
    lib.wn_last_error.argtypes = [vp]
    lib.wn_last_error.restype = c.c_char_p
    lib.wn_destroy.argtypes = [vp]
    lib.wn_destroy.restype = None
    lib.wn_crc32c.argtypes = [c.c_char_p, sz, c.c_uint32]
    lib.wn_crc32c.restype = c.c_uint32
    for s in SYMBOLS:
        getattr(lib, s)
    if lib.wn_abi_version() != 1:
        raise RuntimeError('libwnhip.so ABI version mismatch')
    _lib = lib
    return lib


def check(rc, handle=None):
    """Map a negative return code to the exception the reference would raise
    (its asserts -> ValueError for shape/config, RuntimeError for runtime faults)."""
    if rc == 0:
        return
    msg = load().wn_last_error(handle)
    msg = msg.decode() if msg else ''
    text = '{}: {}'.format(ERRNAMES.get(rc, str(rc)), msg)
    if rc in (-22,):
        raise ValueError(text)
    if rc == -2:
        raise KeyError(text)
    if rc == -12:
        raise MemoryError(text)
    raise RuntimeError(text)
