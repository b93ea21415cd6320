"""Known-answer vectors from the REFERENCE'S OWN numpy codecs, executed in the build container.

Run in the build container only (needs /root/reference; nothing of it travels -- only the .npz below):
    python tests/golden/make_ref_codec.py
Writes tests/golden/ref_codec.npz.

What this pins, and what it does not.  /root/reference/auxilaries/utils.py holds four pure-numpy functions beside
their TensorFlow twins: mu_law_numpy (:90-105), inv_mu_law_numpy (:125-139), cast_quantize_numpy (:162-164),
inv_cast_quantize_numpy (:167-169).  They are EXECUTED here -- the reference's code, not a restatement -- on every code
and on seeded random samples; the arrays are the expected outputs.  The module's import line needs `tensorflow`,
`librosa` and `wavenet.masked`, none of which exist in this image: they are replaced by inert placeholder modules for
the duration of the import only (module-level statements such as `slim = tf.contrib.slim` touch attributes; no
function of a placeholder is ever called -- the placeholder raises if one is).  This pins rows a8 / a11 of SURVEY.md
section 8 on the numpy side only; the float (IAF / AR) path stays "parity unpinned" (no TensorFlow here).

NumPy version caveat, found by running this: mu_law_numpy divides a float32 array by `np.log(1 + mu)`, a float64
SCALAR.  Under the NumPy 1.x of the reference's era (value-based casting) the expression stays float32; under this
image's NumPy 2.2 (NEP 50) it becomes float64, and floor() then lands one code lower where the float32 result sat
exactly on an integer (x = -1.0 gives -129 instead of -128) or crosses one by rounding.  Both are recorded:
`mu_law_*` is the function called as written (NumPy 2.2 arithmetic), `mu_law_*_f32` the same reference function called
with mu=np.float32(255), which keeps every operand float32 -- the arithmetic the reference ran when it was written.

Note the reference's own inconsistency, kept as data: cast_quantize_numpy TRUNCATES (`astype(np.int32)`), its TF twin
`cast_quantize` FLOORS (:153-154) and that is what the generation path runs (parallel_wavenet.py:347-359).  The two
agree for x >= 0 and on grid points; `cast_trunc_*` below is what the reference's numpy function returns.
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_codec.npz')
N_RANDOM = 1 << 17
SEED = 20240607


class _Inert(types.ModuleType):
    """Attribute access yields another inert object (import-time statements like `tf.contrib.slim`); calling raises."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Inert(self.__name__ + '.' + name)

    def __call__(self, *a, **k):
        raise RuntimeError('placeholder for an absent dependency was CALLED: ' + self.__name__)


def import_reference_utils():
    saved = {k: sys.modules.get(k) for k in ('tensorflow', 'librosa', 'wavenet', 'wavenet.masked', 'auxilaries',
                                              'auxilaries.utils')}
    for name in ('tensorflow', 'librosa', 'wavenet', 'wavenet.masked'):
        sys.modules[name] = _Inert(name)
    sys.path.insert(0, REF)
    try:
        for k in ('auxilaries', 'auxilaries.utils'):
            sys.modules.pop(k, None)
        import importlib
        mod = importlib.import_module('auxilaries.utils')
        assert os.path.realpath(mod.__file__).startswith(REF), mod.__file__
        return mod
    finally:
        sys.path.remove(REF)
        for k, v in saved.items():
            if k.startswith('auxilaries'):
                continue                      # keep the imported reference module alive for the caller only
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def inputs():
    """The seeded inputs; tests regenerate them with the same calls (legacy RandomState is stable across numpy builds)."""
    rs = np.random.RandomState(SEED)
    x = rs.uniform(-1.0, 1.0, N_RANDOM).astype(np.float32)
    edge = np.array([-1.0, -0.999969482421875, -0.5, -2.0 ** -15, -1e-9, 0.0, 1e-9, 2.0 ** -15, 0.5,
                     1.0 - 2.0 ** -15, 1.0 - 2.0 ** -7, 0.9999999, 1.0], np.float32)
    return np.concatenate([edge, x])


def main():
    U = import_reference_utils()
    x = inputs()
    grid16 = (np.arange(-32768, 32768, dtype=np.float32) / np.float32(32768.0))
    out = {
        'seed': np.int64(SEED), 'n_random': np.int64(N_RANDOM), 'n_edge': np.int64(x.size - N_RANDOM),
        # utils.py:90-105 on seeded samples and on every 16-bit grid point (floor(out * 128): integers, stored as int16)
        'mu_law_x': U.mu_law_numpy(x).astype(np.int16),
        'mu_law_grid16': U.mu_law_numpy(grid16).astype(np.int16),
        'mu_law_x_f32': U.mu_law_numpy(x, mu=np.float32(255)).astype(np.int16),
        'mu_law_grid16_f32': U.mu_law_numpy(grid16, mu=np.float32(255)).astype(np.int16),
        # utils.py:125-139 on every code
        'inv_mu_law_codes': U.inv_mu_law_numpy(np.arange(-128, 128)).astype(np.float32),
        # utils.py:162-164 (truncating numpy twin) and :167-169 on every code
        'cast_trunc_x_65536': U.cast_quantize_numpy(x, 65536),
        'cast_trunc_x_256': U.cast_quantize_numpy(x, 256),
        'cast_trunc_grid16': U.cast_quantize_numpy(grid16, 65536),
        'inv_cast_codes_65536': U.inv_cast_quantize_numpy(np.arange(-32768, 32768, dtype=np.int32), 65536),
        'inv_cast_codes_256': U.inv_cast_quantize_numpy(np.arange(-128, 128, dtype=np.int32), 256),
    }
    # K7 round trips, evaluated with the reference's functions themselves
    codes = np.arange(-128, 128)
    out['roundtrip_mu_law_ok'] = np.array(np.array_equal(U.mu_law_numpy(U.inv_mu_law_numpy(codes)), codes.astype(np.float32)))
    q = np.arange(-32768, 32768, dtype=np.int32)
    out['roundtrip_cast_ok'] = np.array(np.array_equal(U.cast_quantize_numpy(U.inv_cast_quantize_numpy(q, 65536), 65536), q))
    assert U.mu_law_numpy(x, mu=np.float32(255)).dtype == np.float32 and U.mu_law_numpy(x).dtype == np.float64
    out['numpy_version'] = np.array(np.__version__)
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, {k: (v.shape, str(v.dtype)) for k, v in out.items() if hasattr(v, 'shape') and v.ndim})
    print('round trips (reference functions):', bool(out['roundtrip_mu_law_ok']), bool(out['roundtrip_cast_ok']))


if __name__ == '__main__':
    main()
