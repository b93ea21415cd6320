"""Builds libwnhip.so (the gfx950 HIP engine) in-tree with hipcc.

    python -m nsynth_wavenet_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The shared object is written to
nsynth_wavenet_amd/lib/libwnhip.so so that it travels with the source tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB_DIR = os.path.join(HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, os.environ.get('WN_LIB_NAME', 'libwnhip.so'))     # WN_LIB_NAME: variant builds for A/B runs
# Code-generation flags of every kernel source.  -fno-slp-vectorize: the SLP vectorizer pairs scalar fp32 operations into
# v_pk_*_f32 and folds "both lanes take the same register" into operand selects; the form that takes the HIGH register of
# src1 for the low lane is not reliable on gfx950 next to matrix instructions (tests/test_gpu_hazards.py,
# scripts/ubench/pk_opsel.hip, profiles/r04_pk_opsel_hazard.txt).  Without the pass the compiler emits no packed-fp32
# instruction with selects at all (scripts/audit_store_hazard.py checks it), and the kernels are no slower (measured
# 58.4-59.3 against 60.8-61.1 us per group launch on one box).
CODEGEN_FLAGS = ['-O3', '-std=c++17', '-fno-slp-vectorize']
# The C ABI of include/wnhip.h (WN_API = default visibility) is the library's whole dynamic symbol table: everything
# else -- the C++ helpers shared by the translation units, STL instantiations -- is local to it.
VISIBILITY_FLAGS = ['-fvisibility=hidden', '-fvisibility-inlines-hidden']
SOURCES = ['wn_host.cpp', 'wn_deconv.hip', 'wn_iaf.hip', 'wn_iaf_h.hip', 'wn_iaf_c.hip', 'wn_iaf_g.hip', 'wn_iaf_f.hip', 'wn_iaf_x.hip', 'wn_ar.hip', 'wn_teacher.hip', 'wn_mel.hip']
HEADERS = ['wn_internal.h', 'wn_codec.h', 'wn_pack_h.h', 'wn_mfma_h.h', 'wn_iaf_c.h', os.path.join(ROOT, 'include', 'wnhip.h')]


def find_hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


STAMP_PATH = os.path.splitext(LIB_PATH)[0] + '.sha256'


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + \
           [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]


def source_hash():
    """sha256 over the kernel / host sources and headers the library is built from (names + contents,
    plus the extra compiler flags).  Stored next to the built library; bench.py stamps it into the PMC
    summaries so that a profile of other sources is never replayed as a measurement."""
    import hashlib
    hh = hashlib.sha256()
    for d in sorted(_deps(), key=os.path.basename):
        hh.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            hh.update(f.read())
    hh.update(' '.join(CODEGEN_FLAGS + VISIBILITY_FLAGS).encode())
    hh.update(os.environ.get('WN_EXTRA_FLAGS', '').encode())
    return hh.hexdigest()


def _stale():
    """The library is current iff it exists and was built from exactly these sources (content hash, not
    mtimes: a checkout or an rsync does not preserve them)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(STAMP_PATH):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_hash()


def audit_objects(objs, verbose=True):
    """Hazard audit of built objects (device code); raises RuntimeError listing the findings."""
    from . import hazard_audit
    findings = []
    for o in objs:
        bad = hazard_audit.audit_object(o, allow_no_kernels=os.path.basename(o).startswith('wn_host'))
        if verbose:
            print('hazard audit %-18s %d finding(s)' % (os.path.basename(o), len(bad)), flush=True)
        findings += [(os.path.basename(o),) + tuple(b) for b in bad]
    if findings:
        raise RuntimeError('gfx950 hazard audit failed, library NOT linked:\n' + '\n'.join(
            '  {}: {}\n      {}: {}\n      {}: {}'.format(o, k, ln, st, ln2, w) for o, k, ln, st, ln2, w in findings))


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared object."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs = []
    hipcc = find_hipcc()
    common = ['--offload-arch=gfx950'] + CODEGEN_FLAGS + VISIBILITY_FLAGS + ['-fPIC', '-Wall',
              '-Wno-unused-function', '-I', os.path.join(ROOT, 'include'), '-I', CSRC] + \
        os.environ.get('WN_EXTRA_FLAGS', '').split()
    procs = []
    for s in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(s)[0] + ('' if LIB_PATH.endswith('libwnhip.so') else '_v') + '.o')
        cmd = [hipcc] + common + (['-x', 'hip'] if s.endswith('.cpp') else []) + \
              ['-c', os.path.join(CSRC, s), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for s, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors='replace'))
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on {}:\n{}'.format(s, out.decode(errors='replace')))
    # Gate: the device code of every object just produced is disassembled and checked for the three gfx950 hazards hipcc
    # does not (or cannot) guard -- a wide store whose data a VALU instruction overwrites within two slots, an MFMA or DPP
    # read of a VGPR less than two wait states behind the VALU instruction that wrote it, packed fp32 with a high-for-low
    # src1 select (hazard_audit.py).  The kernels are exact only as long as the compiler's output keeps
    # clear of them, so a compiler that reintroduces one must not produce a library.
    audit_objects(objs, verbose)
    # ... and what -fvisibility=hidden cannot reach (libstdc++'s template instantiations carry their own default-visibility
    # attribute, hipcc adds one __hip_cuid_* per translation unit) is made local by a linker version script
    vers = os.path.join(LIB_DIR, 'libwnhip.map')
    with open(vers, 'w') as f:
        f.write('{ global: wn_*; local: *; };\n')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + VISIBILITY_FLAGS + ['-Wl,--version-script=' + vers] + objs + ['-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP_PATH, 'w') as f:
        f.write(source_hash() + '\n')
    return LIB_PATH


# MEASUREMENT builds for bench.py's labelled `roofline_f16x2` extra (never loaded by the product: _lib.py loads libwnhip.so):
# the split-fp16 contraction with TWO terms per product (wh.xh + wl.xh: the activations' lo halves are not multiplied) in the
# conditioning GEMM only (-DWN_F16X2=1) and in the conditioning GEMM + residual stack + heads (-DWN_F16X2=3).  Narrower than
# the reference's fp32 -- it prices the error budget the contract leaves.  Only the sources that hold an mfma3 site of the
# student path are recompiled; the other objects are the shipped library's.
F16X2_VARIANTS = {'libwnhip_f16x2c.so': ('1', ['wn_iaf_c.hip']),
                  'libwnhip_f16x2.so': ('3', ['wn_iaf_c.hip', 'wn_iaf_g.hip', 'wn_iaf_h.hip'])}


def build_f16x2(force=False, verbose=True):
    """Build the two f16x2 measurement libraries next to libwnhip.so (which must exist).  Returns their paths."""
    out = []
    hipcc = find_hipcc()
    have = source_hash()
    for name, (bits, srcs) in F16X2_VARIANTS.items():
        path = os.path.join(LIB_DIR, name)
        stamp = os.path.splitext(path)[0] + '.sha256'
        if not force and os.path.exists(path) and os.path.exists(stamp) and open(stamp).read().strip() == have + ':' + bits:
            out.append(path)
            continue
        common = ['--offload-arch=gfx950'] + CODEGEN_FLAGS + VISIBILITY_FLAGS + ['-fPIC', '-Wall', '-Wno-unused-function',
                  '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-DWN_F16X2=' + bits]
        objs, procs = [], []
        for s in SOURCES:
            base = os.path.splitext(s)[0]
            if s in srcs:
                obj = os.path.join(LIB_DIR, base + '_x2_' + bits + '.o')
                cmd = [hipcc] + common + ['-c', os.path.join(CSRC, s), '-o', obj]
                procs.append((s, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            else:
                obj = os.path.join(LIB_DIR, base + '.o')
            objs.append(obj)
        for s, obj, p_ in procs:
            o, _ = p_.communicate()
            if p_.returncode != 0:
                raise RuntimeError('hipcc failed on {} (f16x2 build):\n{}'.format(s, o.decode(errors='replace')))
        audit_objects([o for _, o, _ in procs], verbose)
        vers = os.path.join(LIB_DIR, 'libwnhip.map')
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + VISIBILITY_FLAGS +
                              ['-Wl,--version-script=' + vers] + objs + ['-o', path])
        with open(stamp, 'w') as f:
            f.write(have + ':' + bits + '\n')
        if verbose:
            print('built', path, flush=True)
        out.append(path)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    if '--f16x2' in sys.argv:
        print(build_f16x2(force='--force' in sys.argv))
