#!/usr/bin/env python3
"""Command line of nsynth_wavenet_amd/hazard_audit.py (the gfx950 hazard audit of the compiled kernels; the rules and their
measurements are described there).

    python scripts/audit_store_hazard.py            # compiles csrc/*.hip with -save-temps into a temp dir, audits the .s
    python scripts/audit_store_hazard.py --objects  # audits the device code of nsynth_wavenet_amd/lib/*.o (what build() does)
"""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsynth_wavenet_amd.hazard_audit import audit, audit_object, disassemble_object, report, regs, WINDOW  # noqa: E402,F401


def main():
    from nsynth_wavenet_amd import build as B
    if '--objects' in sys.argv:
        total = 0
        for s in B.SOURCES:
            obj = os.path.join(B.LIB_DIR, os.path.splitext(s)[0] + '.o')
            bad = audit_object(obj)
            report(os.path.basename(obj), bad)
            total += len(bad)
        return 1 if total else 0
    hipcc = B.find_hipcc()
    tmp = tempfile.mkdtemp(prefix='wn_audit_')
    total = 0
    srcs = [s for s in B.SOURCES if s.endswith('.hip')]
    procs = []
    for s in srcs:
        d = os.path.join(tmp, s)
        os.makedirs(d)
        cmd = [hipcc, '--offload-arch=gfx950'] + B.CODEGEN_FLAGS + ['-fPIC', '-I', os.path.join(ROOT, 'include'), '-I', B.CSRC,
               '-save-temps', '-c', os.path.join(B.CSRC, s), '-o', 'x.o'] + os.environ.get('WN_EXTRA_FLAGS', '').split()
        procs.append((s, d, subprocess.Popen(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    for s, d, p in procs:
        p.wait()
        asm = [f for f in os.listdir(d) if f.endswith('gfx950.s')]
        if not asm:
            print('no assembly for', s)
            total += 1
            continue
        bad = audit(os.path.join(d, asm[0]))
        report(s, bad)
        total += len(bad)
    return 1 if total else 0


if __name__ == '__main__':
    sys.exit(main())
