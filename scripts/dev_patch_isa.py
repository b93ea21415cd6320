"""dev: rebuild wn_iaf_g.o from a PATCHED copy of the compiler's device assembly (experiments on instruction forms / waits
that cannot be forced from the source).  Needs /tmp/wn_iaf_g-{hip-amdgcn-amd-amdhsa-gfx950,host-x86_64-unknown-linux-gnu}.s
from `hipcc ... -c wn_iaf_g.hip -save-temps=obj` run in /tmp.   python scripts/dev_patch_isa.py <patch> -> vlibs/lib_p<patch>.so"""
import glob, os, re, subprocess, sys
LLVM = '/opt/rocm/lib/llvm/bin/'
name = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else '/tmp/wn_iaf_g'
dev = open(src + '-hip-amdgcn-amd-amdhsa-gfx950.s').read()
n = [0]
def pk_scalar(m):
    n[0] += 1
    a, b, c, d, e, f, g, h = (int(x) for x in m.groups())
    assert a not in (d,) and True
    return '\tv_fma_f32 v%d, v%d, v%d, v%d\n\tv_fma_f32 v%d, v%d, v%d, v%d' % (a, c, f, g, b, d, f, h)
if name == 'pkscalar':
    dev = re.sub(r'\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\]', pk_scalar, dev)
elif name == 'pknop':
    def f(m):
        n[0] += 1
        return '\ts_nop 7\n\ts_nop 7\n' + m.group(0)
    dev = re.sub(r'\tv_pk_fma_f32 [^\n]* op_sel:\[0,1,0\]', f, dev)
elif name.startswith('keep'):
    # every high-for-low v_pk_fma_f32 replaced EXCEPT those of one (previous opcode, next opcode) class: which surroundings fail?
    lines = dev.split('\n')
    real = [i for i, l in enumerate(lines) if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    classes = {}
    for k, i in enumerate(real):
        if re.match(r'\tv_pk_fma_f32 .* op_sel:\[0,1,0\]', lines[i]):
            classes.setdefault((lines[real[k - 1]].split()[0], lines[real[k + 1]].split()[0]), []).append(i)
    order = sorted(classes, key=lambda c: (-len(classes[c]), c))
    keep = order[int(name[4:])]
    print('classes:', [(c, len(classes[c])) for c in order], ' kept unpatched:', keep)
    for c in order:
        if c == keep:
            continue
        for i in classes[c]:
            lines[i] = re.sub(r'\tv_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,0\]', pk_scalar, lines[i])
    dev = '\n'.join(lines)
elif name == 'none':
    pass
else:
    sys.exit('unknown patch')
print('patched', n[0], 'sites')
os.makedirs('/tmp/pisa', exist_ok=True)
open('/tmp/pisa/dev.s', 'w').write(dev)
run = lambda *a: subprocess.run(a, check=True)
run(LLVM + 'clang', '-cc1as', '-triple', 'amdgcn-amd-amdhsa', '-filetype', 'obj', '-target-cpu', 'gfx950', '-mrelocation-model', 'pic', '-o', '/tmp/pisa/dev.o', '/tmp/pisa/dev.s')
run(LLVM + 'lld', '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', '/tmp/pisa/dev.out', '/tmp/pisa/dev.o')
run(LLVM + 'clang-offload-bundler', '-type=o', '-bundle-align=4096', '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950',
    '-input=/dev/null', '-input=/tmp/pisa/dev.out', '-output=/tmp/pisa/dev.hipfb')
host = open(src + '-host-x86_64-unknown-linux-gnu.s').read()
size = os.path.getsize('/tmp/pisa/dev.hipfb')
host, k = re.subn(r'(\.L__unnamed_\d+):\n\t\.asciz\t"__CLANG_OFFLOAD_BUNDLE__[^\n]*\n\t\.size\t\.L__unnamed_\d+, \d+',
                  lambda m: '%s:\n\t.incbin "/tmp/pisa/dev.hipfb"\n\t.size\t%s, %d' % (m.group(1), m.group(1), size), host)
assert k == 1
open('/tmp/pisa/host.s', 'w').write(host)
run(LLVM + 'clang', '-c', '-fPIC', '/tmp/pisa/host.s', '-o', 'vlibs/p_%s.o' % name)
objs = [o for o in glob.glob('nsynth_wavenet_amd/lib/*.o') if not o.endswith('_v.o') and not o.endswith('/wn_iaf_g.o')]
run('/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, 'vlibs/p_%s.o' % name, '-o', 'vlibs/lib_p%s.so' % name)
print('vlibs/lib_p%s.so' % name)
