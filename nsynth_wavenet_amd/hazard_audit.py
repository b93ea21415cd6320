"""Audit of the compiled gfx950 kernels for the wide-store data hazard.

A buffer / global store of more than 64 bits reads its data VGPRs some cycles AFTER it issues.  The ISA
manual asks for one wait state before a VALU instruction overwrites them and exempts buffer stores whose
SOFFSET is an SGPR; LLVM's hazard recognizer implements that exemption (GCNHazardRecognizer::
createsVALUHazard).  On gfx950 the exemption does not hold: round 3 lost quads of lanes of a
`buffer_store_dwordx4 v[48:51], v115, s[40:43], s75 offen` to the `v_max_f32 v48, ...` right behind it
(profiles/r03_store_hazard.txt).  The one store helper of the residual stream (buf_st4, wn_mfma_h.h) therefore holds two
wait states behind every such store, with the stored registers kept alive up to them; this script is the second line:
it disassembles nothing, it reads the compiler's own .s and fails if a wide store
is followed within WINDOW instruction slots by a VALU write of its data registers.

Second rule, same mechanism: an MFMA that reads a VGPR with fewer than two wait states behind the VALU instruction that
wrote it gets the OLD register content (scripts/ubench/valu_to_mfma.hip).  hipcc pads its own VALU -> MFMA pairs but
cannot see into an asm statement.  Round 3's split codec was such asm and needed a fence at every use site; since round
4 the split is plain C++ (wn_codec.h) and no asm result feeds an MFMA any more.  The audit still flags any asm VALU
result that an MFMA reads earlier than two wait states behind it, should one come back.

Third rule (round 4): a packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose op_sel takes the HIGH
register of its src1 pair for the LOW lane -- what the compiler makes of two scalar operations that share a factor sitting
in an odd register, e.g. the second word of a ds_read_b64 -- is not reliable on gfx950 next to matrix instructions:
scripts/ubench/pk_opsel.hip gets ~11 % wrong results from `v_pk_fma_f32 vD, vA, v[p:p+1], vC op_sel:[0,1,0]` with an
MFMA issued directly behind it (none with one instruction in between, none for op_sel_hi / plain forms / src2 selects), and
a group-kernel build whose epilogue scales were read per block (compiler: 24 such instructions per kernel, no MFMA directly
behind any of them) was not repeatable call to call until those instructions were replaced, in the compiler's assembly, by
two v_fma_f32 on the same registers (profiles/r04_pk_opsel_hazard.txt).  The kernel sources are built without the SLP
vectorizer (build.py: CODEGEN_FLAGS), which is what formed those instructions; the audit flags every packed-fp32 instruction
with a src1 high-for-low select.

Two front ends, one rule set:
  * audit(path.s)          the compiler's own assembly (-save-temps): inline-asm statements are marked there, so the second
                           rule can be limited to producers the compiler cannot see into;
  * audit_object(path.o)   the device code of a BUILT object (llvm-objdump --offloading + -d): what actually ships.  No asm
                           markers survive, so the second rule is applied to every VALU producer -- hipcc pads its own
                           pairs, a finding means an unpadded one.  nsynth_wavenet_amd.build.build() runs this on every
                           object it has just produced and refuses to link the library on a finding.

    python scripts/audit_store_hazard.py            # (CLI of this module) compiles csrc/*.hip with -save-temps into a temp dir, audits the .s
    python scripts/audit_store_hazard.py --objects  # audits the device code of nsynth_wavenet_amd/lib/*.o
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = 2          # instruction slots behind the store that must not write its data


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(path, every_valu=False):
    """Findings in one assembly listing: the compiler's .s, or (every_valu=True) llvm-objdump's disassembly of a built
    object, where no asm markers exist and the second rule covers every VALU producer."""
    bad = []
    kernel = None
    ins = []            # (line number, text, kernel)
    in_asm = set()      # indices of instructions that came from an inline-asm statement
    asm = False
    for ln, l in enumerate(open(path), 1):
        s = l.strip()
        m = re.match(r'^(?:[0-9a-f]+ <)?(_Z\w+)>?:', s)
        if m:
            kernel = m.group(1)
        if s.startswith(';;#ASMSTART'):
            asm = True
        elif s.startswith(';;#ASMEND'):
            asm = False
        if not s or s[0] in ';.' or s.endswith(':') or s.startswith('//') or 'file format' in s or s.startswith('Disassembly'):
            continue
        if asm or every_valu:
            in_asm.add(len(ins))
        ins.append((ln, s.split(';')[0].split('//')[0].strip(), kernel))
    # second rule: a VALU instruction inside an asm statement (the split-fp16 codec's v_fma_mix*) whose result an MFMA
    # reads as an operand with fewer than two wait states between them -- the MFMA then gets the OLD register content
    # (scripts/ubench/valu_to_mfma.hip); hipcc pads nothing around instructions it cannot see into
    for i, (ln, s, k) in enumerate(ins):
        if i not in in_asm or not s.startswith('v_') or re.match(r'v_(mfma|cmp|readfirstlane|readlane)', s):
            continue
        dst = regs(s[len(s.split()[0]):].split(',')[0])
        states = 0          # wait states between the write and the candidate reader (an instruction = 1, s_nop N = N + 1)
        j = 1
        while states < 2 and i + j < len(ins):
            ln2, s2, _ = ins[i + j]
            op2 = s2.split()[0]
            # (a DPP instruction reading a freshly written VGPR needs the same two states: also flagged)
            if (op2.startswith('v_mfma') or op2.endswith('_dpp')) and dst & regs(','.join(s2.split(',')[1:])):
                bad.append((k, ln, s, ln2, s2))
                break
            states += int(s2.split()[1]) + 1 if s2.startswith('s_nop') else 1
            j += 1
    # third rule: packed fp32 with the high register of src1 selected for the low lane
    for i, (ln, s, k) in enumerate(ins):
        if re.match(r'v_pk_(fma|mul|add)_f32\b', s):
            m = re.search(r'\bop_sel:\[([01]),([01])', s)
            if m and m.group(2) == '1':
                bad.append((k, ln, s, ln, 'op_sel takes the high register of src1 for the low lane'))
    for i, (ln, s, k) in enumerate(ins):
        m = re.match(r'(buffer|global|scratch|flat)_store_dwordx[34]\s+(.*)', s)
        if not m:
            continue
        args = [a.strip() for a in m.group(2).split(',')]
        data = regs(args[0]) if m.group(1) == 'buffer' else regs(args[1])
        for j in range(1, WINDOW + 1):
            if i + j >= len(ins):
                break
            ln2, s2, _ = ins[i + j]
            op = s2.split()[0]
            if op.startswith('s_nop'):
                n = int(s2.split()[1]) + 1
                if j + n > WINDOW:
                    break
                continue
            if not op.startswith('v_') or op.startswith('v_cmp') or op.startswith('v_mfma'):
                if op.startswith('s_') or op.startswith('ds_') or op.startswith('buffer_') or op.startswith('global_'):
                    continue
            dst = s2[len(op):].split(',')[0]
            if op.startswith('v_') and regs(dst) & data:
                bad.append((k, ln, s, ln2, s2))
    return bad


def llvm_tool(name):
    for d in (os.environ.get('WN_LLVM_BIN'), '/opt/rocm/lib/llvm/bin', '/opt/rocm/llvm/bin'):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    raise RuntimeError(name + ' not found (set WN_LLVM_BIN)')


def disassemble_object(obj, workdir, allow_no_kernels=False):
    """Device code of a hipcc object (offload bundle) or of a bare amdgcn ELF -> path of its llvm-objdump -d listing.

    FAILS CLOSED: the listing handed to the audit must be that of a gfx950 device image and must contain kernel code.
    A bundle llvm-objdump cannot unpack, a changed image naming, or an object with no device image at all would otherwise
    be "audited" as the host object inside -- which has no gfx950 instruction and therefore no finding, exactly like a clean
    device object -- and build() would link a library whose device code nobody looked at.  `allow_no_kernels` is for the
    one object that really has no kernel (wn_host.o)."""
    import shutil
    objdump = llvm_tool('llvm-objdump')
    local = os.path.join(workdir, os.path.basename(obj))
    shutil.copy(obj, local)
    with open(local, 'rb') as f:
        head = f.read(20)
    is_amdgcn_elf = head[:4] == b'\x7fELF' and head[18:20] == (224).to_bytes(2, 'little')     # e_machine EM_AMDGPU
    if is_amdgcn_elf:
        target = local                                                  # a bare device ELF (the crafted test objects)
    else:
        r = subprocess.run([objdump, '--offloading', local], cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError('llvm-objdump --offloading failed on {} (exit {}): {}'.format(obj, r.returncode, r.stderr.strip()))
        dev = sorted(f for f in os.listdir(workdir) if f.startswith(os.path.basename(obj) + '.') and 'amdgcn' in f and 'gfx950' in f)
        if not dev and allow_no_kernels:
            # the host-only unit: acceptable only if the object really carries no offload bundle
            sec = subprocess.run([objdump, '-h', local], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if sec.returncode == 0 and '.hip_fatbin' not in sec.stdout:
                return None
        if not dev:
            raise RuntimeError('hazard audit: no gfx950 device image could be extracted from {} (llvm-objdump --offloading '
                               'produced {}); refusing to audit the host object in its place'.format(
                                   obj, sorted(os.listdir(workdir))))
        target = os.path.join(workdir, dev[0])
    out = local + '.dis'
    with open(out, 'w') as f:
        r = subprocess.run([objdump, '-d', target], stdout=f, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError('llvm-objdump failed on {}: {}'.format(obj, r.stderr))
    with open(out) as f:
        listing = f.read()
    if 'elf64-amdgpu' not in listing:
        raise RuntimeError('hazard audit: the listing of {} is not an AMDGPU disassembly'.format(obj))
    if 's_endpgm' not in listing and not allow_no_kernels:
        raise RuntimeError('hazard audit: the device image of {} contains no kernel code (no s_endpgm): nothing was audited'.format(obj))
    return out


def audit_object(obj, allow_no_kernels=False):
    """Findings in the device code of one built object; raises when no gfx950 kernel code could be extracted from it
    (allow_no_kernels: the host-only translation unit)."""
    import shutil
    tmp = tempfile.mkdtemp(prefix='wn_audit_obj_')
    try:
        listing = disassemble_object(obj, tmp, allow_no_kernels)
        return audit(listing, every_valu=True) if listing else []
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def report(name, bad):
    print('%-16s %d finding(s)' % (name, len(bad)))
    for k, ln, st, ln2, w in bad:
        print('   %s\n      %d: %s\n      %d: %s' % (k, ln, st, ln2, w))


