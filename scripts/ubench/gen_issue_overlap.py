#!/usr/bin/env python3
"""Generator of scripts/ubench/issue_overlap.hip (dev tool, not product).

Question (round-3 review, item 1): do VALU / transcendental instructions issue in the shadow of a running MFMA on a
gfx950 SIMD -- inside one wave, and between the waves of a SIMD?  scripts/ubench/mfma_valu_overlap.hip (round 3) said
no; MI355X_MICROARCH.md measures five hidden issue slots per v_mfma_f32_32x32x16.  That benchmark left the instruction
order to the compiler (which also SLP-packed the filler into v_pk_* ops); here every loop body is a HAND-PLACED asm
stream: explicit registers, no compiler scheduling, the ISA in the binary is the text below.

Every kernel: one workgroup per CU, waves [0, nm) run the M body (U MFMAs per iteration, each followed by N fillers),
waves [nm, nm + nv) the V body (64 fillers per iteration).  A wave records s_memtime before / after its loop and its
HW_ID (SIMD id), so the table can be read per SIMD.
"""
import itertools
import sys

SHAPES = {
    # name: (opcode, acc regs, independent accumulators per iteration, pipe cycles)
    "16": ("v_mfma_f32_16x16x32_f16", 4, 12, 16),
    "32": ("v_mfma_f32_32x32x16_f16", 16, 4, 32),
}
ACC0 = 8          # v[0:3] = A, v[4:7] = B
FIL0 = 96         # 16 independent filler registers v96..v111 (pairs for pk)
C1, C0 = 112, 113   # constants 1.0 and 0.0 (and pairs 112:113 / 114:115)


def filler(kind, i):
    r = FIL0 + (i % 16)
    if kind == "fma":
        return f"v_fma_f32 v{r}, v{C1}, v{r}, v{C0}"
    if kind == "exp":
        return f"v_exp_f32 v{r}, v{r}"
    if kind == "rcp":
        return f"v_rcp_f32 v{r}, v{r}"
    if kind == "pk":
        p = FIL0 + 2 * (i % 8)
        return f"v_pk_fma_f32 v[{p}:{p + 1}], v[{C1}:{C1 + 1}], v[{p}:{p + 1}], v[{C1 + 2}:{C1 + 3}]"
    if kind == "mix":   # the split codec's remainder instruction
        return f"v_fma_mixlo_f16 v{r}, v{C1}, -1.0, v{r} op_sel:[0,0,0] op_sel_hi:[1,0,0]"
    if kind == "epi":   # issue mix of a residual epilogue: 5 plain : 1 transcendental, dependent pairs
        j = i % 6
        if j == 5:
            return f"v_exp_f32 v{r}, v{r}"
        return f"v_fma_f32 v{r}, v{C1}, v{r}, v{C0}"
    raise ValueError(kind)


def m_body(shape, n, kind, chain=0):
    op, nacc, u, _ = SHAPES[shape]
    if chain:
        u_eff, accs = 12 if shape == "16" else 4, chain
    else:
        u_eff, accs = u, u
    lines, f = [], 0
    for i in range(u_eff):
        a = ACC0 + nacc * (i % accs)
        lines.append(f"{op} v[{a}:{a + nacc - 1}], v[0:3], v[4:7], v[{a}:{a + nacc - 1}]")
        for _ in range(n):
            lines.append(filler(kind, f))
            f += 1
    return lines, u_eff


def v_body(kind, count=64):
    return [filler(kind, i) for i in range(count)]


def asm_block(body, prio=None):
    init = [f"v_mov_b32 v{i}, 0x14001400" for i in range(8)]
    init += [f"v_mov_b32 v{i}, 0" for i in range(ACC0, ACC0 + 64)]
    init += [f"v_mov_b32 v{FIL0 + i}, 1.0" for i in range(16)]
    init += [f"v_mov_b32 v{C1}, 1.0", f"v_mov_b32 v{C0}, 0", f"v_mov_b32 v{C1 + 2}, 0", f"v_mov_b32 v{C1 + 3}, 0",
             f"v_mov_b32 v{C1 + 1}, 1.0"]
    pre = init[:]
    if prio is not None:
        pre.append(f"s_setprio {prio}")
    pre += ["s_barrier", "s_memrealtime %3", "s_memtime %0", "s_waitcnt lgkmcnt(0)", "L_loop_%=:"]
    post = ["s_sub_u32 %2, %2, 1", "s_cmp_lg_u32 %2, 0", "s_cbranch_scc1 L_loop_%=", "s_nop 7", "s_nop 7", "s_memtime %1",
            "s_memrealtime %4", "s_waitcnt lgkmcnt(0)"]
    if prio is not None:
        post.append("s_setprio 0")
    text = "\\n\\t".join(pre + body + post)
    clob = ", ".join(f'"v{i}"' for i in range(0, 120))
    return (f'asm volatile("{text}"\n                 : "=&s"(t0), "=&s"(t1), "+s"(it), "=&s"(r0), "=&s"(r1) : : {clob}, "scc", "memory");')


KERNELS = []   # (name, nm, nv, m_lines, v_lines, mfma per iter, fillers per M iter, fillers per V iter, prio_m, prio_v, descr)


def add(name, nm, nv, shape=None, n=0, kind="fma", vkind="fma", chain=0, prio_m=None, prio_v=None):
    ml, u = (m_body(shape, n, kind, chain) if nm else ([], 0))
    vl = v_body(vkind) if nv else []
    KERNELS.append(dict(name=name, nm=nm, nv=nv, ml=ml, vl=vl, u=u, fm=u * n, fv=64 if nv else 0, prio_m=prio_m, prio_v=prio_v,
                        shape=shape or "-", n=n, kind=kind if nm else "-", vkind=vkind if nv else "-", chain=chain))


# E1: one wave per SIMD, N fillers hand-placed behind every MFMA
for shape, kind in itertools.product(("16", "32"), ("fma", "exp", "pk", "mix", "epi")):
    for n in (0, 1, 2, 3, 4, 5, 6, 8, 12):
        if n == 0 and kind != "fma":
            continue
        add(f"e1_s{shape}_{kind}_n{n}", 4, 0, shape, n, kind)
# E2: filler only, 1 / 2 / 3 waves per SIMD
for kind in ("fma", "exp", "rcp", "pk", "mix", "epi"):
    for w in (1, 2, 3):
        add(f"e2_{kind}_w{w}", 0, 4 * w, vkind=kind)
# E3: MFMA-only waves beside filler-only waves on the same SIMD (waves w and w + 4 share a SIMD)
for shape in ("16", "32"):
    for vkind in ("fma", "exp", "epi"):
        add(f"e3_s{shape}_m1_v1_{vkind}", 4, 4, shape, 0, vkind=vkind)
        add(f"e3_s{shape}_m1_v2_{vkind}", 4, 8, shape, 0, vkind=vkind)
    add(f"e3_s{shape}_m1_v1_fma_priom", 4, 4, shape, 0, vkind="fma", prio_m=1)
    add(f"e3_s{shape}_m1_v1_fma_priov", 4, 4, shape, 0, vkind="fma", prio_v=1)
    add(f"e3_s{shape}_m1_v2_epi_priom", 4, 8, shape, 0, vkind="epi", prio_m=1)
# E4: MFMA only, 2 / 3 waves per SIMD (the pipe is shared)
for shape in ("16", "32"):
    for w in (2, 3):
        add(f"e4_s{shape}_w{w}", 4 * w, 0, shape, 0)
# E5: interleaved waves: every wave carries MFMA + N fillers, 2 / 3 waves per SIMD
for shape in ("16", "32"):
    for w, n in ((2, 2), (2, 3), (3, 2), (3, 3), (3, 4)):
        add(f"e5_s{shape}_w{w}_epi_n{n}", 4 * w, 0, shape, n, "epi")
# E6: dependent accumulator chains (mfma3 of the split product is three MFMAs on ONE accumulator)
for shape in ("16", "32"):
    for chain in (1, 2, 3, 4):
        add(f"e6_s{shape}_chain{chain}", 4, 0, shape, 0, chain=chain)
        add(f"e6_s{shape}_chain{chain}_fma3", 4, 0, shape, 3, "fma", chain=chain)


def emit(out):
    w = out.write
    w("// GENERATED by scripts/ubench/gen_issue_overlap.py -- do not edit.  (dev tool, not product)\n")
    w("// Hand-placed MFMA / VALU issue streams on gfx950: see the generator's docstring.\n")
    w("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>\n#include <string>\n#include <vector>\n\n")
    w("struct Rec { unsigned long long t0, t1, r0, r1; unsigned hwid, role; };\n\n")
    for k in KERNELS:
        nt = 64 * (k["nm"] + k["nv"])
        w(f"// {k['name']}: M waves {k['nm']}, V waves {k['nv']}\n")
        w(f"extern \"C\" __global__ __launch_bounds__({nt}) void {k['name']}(Rec* out, int iters_m, int iters_v) {{\n")
        w("    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);\n")
        w("    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;\n    unsigned role;\n    int it;\n")
        if k["nm"] and k["nv"]:
            w(f"    if (wave < {k['nm']}) {{\n        role = 0; it = iters_m;\n        ")
            w(asm_block(k["ml"], k["prio_m"]))
            w("\n    } else {\n        role = 1; it = iters_v;\n        ")
            w(asm_block(k["vl"], k["prio_v"]))
            w("\n    }\n")
        elif k["nm"]:
            w("    role = 0; it = iters_m;\n    ")
            w(asm_block(k["ml"], k["prio_m"]))
            w("\n")
        else:
            w("    role = 1; it = iters_v;\n    ")
            w(asm_block(k["vl"], k["prio_v"]))
            w("\n")
        w("    unsigned hwid;\n    asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hwid));\n")
        w(f"    if ((threadIdx.x & 63) == 0) out[blockIdx.x * {k['nm'] + k['nv']} + wave] = Rec{{t0, t1, r0, r1, hwid, role}};\n}}\n\n")
    w("struct K { const char* name; void (*fn)(Rec*, int, int); int nm, nv, u, fm, fv, pipe; };\n")
    w("static const K kernels[] = {\n")
    for k in KERNELS:
        pipe = SHAPES[k["shape"]][3] if k["nm"] else 0
        w(f"    {{\"{k['name']}\", {k['name']}, {k['nm']}, {k['nv']}, {k['u']}, {k['fm']}, {k['fv']}, {pipe}}},\n")
    w("};\n\n")
    w(r'''
int main(int argc, char** argv) {
    const char* only = argc > 1 ? argv[1] : "";
    const int iters = 4000;
    Rec* d;
    hipMalloc(&d, 256 * 16 * sizeof(Rec));
    std::vector<Rec> h(256 * 16);
    printf("# kernel | waves M/V | per iteration: MFMAs, fillers(M), fillers(V) | M waves: cycles/iter (cycles/MFMA, pipe cycles/MFMA) |"
           " V waves: cycles/iter (cycles/filler) | co-run window: see notes\n");
    for (const K& k : kernels) {
        if (only[0] && !strstr(k.name, only)) continue;
        // the V waves of a mixed kernel run about half as long as the M waves (so that they live entirely beside MFMAs),
        // a second launch swaps that (M waves entirely beside fillers)
        for (int pass = 0; pass < ((k.nm && k.nv) ? 2 : 1); ++pass) {
            int im = iters, iv = iters;
            if (k.nm && k.nv) {
                const double tm = (double)k.u * k.pipe, tv = 64.0 * 4.0;    // rough solo cycles per iteration
                if (pass == 0) iv = (int)(0.4 * iters * tm / tv); else im = (int)(0.4 * iters * tv / tm);
                if (iv < 50) iv = 50;
                if (im < 50) im = 50;
            }
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(k.fn, dim3(256), dim3(64 * (k.nm + k.nv)), 0, 0, d, im, iv);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, 256 * (k.nm + k.nv) * sizeof(Rec), hipMemcpyDeviceToHost);
            // workgroup 37 (any): per wave cycles; averages over all workgroups
            double sm = 0, sv = 0, mn[2] = {1e30, 1e30}, mx[2] = {0, 0}, ticks = 0, real = 0;
            int cm = 0, cv = 0;
            for (int b = 0; b < 256; ++b)
                for (int w = 0; w < k.nm + k.nv; ++w) {
                    const Rec& r = h[b * (k.nm + k.nv) + w];
                    const double c = (double)(r.t1 - r.t0);
                    if (r.role == 0) { sm += c / im; ++cm; } else { sv += c / iv; ++cv; }
                    const double per = c / (r.role == 0 ? im : iv);
                    if (per < mn[r.role]) mn[r.role] = per;
                    if (per > mx[r.role]) mx[r.role] = per;
                    ticks += c;
                    real += (double)(r.r1 - r.r0);
                }
            printf("%-28s %2d/%-2d u=%2d fm=%3d fv=%2d iters %5d/%-5d |", k.name, k.nm, k.nv, k.u, k.fm, k.fv, im, iv);
            if (cm) printf(" M %8.1f cyc/iter = %6.2f cyc/MFMA (pipe %d)", sm / cm, sm / cm / k.u, k.pipe);
            if (cm && k.fm) printf(" [%5.2f cyc/instr]", sm / cm / (k.u + k.fm));
            if (cv) printf(" | V %8.1f cyc/iter = %5.2f cyc/filler", sv / cv, sv / cv / k.fv);
            if (cm) printf(" | M min %.1f max %.1f", mn[0], mx[0]);
            if (cv) printf(" | V min %.1f max %.1f", mn[1], mx[1]);
            printf(" | s_memtime %.3f GHz (vs 100 MHz s_memrealtime)", real > 0 ? ticks / real * 0.1 : 0.0);
            // SIMD placement of workgroup 37
            printf(" | simd:");
            for (int w = 0; w < k.nm + k.nv; ++w) printf("%u", (h[37 * (k.nm + k.nv) + w].hwid >> 4) & 3);
            printf("\n");
        }
    }
    return 0;
}
''')


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else "issue_overlap.hip"
    with open(path, "w") as f:
        emit(f)
    print(f"{len(KERNELS)} kernels -> {path}")
