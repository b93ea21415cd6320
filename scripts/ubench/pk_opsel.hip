// ubench: packed-fp32 instructions with operand selects next to matrix instructions on gfx950 (round 4).
// Found through a group-kernel build that was not repeatable call to call: the compiler had turned two scalar FMAs whose
// common factor sat in the HIGH register of a pair loaded by ds_read_b64 into
//     v_pk_fma_f32 vD, vA, v[p:p+1], vC op_sel:[0,1,0]          (both lanes take p+1)
// and replacing those 24 instructions per kernel, in the compiler's assembly, by two v_fma_f32 on the same registers made
// the kernel bit-exact again (scripts/dev_patch_isa.py; profiles/r04_pk_opsel_hazard.txt).  This file hunts the trigger
// with hand-placed sequences: every iteration loads a different (p0, p1) into the same pair v[100:101] with ds_read_b64,
// waits for it, runs ONE packed instruction in the named surroundings and compares with scalar instructions on the same
// registers.  Result: a src1 select that takes the high register for the LOW lane goes wrong ~11 % of the time when an
// MFMA of the same wave is issued DIRECTLY behind the packed instruction (one instruction or s_nop 0 in between: never);
// plain forms, op_sel_hi (low register for both lanes), src0 / src2 selects: never.  The surroundings of the 24
// instructions in the kernel (no MFMA directly behind any of them) are not among the forms below -- the in-kernel
// trigger is narrower than what was tried here, which is why the kernels avoid the instruction form altogether
// (-fno-slp-vectorize, scripts/audit_store_hazard.py rule 3).
//   hipcc --offload-arch=gfx950 -O2 pk_opsel.hip -o pk_opsel && ./pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define BODY(MFMA, AFTER, PKINSTR, E0, E1)                                                                                       \
    asm volatile("ds_read_b64 v[100:101], %4\n\t"                                                                          \
                 "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v110, %7\n\tv_mov_b32 v111, %8\n\t" MFMA "\n\t"    \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                                \
                 "s_nop 7\n\t" PKINSTR "\n\t" AFTER "\n\t"                                                                \
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" E0 "\n\t" E1 "\n\t"                                                                          \
                 "v_mov_b32 %0, v106\n\tv_mov_b32 %1, v107"                                                                \
                 : "=&v"(d0), "=&v"(d1), "=&v"(e0), "=&v"(e1) : "v"(lds_pair), "v"(c0), "v"(c1), "v"(a0), "v"(a1), "v"(a), "v"(b), "v"(gptr) \
                 : "v100", "v101", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "memory")
#define BODY2(NOPS, PKINSTR, E0, E1)                                                                                      \
    asm volatile("ds_read_b64 v[100:101], %4\n\t"                                                                          \
                 "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v110, %7\n\tv_mov_b32 v111, %8\n\t"                \
                 "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t" MF "\n\t" NOPS "\n\t" PKINSTR "\n\t"                                \
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" E0 "\n\t" E1 "\n\t"                                                   \
                 "v_mov_b32 %0, v106\n\tv_mov_b32 %1, v107"                                                                \
                 : "=&v"(d0), "=&v"(d1), "=&v"(e0), "=&v"(e1) : "v"(lds_pair), "v"(c0), "v"(c1), "v"(a0), "v"(a1), "v"(a), "v"(b), "v"(gptr) \
                 : "v100", "v101", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "memory")
#define MF3 "v_mfma_f32_16x16x32_f16 v[114:117], %9, %10, 0\n\tv_mfma_f32_16x16x32_f16 v[114:117], %9, %10, v[114:117]\n\tv_mfma_f32_16x16x32_f16 v[114:117], %9, %10, v[114:117]"
#define BODY3(NOPS, PKINSTR, E0, E1)                                                                                      \
    asm volatile("ds_read_b64 v[100:101], %4\n\t"                                                                          \
                 "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v110, %7\n\tv_mov_b32 v111, %8\n\t"                \
                 "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\t" MF3 "\n\t" NOPS "\n\t" PKINSTR "\n\t"                               \
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t" E0 "\n\t" E1 "\n\t"                    \
                 "v_mov_b32 %0, v106\n\tv_mov_b32 %1, v107"                                                                \
                 : "=&v"(d0), "=&v"(d1), "=&v"(e0), "=&v"(e1) : "v"(lds_pair), "v"(c0), "v"(c1), "v"(a0), "v"(a1), "v"(a), "v"(b), "v"(gptr) \
                 : "v100", "v101", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "memory")
#define MF "v_mfma_f32_16x16x32_f16 v[114:117], %9, %10, 0"
#define NOMF "s_nop 0"
template <int FORM>
__device__ inline void body(float& d0, float& d1, float& e0, float& e1, unsigned lds_pair, float c0, float c1, float a0, float a1, h8 a, h8 b, const float* gptr) {
    if (FORM == 0) BODY(NOMF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 1) BODY(MF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 2) BODY(NOMF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105]", "v_fma_f32 %2, v110, v100, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 3) BODY(NOMF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel_hi:[1,0,1]", "v_fma_f32 %2, v110, v100, v104", "v_fma_f32 %3, v111, v100, v105");
    if (FORM == 4) BODY(NOMF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v100, v105");
    if (FORM == 5) BODY(NOMF, NOMF, "v_pk_mul_f32 v[106:107], v[110:111], v[100:101] op_sel:[0,1]", "v_mul_f32 %2, v110, v101", "v_mul_f32 %3, v111, v101");
    if (FORM == 6) BODY(NOMF, NOMF, "v_pk_fma_f32 v[106:107], v[100:101], v[110:111], v[104:105] op_sel:[1,0,0]", "v_fma_f32 %2, v101, v110, v104", "v_fma_f32 %3, v101, v111, v105");
    if (FORM == 7) BODY(NOMF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[104:105], v[100:101] op_sel:[0,0,1]", "v_fma_f32 %2, v110, v104, v101", "v_fma_f32 %3, v111, v105, v101");
    if (FORM == 8) BODY(NOMF, MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 9) BODY(NOMF, MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105]", "v_fma_f32 %2, v110, v100, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 10) BODY(NOMF, MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel_hi:[1,0,1]", "v_fma_f32 %2, v110, v100, v104", "v_fma_f32 %3, v111, v100, v105");
    if (FORM == 11) BODY(NOMF, MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v100, v105");
    if (FORM == 12) BODY(NOMF, MF, "v_mov_b32 v112, v100\n\tv_mov_b32 v113, v101\n\tv_pk_fma_f32 v[106:107], v[110:111], v[112:113], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 13) BODY(NOMF, MF, "v_pk_mul_f32 v[106:107], v[110:111], v[100:101] op_sel:[0,1]", "v_mul_f32 %2, v110, v101", "v_mul_f32 %3, v111, v101");
    if (FORM == 14) BODY(NOMF, MF, "v_pk_fma_f32 v[106:107], v[110:111], v[104:105], v[100:101] op_sel:[0,0,1]", "v_fma_f32 %2, v110, v104, v101", "v_fma_f32 %3, v111, v105, v101");
    if (FORM == 15) BODY(NOMF, MF, "v_pk_add_f32 v[106:107], v[110:111], v[100:101] op_sel:[0,1]", "v_add_f32 %2, v110, v101", "v_add_f32 %3, v111, v101");
#define FILL1 "v_add_u32 v112, v112, v113\n\t"
    if (FORM == 16) BODY(NOMF, FILL1 MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 17) BODY(NOMF, FILL1 FILL1 MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 18) BODY(NOMF, FILL1 FILL1 FILL1 FILL1 MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 19) BODY(NOMF, FILL1 FILL1 FILL1 FILL1 FILL1 FILL1 FILL1 FILL1 MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 20) BODY(NOMF, "s_nop 0\n\t" MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 21) BODY(NOMF, "s_nop 3\n\t" MF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 22) BODY(NOMF, MF, "v_pk_fma_f32 v[106:107], v[100:101], v[110:111], v[104:105] op_sel:[1,0,0]", "v_fma_f32 %2, v101, v110, v104", "v_fma_f32 %3, v101, v111, v105");
    if (FORM == 23) BODY(MF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 24) BODY(NOMF, "v_pk_fma_f32 v[112:113], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 25) BODY(NOMF, "v_pk_add_f32 v[112:113], v[104:105], v[110:111]", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 26) BODY(NOMF, "ds_read_b128 v[114:117], %4", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "s_waitcnt lgkmcnt(0)\n\tv_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 27) BODY(NOMF, NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 28) BODY(NOMF, "v_pk_add_f32 v[112:113], v[104:105], v[110:111]", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 29) BODY2("s_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 30) BODY2("s_nop 1", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 31) BODY2("s_nop 2", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 32) BODY2("s_nop 3", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 33) BODY2("s_nop 4", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 34) BODY2("s_nop 5", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 35) BODY2("s_nop 6", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 36) BODY2("s_nop 7", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 37) BODY2("s_nop 8", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 38) BODY2("s_nop 9", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 39) BODY2("s_nop 10", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 40) BODY2("s_nop 11", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 41) BODY2("s_nop 12", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 42) BODY2("s_nop 13", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 43) BODY2("s_nop 14", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 44) BODY2("s_nop 15", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 45) BODY2("s_nop 3", "v_pk_fma_f32 v[106:107], v[114:115], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v114, v101, v104", "v_fma_f32 %3, v115, v101, v105");
    if (FORM == 46) BODY2("s_nop 4", "v_pk_fma_f32 v[106:107], v[114:115], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v114, v101, v104", "v_fma_f32 %3, v115, v101, v105");
    if (FORM == 47) BODY2("s_nop 5", "v_pk_fma_f32 v[106:107], v[114:115], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v114, v101, v104", "v_fma_f32 %3, v115, v101, v105");
    if (FORM == 48) BODY2("s_nop 6", "v_pk_fma_f32 v[106:107], v[114:115], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v114, v101, v104", "v_fma_f32 %3, v115, v101, v105");
    if (FORM == 49) BODY2("s_nop 7", "v_pk_fma_f32 v[106:107], v[114:115], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v114, v101, v104", "v_fma_f32 %3, v115, v101, v105");
    if (FORM == 50) BODY2("s_nop 9", "v_pk_fma_f32 v[106:107], v[114:115], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v114, v101, v104", "v_fma_f32 %3, v115, v101, v105");
    if (FORM == 51) BODY2("s_nop 5", "v_pk_fma_f32 v[106:107], v[114:115], v[100:101], v[104:105] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[112:113], v[116:117], v[100:101], v[104:105] op_sel:[0,1,0]\n\tv_pk_add_f32 v[112:113], v[112:113], v[104:105]", "v_fma_f32 %2, v114, v101, v104", "v_fma_f32 %3, v115, v101, v105");
    if (FORM == 52) BODY2("s_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 53) BODY2("v_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 54) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 55) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 56) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 57) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 58) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 59) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 60) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 61) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 62) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 63) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 64) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 65) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 66) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 67) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 68) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 69) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 70) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 71) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 72) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 73) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 74) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 75) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 76) BODY2("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    // VMEM / LDS loads whose RETURNS (asynchronous register writes) land at arbitrary times around the packed instruction
    if (FORM == 77) BODY("global_load_dwordx4 v[114:117], %11, off", NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 78) BODY("global_load_dwordx4 v[114:117], %11, off", NOMF, "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel_hi:[1,0,1]", "v_fma_f32 %2, v110, v100, v104", "v_fma_f32 %3, v111, v100, v105");
    if (FORM == 79) BODY2("ds_read_b128 v[114:117], %4\n\tds_read_b128 v[114:117], %4 offset:16\n\tds_read_b128 v[114:117], %4 offset:32\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 3", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "s_waitcnt lgkmcnt(0)\n\tv_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 80) BODY3("s_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 81) BODY3("v_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 82) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 83) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 84) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 85) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 86) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 87) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 88) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 89) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 90) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 91) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 92) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 93) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 94) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 95) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 96) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 97) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 98) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 99) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 100) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 101) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 102) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 103) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 104) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 105) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 106) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
    if (FORM == 107) BODY3("v_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\tv_add_u32 v112, v112, v113\n\ts_nop 0", "v_pk_fma_f32 v[106:107], v[110:111], v[100:101], v[104:105] op_sel:[0,1,0]", "v_fma_f32 %2, v110, v101, v104", "v_fma_f32 %3, v111, v101, v105");
}
template <int FORM>
__global__ __launch_bounds__(512) void probe(const float* in, unsigned* bad, unsigned* stale, int iters) {
    __shared__ __attribute__((aligned(16))) float pairs[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) pairs[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.01f * (lane + k)); b[k] = (_Float16)(0.02f * (lane - k)); }
    if (threadIdx.x >= 256) {      // forms 27, 28 are launched with 512 threads: waves 4 .. 7 share the SIMDs of 0 .. 3 and keep the matrix pipe busy
        f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int i = 0; i < iters * 6; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
        if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) bad[3] = 1;
        return;
    }
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)pairs;
    unsigned nbad = 0, nstale = 0;
    float prev0 = 0.f, prev1 = 0.f;
    for (int i = 0; i < iters; ++i) {
        float d0, d1, e0, e1;
        const float a0 = in[(lane * 7 + i) & 1023], a1 = in[(lane * 13 + i + 5) & 1023], c0 = in[(lane + i) & 1023], c1 = in[(lane * 3 + i) & 1023];
        body<FORM>(d0, d1, e0, e1, base + 8 * (i & 127), c0, c1, a0, a1, a, b, in + ((lane * 4 + i * 64) & 1020));
        if (d0 != e0 || d1 != e1) {
            ++nbad;
            // what the instruction would give with the PREVIOUS iteration's pair
            const float q1 = pairs[2 * ((i - 1) & 127) + 1];
            (void)q1;
            const float p0 = pairs[2 * (i & 127)];
            if (FORM == 8 && d0 == fmaf(a0, p0, c0) && d1 == e1) ++nstale;      // the low lane took the LOW register, the high lane is right
        }
        prev0 = d0; prev1 = d1;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (nstale) atomicAdd(stale, nstale);
    if (prev0 == 12345.f && prev1 == 1.f) bad[1] = 1;
}
int main(int argc, char** argv) {
    std::vector<float> in(1024);
    for (int i = 0; i < 1024; ++i) in[i] = 0.001f * (float)((i * 7919) % 1999) - 1.f;
    float* din;
    unsigned* dbad;
    (void)hipMalloc(&din, 4096); (void)hipMalloc(&dbad, 16);
    (void)hipMemcpy(din, in.data(), 4096, hipMemcpyHostToDevice);
    const char* fn[108] = {"pk_fma  src1 = pair, op_sel:[0,1,0] (high for both)", "same, an MFMA of the wave issued in front", "pk_fma  src1 = pair, no selects (low, high)",
                         "pk_fma  src1 = pair, op_sel_hi:[1,0,1] (low for both)", "pk_fma  src1 = pair, swapped (high, low)", "pk_mul  src1 = pair, op_sel:[0,1] (high for both)",
                         "pk_fma  src0 = pair, op_sel:[1,0,0] (high for both)", "pk_fma  src2 = pair, op_sel:[0,0,1] (high for both)",
                         "MFMA BEHIND: pk_fma src1 = pair, op_sel:[0,1,0]", "MFMA BEHIND: pk_fma src1 = pair, no selects", "MFMA BEHIND: pk_fma src1 = pair, op_sel_hi:[1,0,1]",
                         "MFMA BEHIND: pk_fma src1 = pair, swapped (high, low)", "MFMA BEHIND: pair copied by v_mov first, op_sel:[0,1,0]", "MFMA BEHIND: pk_mul src1 = pair, op_sel:[0,1]",
                         "MFMA BEHIND: pk_fma src2 = pair, op_sel:[0,0,1]", "MFMA BEHIND: pk_add src1 = pair, op_sel:[0,1]",
                         "MFMA 1 VALU instruction behind: pk_fma op_sel:[0,1,0]", "MFMA 2 VALU instructions behind", "MFMA 4 VALU instructions behind", "MFMA 8 VALU instructions behind",
                         "MFMA behind s_nop 0", "MFMA behind s_nop 3", "MFMA BEHIND: pk_fma src0 = pair, op_sel:[1,0,0]", "MFMA in FRONT: op_sel:[0,1,0]",
                         "another pk_fma op_sel:[0,1,0] BEHIND", "a pk_add BEHIND", "a ds_read_b128 BEHIND", "no MFMA of the wave, ANOTHER WAVE of the SIMD runs MFMAs",
                         "a pk_add behind, ANOTHER WAVE of the SIMD runs MFMAs",
                         "MFMA in front, s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 1, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 2, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 3, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 4, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 5, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 6, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 7, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 8, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 9, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 10, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 11, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 12, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 13, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 14, then pk_fma op_sel:[0,1,0]","MFMA in front, s_nop 15, then pk_fma op_sel:[0,1,0]","src0 = rows of the MFMA in front, s_nop 3, op_sel:[0,1,0]","src0 = rows of the MFMA in front, s_nop 4, op_sel:[0,1,0]","src0 = rows of the MFMA in front, s_nop 5, op_sel:[0,1,0]","src0 = rows of the MFMA in front, s_nop 6, op_sel:[0,1,0]","src0 = rows of the MFMA in front, s_nop 7, op_sel:[0,1,0]","src0 = rows of the MFMA in front, s_nop 9, op_sel:[0,1,0]","src0 = MFMA rows, s_nop 5, two pk_fma + pk_add like the compiler","MFMA in front, 0 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 1 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 2 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 3 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 4 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 5 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 6 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 7 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 8 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 9 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 10 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 11 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 12 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 13 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 14 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 15 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 16 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 17 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 18 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 19 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 20 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 21 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 22 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 23 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]","MFMA in front, 24 VALU instructions + s_nop 0, then pk_fma op_sel:[0,1,0]", "global loads in flight (returns at any time): op_sel:[0,1,0]", "global loads in flight: op_sel_hi:[1,0,1]", "LDS loads returning about now: op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 0 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 1 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 2 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 3 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 4 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 5 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 6 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 7 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 8 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 9 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 10 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 11 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 12 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 13 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 14 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 15 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 16 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 17 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 18 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 19 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 20 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 21 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 22 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 23 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 24 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 25 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 26 VALU instructions, then pk_fma op_sel:[0,1,0]","chain of 3 DEPENDENT MFMAs in front, 27 VALU instructions, then pk_fma op_sel:[0,1,0]"};
    const int iters = 4000;
    for (int form = (argc > 1 ? atoi(argv[1]) : 0); form < 108; ++form) {
        (void)hipMemset(dbad, 0, 16);
        if (form == 0) probe<0><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 1) probe<1><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 2) probe<2><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 3) probe<3><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 4) probe<4><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 5) probe<5><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 6) probe<6><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 7) probe<7><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 8) probe<8><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 9) probe<9><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 10) probe<10><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 11) probe<11><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 12) probe<12><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 13) probe<13><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 14) probe<14><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 15) probe<15><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 16) probe<16><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 17) probe<17><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 18) probe<18><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 19) probe<19><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 20) probe<20><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 21) probe<21><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 22) probe<22><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 23) probe<23><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 24) probe<24><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 25) probe<25><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 26) probe<26><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 27) probe<27><<<256, 512>>>(din, dbad, dbad + 2, iters);
        if (form == 28) probe<28><<<256, 512>>>(din, dbad, dbad + 2, iters);
        if (form == 29) probe<29><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 30) probe<30><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 31) probe<31><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 32) probe<32><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 33) probe<33><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 34) probe<34><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 35) probe<35><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 36) probe<36><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 37) probe<37><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 38) probe<38><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 39) probe<39><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 40) probe<40><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 41) probe<41><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 42) probe<42><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 43) probe<43><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 44) probe<44><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 45) probe<45><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 46) probe<46><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 47) probe<47><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 48) probe<48><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 49) probe<49><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 50) probe<50><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 51) probe<51><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 52) probe<52><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 53) probe<53><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 54) probe<54><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 55) probe<55><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 56) probe<56><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 57) probe<57><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 58) probe<58><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 59) probe<59><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 60) probe<60><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 61) probe<61><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 62) probe<62><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 63) probe<63><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 64) probe<64><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 65) probe<65><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 66) probe<66><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 67) probe<67><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 68) probe<68><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 69) probe<69><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 70) probe<70><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 71) probe<71><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 72) probe<72><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 73) probe<73><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 74) probe<74><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 75) probe<75><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 76) probe<76><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 77) probe<77><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 78) probe<78><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 79) probe<79><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 80) probe<80><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 81) probe<81><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 82) probe<82><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 83) probe<83><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 84) probe<84><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 85) probe<85><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 86) probe<86><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 87) probe<87><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 88) probe<88><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 89) probe<89><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 90) probe<90><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 91) probe<91><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 92) probe<92><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 93) probe<93><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 94) probe<94><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 95) probe<95><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 96) probe<96><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 97) probe<97><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 98) probe<98><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 99) probe<99><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 100) probe<100><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 101) probe<101><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 102) probe<102><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 103) probe<103><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 104) probe<104><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 105) probe<105><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 106) probe<106><<<256, 256>>>(din, dbad, dbad + 2, iters);
        if (form == 107) probe<107><<<256, 256>>>(din, dbad, dbad + 2, iters);
        unsigned r[4];
        (void)hipMemcpy(r, dbad, 16, hipMemcpyDeviceToHost);
        printf("%-58s wrong: %9u of %ld   (low lane = product with the LOW register, high lane right: %u)\n", fn[form], r[0], 256L * 256 * iters, r[2]);
    }
    return 0;
}
