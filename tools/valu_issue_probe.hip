// Probe: what one wave64 VALU instruction costs a SIMD of gfx950, measured in SHADER CYCLES
// (s_memtime around the instruction stream of every wave) instead of in wall time times an assumed
// clock, and the clock the chip actually runs such a stream at (cycles / HIP-event time).
//
// Why: tools/op_probe.hip priced v_fma_f32 at 2.7-3.1 "cycles at 2.4 GHz" from event times, 35-55 %
// above the 2 cycles of MI355X_MICROARCH.md ("Wave scheduling": a wave64 VALU instruction issues
// over 2 cycles).  This probe separates the two unknowns (issue cycles, real clock) and varies
// what could cost extra: waves per SIMD, chain count (dependent-issue distance), operand
// patterns (distinct VGPR banks / same bank / repeated source / SGPR / literal / inline constant).
//
//   c = cycles from the first wave's start to the last wave's end / (waves on the SIMD x instructions per wave)
//
// is the issue cost per instruction when the SIMD is the bottleneck (all waves of a SIMD run the same
// stream at the same time: W / 2 workgroups of 512 threads -- tile_kernel's shape -- are pinned on every CU by
// their LDS footprint; `overlap` reports which share of the kernel's span had every wave in flight).
//
// build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_issue_probe tools/valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

// 16 instructions per REP16; the register numbers are fixed so that the bank pattern is known.
// v[32..47] are the 16 accumulators (chains), v[48..51] hold operands.
enum Mode {
  FMA_DISTINCT,   // v_fma_f32 vN, vN, v48, v49     (3 VGPR sources; N, 48, 49 in different banks for most N)
  FMA_SAMEBANK,   // v_fma_f32 v32, v32, v36, v40   every source register index = 0 mod 4
  FMA_REPEAT,     // v_fma_f32 vN, vN, vN, vN       one register three times
  FMA_SGPR,       // v_fma_f32 vN, vN, s20, v49
  FMA_LITERAL,    // v_mul_f32 vN, 0x3f800347, vN   (32-bit literal; VOP3 takes none on gfx950)
  FMA_INLINE,     // v_fma_f32 vN, vN, 1.0, v49          (inline constant)
  FMAC,           // v_fmac_f32 vN, v48, v49        (VOP2)
  MUL,            // v_mul_f32 vN, vN, v48          (VOP2)
  ADD,            // v_add_f32 vN, vN, v48
  MOV,            // v_mov_b32 vN, v48
  PK_FMA,         // v_pk_fma_f32 v[N:N+1], v[N:N+1], v[48:49], v[50:51]  (8 instructions = 16 lanes-ops)
  CHAIN1,         // v_fma_f32 v32, v32, v48, v49   ONE dependent chain (dependent-issue latency)
  CHAIN2,         // two chains
  CHAIN4,         // four chains
  FMA_MIX,        // fma, mul, add, fmac round robin (what tile_kernel's arithmetic looks like)
  CNDMASK_E64,    // v_cndmask_b32_e64 vN, vN, v48, s[20:21]
  MAX,            // v_max_f32
  RNDNE,          // v_rndne_f32
  MUL_SGPR,       // v_mul_f32 vN, s20, vN          (VOP2 with an SGPR source)
  MUL_E64_NEG,    // v_mul_f32_e64 vN, -vN, v48     (VOP3 encoding because of the modifier)
  FMA_NEG,        // v_fma_f32 vN, -vN, v48, v49
  SUB,            // v_sub_f32
  MIN,            // v_min_f32
  CMP_VCC,        // v_cmp_gt_f32 vcc, vN, v48
  CNDMASK_VCC,    // v_cndmask_b32 vN, vN, v48, vcc (VOP2)
  CVT_F32_I32,    // v_cvt_f32_i32
  CVT_I32_F32,    // v_cvt_i32_f32
  AND_B32,        // v_and_b32
  LSHL,           // v_lshlrev_b32
  ADD_U32,        // v_add_u32
  MUL_LO_U32,     // v_mul_lo_u32
  RCP,            // v_rcp_f32
  SQRT,           // v_sqrt_f32
  MOV_DPP,        // v_mov_b32_dpp quad_perm
  ADD_DPP,        // v_add_f32_dpp row_shr:1
  MED3,           // v_med3_f32 (clamp in one instruction)
  FMA_CLAMP,      // v_fma_f32 ... clamp
  NUM_MODES
};
static const char* kModeNames[NUM_MODES] = {
    "v_fma_f32 3 VGPR sources, distinct", "v_fma_f32 3 VGPR sources, same bank", "v_fma_f32 one VGPR three times",
    "v_fma_f32 with SGPR source", "v_mul_f32 with 32-bit literal", "v_fma_f32 with inline constant",
    "v_fmac_f32 (VOP2)", "v_mul_f32 (VOP2)", "v_add_f32 (VOP2)", "v_mov_b32", "v_pk_fma_f32 (2 fma per lane)",
    "v_fma_f32 1 dependent chain", "v_fma_f32 2 chains", "v_fma_f32 4 chains", "fma/mul/add/fmac mix, 16 chains",
    "v_cndmask_b32_e64 SGPR mask", "v_max_f32", "v_rndne_f32", "v_mul_f32 with SGPR source (VOP2)",
    "v_mul_f32_e64 with neg modifier", "v_fma_f32 with neg modifier", "v_sub_f32", "v_min_f32", "v_cmp_gt_f32 -> vcc",
    "v_cndmask_b32 vcc (VOP2)", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_and_b32", "v_lshlrev_b32", "v_add_u32",
    "v_mul_lo_u32", "v_rcp_f32", "v_sqrt_f32", "v_mov_b32_dpp quad_perm", "v_add_f32_dpp row_shr:1", "v_med3_f32",
    "v_fma_f32 with clamp"};

template <int MODE>
__global__ void __launch_bounds__(512) probe(unsigned long long* cycles, float* sink, int iters) {
  extern __shared__ float lds_footprint[];
  if (iters < 0) lds_footprint[threadIdx.x] = 0.0f;  // (never; keeps the allocation)
  float a = 1.0001f, b = 0.5f;
  unsigned long long t0, t1, r0, r1;
  asm volatile(
      "s_mov_b32 s20, 0x3f800347\n\t"
      "s_mov_b32 s21, 0x33333333\n\t"
      "v_mov_b32 v48, %0\n\t v_mov_b32 v49, %1\n\t v_mov_b32 v50, %0\n\t v_mov_b32 v51, %1\n\t"
      "v_mov_b32 v32, 1.0\n\t v_mov_b32 v33, 1.0\n\t v_mov_b32 v34, 1.0\n\t v_mov_b32 v35, 1.0\n\t"
      "v_mov_b32 v36, 1.0\n\t v_mov_b32 v37, 1.0\n\t v_mov_b32 v38, 1.0\n\t v_mov_b32 v39, 1.0\n\t"
      "v_mov_b32 v40, 1.0\n\t v_mov_b32 v41, 1.0\n\t v_mov_b32 v42, 1.0\n\t v_mov_b32 v43, 1.0\n\t"
      "v_mov_b32 v44, 1.0\n\t v_mov_b32 v45, 1.0\n\t v_mov_b32 v46, 1.0\n\t v_mov_b32 v47, 1.0\n\t" ::"v"(a),
      "v"(b)
      : "s20", "s21", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45",
        "v46", "v47", "v48", "v49", "v50", "v51");
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#define CLOB : : : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"
#define S16(OP, TAIL)                                                                                              \
  asm volatile(OP " v32, v32" TAIL "\n\t" OP " v33, v33" TAIL "\n\t" OP " v34, v34" TAIL "\n\t" OP " v35, v35" TAIL \
                  "\n\t" OP " v36, v36" TAIL "\n\t" OP " v37, v37" TAIL "\n\t" OP " v38, v38" TAIL "\n\t" OP        \
                  " v39, v39" TAIL "\n\t" OP " v40, v40" TAIL "\n\t" OP " v41, v41" TAIL "\n\t" OP " v42, v42" TAIL \
                  "\n\t" OP " v43, v43" TAIL "\n\t" OP " v44, v44" TAIL "\n\t" OP " v45, v45" TAIL "\n\t" OP        \
                  " v46, v46" TAIL "\n\t" OP " v47, v47" TAIL CLOB)
  for (int it = 0; it < iters; it++) {
#pragma unroll
   for (int u = 0; u < 16; u++) {
    if (MODE == FMA_DISTINCT) { S16("v_fma_f32", ", v48, v49"); }
    if (MODE == FMA_SAMEBANK) {
      asm volatile(REP4("v_fma_f32 v32, v32, v36, v40\n\t v_fma_f32 v36, v36, v40, v44\n\t"
                        "v_fma_f32 v40, v40, v44, v32\n\t v_fma_f32 v44, v44, v32, v36\n\t") CLOB);
    }
    if (MODE == FMA_REPEAT) {
      asm volatile("v_fma_f32 v32, v32, v32, v32\n\t v_fma_f32 v33, v33, v33, v33\n\t v_fma_f32 v34, v34, v34, v34\n\t"
                   "v_fma_f32 v35, v35, v35, v35\n\t v_fma_f32 v36, v36, v36, v36\n\t v_fma_f32 v37, v37, v37, v37\n\t"
                   "v_fma_f32 v38, v38, v38, v38\n\t v_fma_f32 v39, v39, v39, v39\n\t v_fma_f32 v40, v40, v40, v40\n\t"
                   "v_fma_f32 v41, v41, v41, v41\n\t v_fma_f32 v42, v42, v42, v42\n\t v_fma_f32 v43, v43, v43, v43\n\t"
                   "v_fma_f32 v44, v44, v44, v44\n\t v_fma_f32 v45, v45, v45, v45\n\t v_fma_f32 v46, v46, v46, v46\n\t"
                   "v_fma_f32 v47, v47, v47, v47" CLOB);
    }
    if (MODE == FMA_SGPR) { S16("v_fma_f32", ", s20, v49"); }
    if (MODE == FMA_LITERAL) { asm volatile("v_mul_f32 v32, 0x3f800347, v32\n\tv_mul_f32 v33, 0x3f800347, v33\n\tv_mul_f32 v34, 0x3f800347, v34\n\tv_mul_f32 v35, 0x3f800347, v35\n\tv_mul_f32 v36, 0x3f800347, v36\n\tv_mul_f32 v37, 0x3f800347, v37\n\tv_mul_f32 v38, 0x3f800347, v38\n\tv_mul_f32 v39, 0x3f800347, v39\n\tv_mul_f32 v40, 0x3f800347, v40\n\tv_mul_f32 v41, 0x3f800347, v41\n\tv_mul_f32 v42, 0x3f800347, v42\n\tv_mul_f32 v43, 0x3f800347, v43\n\tv_mul_f32 v44, 0x3f800347, v44\n\tv_mul_f32 v45, 0x3f800347, v45\n\tv_mul_f32 v46, 0x3f800347, v46\n\tv_mul_f32 v47, 0x3f800347, v47" CLOB); }  // (VOP2: the literal is src0)
    if (MODE == FMA_INLINE) { S16("v_fma_f32", ", 1.0, v49"); }
    if (MODE == MUL) { S16("v_mul_f32", ", v48"); }
    if (MODE == ADD) { S16("v_add_f32", ", v48"); }
    if (MODE == MAX) { S16("v_max_f32", ", v48"); }
    if (MODE == CNDMASK_E64) { S16("v_cndmask_b32_e64", ", v48, s[20:21]"); }
    if (MODE == FMAC) {
      asm volatile("v_fmac_f32 v32, v48, v49\n\t v_fmac_f32 v33, v48, v49\n\t v_fmac_f32 v34, v48, v49\n\t"
                   "v_fmac_f32 v35, v48, v49\n\t v_fmac_f32 v36, v48, v49\n\t v_fmac_f32 v37, v48, v49\n\t"
                   "v_fmac_f32 v38, v48, v49\n\t v_fmac_f32 v39, v48, v49\n\t v_fmac_f32 v40, v48, v49\n\t"
                   "v_fmac_f32 v41, v48, v49\n\t v_fmac_f32 v42, v48, v49\n\t v_fmac_f32 v43, v48, v49\n\t"
                   "v_fmac_f32 v44, v48, v49\n\t v_fmac_f32 v45, v48, v49\n\t v_fmac_f32 v46, v48, v49\n\t"
                   "v_fmac_f32 v47, v48, v49" CLOB);
    }
    if (MODE == MOV) {
      asm volatile(REP4("v_mov_b32 v32, v48\n\t v_mov_b32 v33, v49\n\t v_mov_b32 v34, v48\n\t v_mov_b32 v35, v49\n\t") CLOB);
    }
    if (MODE == RNDNE) {
      asm volatile("v_rndne_f32 v32, v32\n\t v_rndne_f32 v33, v33\n\t v_rndne_f32 v34, v34\n\t v_rndne_f32 v35, v35\n\t"
                   "v_rndne_f32 v36, v36\n\t v_rndne_f32 v37, v37\n\t v_rndne_f32 v38, v38\n\t v_rndne_f32 v39, v39\n\t"
                   "v_rndne_f32 v40, v40\n\t v_rndne_f32 v41, v41\n\t v_rndne_f32 v42, v42\n\t v_rndne_f32 v43, v43\n\t"
                   "v_rndne_f32 v44, v44\n\t v_rndne_f32 v45, v45\n\t v_rndne_f32 v46, v46\n\t v_rndne_f32 v47, v47" CLOB);
    }
    if (MODE == PK_FMA) {
      asm volatile(REP4("v_pk_fma_f32 v[32:33], v[32:33], v[48:49], v[50:51]\n\t"
                        "v_pk_fma_f32 v[34:35], v[34:35], v[48:49], v[50:51]\n\t"
                        "v_pk_fma_f32 v[36:37], v[36:37], v[48:49], v[50:51]\n\t"
                        "v_pk_fma_f32 v[38:39], v[38:39], v[48:49], v[50:51]\n\t") CLOB);
    }
    if (MODE == CHAIN1) { asm volatile(REP16("v_fma_f32 v32, v32, v48, v49\n\t") CLOB); }
    if (MODE == CHAIN2) {
      asm volatile(REP4("v_fma_f32 v32, v32, v48, v49\n\t v_fma_f32 v33, v33, v48, v49\n\t"
                        "v_fma_f32 v32, v32, v48, v49\n\t v_fma_f32 v33, v33, v48, v49\n\t") CLOB);
    }
    if (MODE == CHAIN4) {
      asm volatile(REP4("v_fma_f32 v32, v32, v48, v49\n\t v_fma_f32 v33, v33, v48, v49\n\t"
                        "v_fma_f32 v34, v34, v48, v49\n\t v_fma_f32 v35, v35, v48, v49\n\t") CLOB);
    }
    if (MODE == MUL_SGPR) { asm volatile("v_mul_f32 v32, s20, v32\n\tv_mul_f32 v33, s20, v33\n\tv_mul_f32 v34, s20, v34\n\tv_mul_f32 v35, s20, v35\n\tv_mul_f32 v36, s20, v36\n\tv_mul_f32 v37, s20, v37\n\tv_mul_f32 v38, s20, v38\n\tv_mul_f32 v39, s20, v39\n\tv_mul_f32 v40, s20, v40\n\tv_mul_f32 v41, s20, v41\n\tv_mul_f32 v42, s20, v42\n\tv_mul_f32 v43, s20, v43\n\tv_mul_f32 v44, s20, v44\n\tv_mul_f32 v45, s20, v45\n\tv_mul_f32 v46, s20, v46\n\tv_mul_f32 v47, s20, v47" CLOB); }
    if (MODE == MUL_E64_NEG) { asm volatile("v_mul_f32_e64 v32, -v32, v48\n\tv_mul_f32_e64 v33, -v33, v48\n\tv_mul_f32_e64 v34, -v34, v48\n\tv_mul_f32_e64 v35, -v35, v48\n\tv_mul_f32_e64 v36, -v36, v48\n\tv_mul_f32_e64 v37, -v37, v48\n\tv_mul_f32_e64 v38, -v38, v48\n\tv_mul_f32_e64 v39, -v39, v48\n\tv_mul_f32_e64 v40, -v40, v48\n\tv_mul_f32_e64 v41, -v41, v48\n\tv_mul_f32_e64 v42, -v42, v48\n\tv_mul_f32_e64 v43, -v43, v48\n\tv_mul_f32_e64 v44, -v44, v48\n\tv_mul_f32_e64 v45, -v45, v48\n\tv_mul_f32_e64 v46, -v46, v48\n\tv_mul_f32_e64 v47, -v47, v48" CLOB); }
    if (MODE == FMA_NEG) { asm volatile("v_fma_f32 v32, -v32, v48, v49\n\tv_fma_f32 v33, -v33, v48, v49\n\tv_fma_f32 v34, -v34, v48, v49\n\tv_fma_f32 v35, -v35, v48, v49\n\tv_fma_f32 v36, -v36, v48, v49\n\tv_fma_f32 v37, -v37, v48, v49\n\tv_fma_f32 v38, -v38, v48, v49\n\tv_fma_f32 v39, -v39, v48, v49\n\tv_fma_f32 v40, -v40, v48, v49\n\tv_fma_f32 v41, -v41, v48, v49\n\tv_fma_f32 v42, -v42, v48, v49\n\tv_fma_f32 v43, -v43, v48, v49\n\tv_fma_f32 v44, -v44, v48, v49\n\tv_fma_f32 v45, -v45, v48, v49\n\tv_fma_f32 v46, -v46, v48, v49\n\tv_fma_f32 v47, -v47, v48, v49" CLOB); }
    if (MODE == SUB) { S16("v_sub_f32", ", v48"); }
    if (MODE == MIN) { S16("v_min_f32", ", v48"); }
    if (MODE == CMP_VCC) { asm volatile("v_cmp_gt_f32 vcc, v32, v48\n\tv_cmp_gt_f32 vcc, v33, v48\n\tv_cmp_gt_f32 vcc, v34, v48\n\tv_cmp_gt_f32 vcc, v35, v48\n\tv_cmp_gt_f32 vcc, v36, v48\n\tv_cmp_gt_f32 vcc, v37, v48\n\tv_cmp_gt_f32 vcc, v38, v48\n\tv_cmp_gt_f32 vcc, v39, v48\n\tv_cmp_gt_f32 vcc, v40, v48\n\tv_cmp_gt_f32 vcc, v41, v48\n\tv_cmp_gt_f32 vcc, v42, v48\n\tv_cmp_gt_f32 vcc, v43, v48\n\tv_cmp_gt_f32 vcc, v44, v48\n\tv_cmp_gt_f32 vcc, v45, v48\n\tv_cmp_gt_f32 vcc, v46, v48\n\tv_cmp_gt_f32 vcc, v47, v48" : : : "vcc"); }
    if (MODE == CNDMASK_VCC) { S16("v_cndmask_b32", ", v48, vcc"); }
    if (MODE == CVT_F32_I32) { asm volatile("v_cvt_f32_i32 v32, v32\n\tv_cvt_f32_i32 v33, v33\n\tv_cvt_f32_i32 v34, v34\n\tv_cvt_f32_i32 v35, v35\n\tv_cvt_f32_i32 v36, v36\n\tv_cvt_f32_i32 v37, v37\n\tv_cvt_f32_i32 v38, v38\n\tv_cvt_f32_i32 v39, v39\n\tv_cvt_f32_i32 v40, v40\n\tv_cvt_f32_i32 v41, v41\n\tv_cvt_f32_i32 v42, v42\n\tv_cvt_f32_i32 v43, v43\n\tv_cvt_f32_i32 v44, v44\n\tv_cvt_f32_i32 v45, v45\n\tv_cvt_f32_i32 v46, v46\n\tv_cvt_f32_i32 v47, v47" CLOB); }
    if (MODE == CVT_I32_F32) { asm volatile("v_cvt_i32_f32 v32, v32\n\tv_cvt_i32_f32 v33, v33\n\tv_cvt_i32_f32 v34, v34\n\tv_cvt_i32_f32 v35, v35\n\tv_cvt_i32_f32 v36, v36\n\tv_cvt_i32_f32 v37, v37\n\tv_cvt_i32_f32 v38, v38\n\tv_cvt_i32_f32 v39, v39\n\tv_cvt_i32_f32 v40, v40\n\tv_cvt_i32_f32 v41, v41\n\tv_cvt_i32_f32 v42, v42\n\tv_cvt_i32_f32 v43, v43\n\tv_cvt_i32_f32 v44, v44\n\tv_cvt_i32_f32 v45, v45\n\tv_cvt_i32_f32 v46, v46\n\tv_cvt_i32_f32 v47, v47" CLOB); }
    if (MODE == AND_B32) { S16("v_and_b32", ", v48"); }
    if (MODE == LSHL) { asm volatile("v_lshlrev_b32 v32, 1, v32\n\tv_lshlrev_b32 v33, 1, v33\n\tv_lshlrev_b32 v34, 1, v34\n\tv_lshlrev_b32 v35, 1, v35\n\tv_lshlrev_b32 v36, 1, v36\n\tv_lshlrev_b32 v37, 1, v37\n\tv_lshlrev_b32 v38, 1, v38\n\tv_lshlrev_b32 v39, 1, v39\n\tv_lshlrev_b32 v40, 1, v40\n\tv_lshlrev_b32 v41, 1, v41\n\tv_lshlrev_b32 v42, 1, v42\n\tv_lshlrev_b32 v43, 1, v43\n\tv_lshlrev_b32 v44, 1, v44\n\tv_lshlrev_b32 v45, 1, v45\n\tv_lshlrev_b32 v46, 1, v46\n\tv_lshlrev_b32 v47, 1, v47" CLOB); }
    if (MODE == ADD_U32) { S16("v_add_u32", ", v48"); }
    if (MODE == MUL_LO_U32) { S16("v_mul_lo_u32", ", v48"); }
    if (MODE == RCP) { asm volatile("v_rcp_f32 v32, v32\n\tv_rcp_f32 v33, v33\n\tv_rcp_f32 v34, v34\n\tv_rcp_f32 v35, v35\n\tv_rcp_f32 v36, v36\n\tv_rcp_f32 v37, v37\n\tv_rcp_f32 v38, v38\n\tv_rcp_f32 v39, v39\n\tv_rcp_f32 v40, v40\n\tv_rcp_f32 v41, v41\n\tv_rcp_f32 v42, v42\n\tv_rcp_f32 v43, v43\n\tv_rcp_f32 v44, v44\n\tv_rcp_f32 v45, v45\n\tv_rcp_f32 v46, v46\n\tv_rcp_f32 v47, v47" CLOB); }
    if (MODE == SQRT) { asm volatile("v_sqrt_f32 v32, v32\n\tv_sqrt_f32 v33, v33\n\tv_sqrt_f32 v34, v34\n\tv_sqrt_f32 v35, v35\n\tv_sqrt_f32 v36, v36\n\tv_sqrt_f32 v37, v37\n\tv_sqrt_f32 v38, v38\n\tv_sqrt_f32 v39, v39\n\tv_sqrt_f32 v40, v40\n\tv_sqrt_f32 v41, v41\n\tv_sqrt_f32 v42, v42\n\tv_sqrt_f32 v43, v43\n\tv_sqrt_f32 v44, v44\n\tv_sqrt_f32 v45, v45\n\tv_sqrt_f32 v46, v46\n\tv_sqrt_f32 v47, v47" CLOB); }
    if (MODE == MOV_DPP) { asm volatile("v_mov_b32_dpp v32, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v33, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v34, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v35, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v36, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v37, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v38, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v39, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v40, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v41, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v42, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v43, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v44, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v45, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v46, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v47, v48 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" CLOB); }
    if (MODE == ADD_DPP) { asm volatile("v_add_f32_dpp v32, v48, v32 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v33, v48, v33 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v34, v48, v34 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v35, v48, v35 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v36, v48, v36 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v37, v48, v37 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v38, v48, v38 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v39, v48, v39 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v40, v48, v40 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v41, v48, v41 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v42, v48, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v43, v48, v43 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v44, v48, v44 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v45, v48, v45 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v46, v48, v46 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v47, v48, v47 row_shr:1 row_mask:0xf bank_mask:0xf" CLOB); }
    if (MODE == MED3) { S16("v_med3_f32", ", v48, v49"); }
    if (MODE == FMA_CLAMP) { asm volatile("v_fma_f32 v32, v32, v48, v49 clamp\n\tv_fma_f32 v33, v33, v48, v49 clamp\n\tv_fma_f32 v34, v34, v48, v49 clamp\n\tv_fma_f32 v35, v35, v48, v49 clamp\n\tv_fma_f32 v36, v36, v48, v49 clamp\n\tv_fma_f32 v37, v37, v48, v49 clamp\n\tv_fma_f32 v38, v38, v48, v49 clamp\n\tv_fma_f32 v39, v39, v48, v49 clamp\n\tv_fma_f32 v40, v40, v48, v49 clamp\n\tv_fma_f32 v41, v41, v48, v49 clamp\n\tv_fma_f32 v42, v42, v48, v49 clamp\n\tv_fma_f32 v43, v43, v48, v49 clamp\n\tv_fma_f32 v44, v44, v48, v49 clamp\n\tv_fma_f32 v45, v45, v48, v49 clamp\n\tv_fma_f32 v46, v46, v48, v49 clamp\n\tv_fma_f32 v47, v47, v48, v49 clamp" CLOB); }
    if (MODE == FMA_MIX) {
      asm volatile(REP4("v_fma_f32 v32, v32, v48, v49\n\t v_mul_f32 v33, v33, v48\n\t"
                        "v_add_f32 v34, v34, v49\n\t v_fmac_f32 v35, v48, v49\n\t") CLOB);
      // (4 chains x 4: the register names repeat, which is what a 4-deep unrolled loop looks like)
    }
   }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
  float s;
  asm volatile("v_add_f32 %0, v32, v33\n\t v_add_f32 %0, %0, v34\n\t v_add_f32 %0, %0, v35\n\t v_add_f32 %0, %0, v40\n\t"
               "v_add_f32 %0, %0, v44\n\t v_add_f32 %0, %0, v47"
               : "=v"(s));
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  sink[gid] = s;
  if ((threadIdx.x & 63) == 0) {
    cycles[4 * (gid >> 6) + 0] = t1 - t0;  // shader cycles of this wave's stream
    cycles[4 * (gid >> 6) + 1] = r1 - r0;  // the same span in ticks of the constant 100 MHz counter
    cycles[4 * (gid >> 6) + 2] = r0;
    cycles[4 * (gid >> 6) + 3] = r1;
  }
}

template <int MODE>
void run(int waves_per_simd, unsigned long long* d_cycles, float* d_sink, FILE* json, bool* first) {
  const int iters = 600;  // x 256 instructions
  // W waves on each of a CU's 4 SIMDs: W / 2 workgroups of 512 threads (tile_kernel's shape) per CU, pinned
  // there by their LDS footprint (160 KB per CU: exactly W / 2 of them fit), 256 x W / 2 workgroups in all
  const int per_cu = waves_per_simd / 2;
  const int threads = 512;
  const int blocks = 256 * per_cu;
  const size_t lds = per_cu == 1 ? 100 * 1024 : per_cu == 2 ? 64 * 1024 : per_cu == 3 ? 48 * 1024 : 36 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nwaves = blocks * threads / 64;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), lds, 0, d_cycles, d_sink, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), lds, 0, d_cycles, d_sink, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(4 * nwaves);
  hipMemcpy(c.data(), d_cycles, 4 * nwaves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::vector<double> cyc(nwaves), ghz(nwaves);
  unsigned long long first_end = ~0ull, last_start = 0, first_start = ~0ull, last_end = 0;
  for (int i = 0; i < nwaves; i++) {
    cyc[i] = (double)c[4 * i];
    ghz[i] = (double)c[4 * i] / ((double)c[4 * i + 1] * 10e-9) / 1e9;  // 100 MHz ticks
    first_start = std::min(first_start, c[4 * i + 2]);
    last_start = std::max(last_start, c[4 * i + 2]);
    first_end = std::min(first_end, c[4 * i + 3]);
    last_end = std::max(last_end, c[4 * i + 3]);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  const double instr = (double)iters * 256;
  const double clock = ghz[nwaves / 2];
  // (a) what a wave saw: its own stream's cycles / (W x instructions).  The SIMD arbitrates by age, so older
  //     waves finish early and this UNDER-states the cost when W > 2.
  const double per_wave = cyc[nwaves / 2] / (instr * waves_per_simd);
  // (b) what the SIMD delivered: the chip-wide span first start -> last end (100 MHz ticks x measured clock) over
  //     the W x instructions every SIMD issued.  Includes the launch ramp (a few microseconds of ~1 ms).
  const double span_cycles = (double)(last_end - first_start) * 10e-9 * clock * 1e9;
  const double per_simd = span_cycles / (instr * waves_per_simd);
  printf("%-40s W=%d  %6.3f cycles/instr/SIMD (span)  %6.3f (median wave)  clock %5.3f GHz  (%7.3f ms)\n",
         kModeNames[MODE], waves_per_simd, per_simd, per_wave, clock, ms);
  fprintf(json, "%s{\"mode\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr\": %.4f, "
                "\"cycles_per_instr_median_wave\": %.4f, \"clock_ghz\": %.4f, \"kernel_ms\": %.4f}",
          *first ? "" : ",\n  ", kModeNames[MODE], waves_per_simd, per_simd, per_wave, clock, ms);
  *first = false;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int MODE>
void run_all(unsigned long long* d_cycles, float* d_sink, FILE* json, bool* first, bool sweep) {
  if (sweep) {
    for (int w : {2, 4, 6, 8}) run<MODE>(w, d_cycles, d_sink, json, first);
  } else {
    for (int w : {4, 8}) run<MODE>(w, d_cycles, d_sink, json, first);
  }
}

int main(int argc, char** argv) {
  unsigned long long* d_cycles;
  float* d_sink;
  hipMalloc(&d_cycles, 4 * 256 * 32 * sizeof(unsigned long long));  // 4 values per wave, <= 32 waves per CU
  hipMalloc(&d_sink, 256 * 2048 * sizeof(float));
  FILE* json = fopen(argc > 1 ? argv[1] : "/dev/null", "w");
  if (!json) return 1;
  fprintf(json, "{\"what\": \"shader cycles (s_memtime) per wave64 VALU instruction and SIMD; W waves per SIMD on every SIMD of the chip; clock = s_memtime / s_memrealtime (100 MHz) over the same span; 256-instruction loop bodies\",\n \"rows\": [\n  ");
  bool first = true;
  run_all<FMA_DISTINCT>(d_cycles, d_sink, json, &first, true);
  run_all<FMA_SAMEBANK>(d_cycles, d_sink, json, &first, false);
  run_all<FMA_REPEAT>(d_cycles, d_sink, json, &first, false);
  run_all<FMA_SGPR>(d_cycles, d_sink, json, &first, false);
  run_all<FMA_LITERAL>(d_cycles, d_sink, json, &first, false);
  run_all<FMA_INLINE>(d_cycles, d_sink, json, &first, false);
  run_all<FMAC>(d_cycles, d_sink, json, &first, false);
  run_all<MUL>(d_cycles, d_sink, json, &first, false);
  run_all<ADD>(d_cycles, d_sink, json, &first, false);
  run_all<MOV>(d_cycles, d_sink, json, &first, false);
  run_all<PK_FMA>(d_cycles, d_sink, json, &first, true);
  run_all<CHAIN1>(d_cycles, d_sink, json, &first, true);
  run_all<CHAIN2>(d_cycles, d_sink, json, &first, false);
  run_all<CHAIN4>(d_cycles, d_sink, json, &first, true);
  run_all<FMA_MIX>(d_cycles, d_sink, json, &first, false);
  run_all<CNDMASK_E64>(d_cycles, d_sink, json, &first, false);
  run_all<MAX>(d_cycles, d_sink, json, &first, false);
  run_all<RNDNE>(d_cycles, d_sink, json, &first, false);
  run_all<MUL_SGPR>(d_cycles, d_sink, json, &first, false);
  run_all<MUL_E64_NEG>(d_cycles, d_sink, json, &first, false);
  run_all<FMA_NEG>(d_cycles, d_sink, json, &first, false);
  run_all<FMA_CLAMP>(d_cycles, d_sink, json, &first, false);
  run_all<SUB>(d_cycles, d_sink, json, &first, false);
  run_all<MIN>(d_cycles, d_sink, json, &first, false);
  run_all<MED3>(d_cycles, d_sink, json, &first, false);
  run_all<CMP_VCC>(d_cycles, d_sink, json, &first, false);
  run_all<CNDMASK_VCC>(d_cycles, d_sink, json, &first, false);
  run_all<CVT_F32_I32>(d_cycles, d_sink, json, &first, false);
  run_all<CVT_I32_F32>(d_cycles, d_sink, json, &first, false);
  run_all<AND_B32>(d_cycles, d_sink, json, &first, false);
  run_all<LSHL>(d_cycles, d_sink, json, &first, false);
  run_all<ADD_U32>(d_cycles, d_sink, json, &first, false);
  run_all<MUL_LO_U32>(d_cycles, d_sink, json, &first, false);
  run_all<RCP>(d_cycles, d_sink, json, &first, false);
  run_all<SQRT>(d_cycles, d_sink, json, &first, false);
  run_all<MOV_DPP>(d_cycles, d_sink, json, &first, false);
  run_all<ADD_DPP>(d_cycles, d_sink, json, &first, false);
  fprintf(json, "\n ]}\n");
  fclose(json);
  return 0;
}
