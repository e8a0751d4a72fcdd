// jxlt_device.h -- gfx950 device code of the JPEG XL tiny encoder hot path.
//
// Pure device code (kernels + __device__ helpers); the launches live in
// jxlt_capi.hip.  tests/ compile this same file against a fiber-based HIP
// execution model on the CPU (tests/hipsim) to check it bit-for-bit against the
// oracle without a GPU; the product only ever builds it with hipcc.
//
// MUST be compiled with -ffp-contract=off: the arithmetic model is "every
// operation one IEEE binary32 operation, fused only where __builtin_fmaf is
// written" (SURVEY.md Appendix B, 8-lane canonical model).  The 8 SIMD lanes of
// the reference map onto 8 adjacent GPU lanes ("octets"); SumOfLanes becomes a
// 3-step xor butterfly inside the octet, which reproduces the halving tree.
//
// Kernels:
//   tile_kernel      one 512-thread workgroup per 64x64 tile: edge-replicated
//                    load + XYB -> LDS, adaptive quant field, chroma-from-luma,
//                    DCT8/16x8/8x16 strategy search, quantisation, DC, nzeros;
//                    writes side-band grids + scan-ordered quantised coefficients.
//                    (ref: enc_frame.cc:597-683 + enc_group.cc:304-443)
//   group_scan_kernel exclusive scan of per-group token counts.
//   token_kernel     one workgroup per 256x256 group: context modelling and raw
//                    3-byte token records in stream order (ref: enc_group.cc:444-494)
#ifndef JXLT_DEVICE_H_
#define JXLT_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

namespace jxlt_dev {

// ---------------------------------------------------------------------------
// Kernel arguments
// ---------------------------------------------------------------------------

// Constant tables, resident in HBM (built on the host by jxlt_capi.hip).
struct DeviceTables {
  float weights[576];      // dequant weights (quant_weights.cc:17-134)
  float inv_weights[576];  // float(1.0 / w), LLF zeroed (quant_weights.cc:144-153)
  float inv_qac[256];      // float(1.0 / (double)(scale * q)) (enc_group.cc:289)
  uint16_t table_offset[9];  // host copy of quant_table_offset() below (checked when the tables are built)
  uint8_t coeff_order[192];
  uint16_t freq_context[64];
  uint16_t nnz_context[64];
  uint8_t block_context_map[81];
  uint8_t ac_context_map[1980];
  uint8_t gradient_lut[1024];  // enc_frame.cc:226-281
  float sqrt_lut[1024];        // sqrtf(i), correctly rounded (EstimateEntropy's cost of a coefficient)
  // What the quantisation needs to know of scan position p (tile_kernel quantises in scan order, lane = scan
  // position), per position class -- 0: DCT8, 1 / 2: first / second 64 positions of a two-block transform:
  // [0..2] InvMatrix of x, y, b at the position's coefficient, [3] dequantisation weight of y, [4..6] zeroing
  // threshold of x, y, b (enc_group.cc:227-242); scan_slot: where the staging area keeps that coefficient
  // (bit 6: in the transform's second block).
  float scan_consts[3][7][64];
  uint8_t scan_slot[3][64];
};

// The quantiser's zeroing threshold (enc_group.cc:227-242) of channel c in quadrant `quad` of a one-block
// (8x8) or two-block transform; quadrants: 8x8: (row >= 4) * 2 + (column >= 4); two-block, coefficient
// index i = r * 8 + l with r = 0..15: (r >= 8) * 2 + (r & 1).
__host__ __device__ inline float quant_zeroing_threshold(int c, bool two_block, int quad) {
  float t0 = 0.58f;
  float t1 = c == 0 ? 0.635f + 0.08f : c == 2 ? 0.75f : 0.635f;
  float t2 = c == 0 ? 0.66f + 0.08f : c == 2 ? 0.75f : 0.66f;
  float t3 = c == 0 ? 0.7f + 0.08f : c == 2 ? 0.75f : 0.7f;
  if (two_block) {
    const float dec = 0.003f * 2 * 1;  // Clamp1(0.003f*xsize*ysize, 0, 0.08|0.12)
    t0 -= dec; t1 -= dec; t2 -= dec; t3 -= dec;
  }
  return quad == 0 ? t0 : quad == 1 ? t1 : quad == 2 ? t2 : t3;
}

// Offset of quant table n = strategy * 3 + channel inside weights[] / inv_weights[]: three
// 64-entry DCT8 tables, then three 128-entry tables shared by DCT16X8 and DCT8X16.
__host__ __device__ constexpr int quant_table_offset(int n) { return n < 3 ? n * 64 : 192 + ((n - 3) % 3) * 128; }

// Entries of the square-root table of the entropy estimate (a power of two; tests build the CPU
// model with a tiny table to exercise the overflow path on ordinary images).
#ifndef JXLT_SQRT_LUT_SIZE
#define JXLT_SQRT_LUT_SIZE 1024
#endif
constexpr int kSqrtLutSize = JXLT_SQRT_LUT_SIZE;  // DeviceTables::sqrt_lut, TileShared::sqrt_lut
static_assert((kSqrtLutSize & (kSqrtLutSize - 1)) == 0 && kSqrtLutSize <= 1024, "power of two, fits DeviceTables");

struct FrameGeom {
  int xsize, ysize;                // pixels
  int xsize_blocks, ysize_blocks;  // 8x8
  int xsize_tiles, ysize_tiles;    // 64x64
  int xsize_groups, ysize_groups;  // 256x256
};

struct TileArgs {
  // Input samples: sample (c, y, x) is planes[c][y * pitch + x * pix_stride].  Planar frames use
  // three base pointers and pix_stride 1; a raw PFM payload (read_pfm.cc:199-209: interleaved
  // RGB, bottom row first, possibly big endian) is one buffer with pix_stride 3, base pointers
  // one float apart that point at its LAST row, a negative pitch, and byteswap set.
  const float* planes[3];
  ptrdiff_t pitch;  // floats per row (may be negative)
  int pix_stride;   // floats between horizontally adjacent samples of a plane
  int byteswap;     // samples are stored byte-reversed
  FrameGeom g;
  float distance, scale, inv_scale, scale_dc;
  float x_qm_mul;  // 1.25^(x_qm_scale-2)
  float strategy_distance;  // distance behind mul8x8 / mul16x8 (enc_ac_strategy.cc:178-185: the
                            // reference freezes them at its first call; normally == distance)
  uint32_t flags;  // bit0: force DCT8
  const DeviceTables* tab;
  // outputs (image-absolute grids)
  int16_t* quant_dc[3];
  uint8_t* raw_quant;
  uint8_t* strategy;
  int8_t* ytox;
  int8_t* ytob;
  uint8_t* nzgrid[3];   // value used for context prediction, per block & channel
  uint8_t* blk_nz;      // [block*3 + c]: number of nonzeros (token value)
  uint8_t* blk_nscan;   // [block*3 + c]: scan positions up to the last nonzero
  unsigned long long* blk_nzmask;  // [block*3 + c][2]: which of the scan positions covered .. 127 are nonzero
  int16_t* coef_scan;   // [block*3 + c][64] quantised coefficients in scan order
  uint32_t* group_ntok; // per group token count (atomic)
  uint32_t* dc_nac;     // per DC group: number of first blocks (atomic)
  uint32_t* lut_overflow;  // [1] set when a quantised magnitude did not fit the root table
  // debug (may be null)
  float* dbg_xyb[3];
  float* dbg_qf;
  float* dbg_mask;
  float* dbg_ent8;
  unsigned long long* dbg_phase;  // [16] accumulated shader cycles per phase (thread 0 of each tile)
};

struct TokenArgs {
  FrameGeom g;
  const DeviceTables* tab;
  const uint8_t* strategy;
  const uint8_t* nzgrid[3];
  const uint8_t* blk_nz;
  const uint8_t* blk_nscan;
  const unsigned long long* blk_nzmask;
  const int16_t* coef_scan;
  const uint32_t* group_ntok;         // tokens of every group (tile_kernel's counts)
  uint64_t* group_tok_offset;        // [groups + 1] OUT: exclusive scan of group_ntok -- every workgroup sums the
                                     // counts of the groups before its own (no scan kernel in front of this one)
  uint8_t* tokens;                   // 3 bytes per token
  uint32_t* histogram;               // optional [64 pre-clusters][64 symbols] (enc_frame.cc:767-782)
  int group_first;                   // workgroup b handles group group_first + b (launches per row of DC groups)
};

// ---------------------------------------------------------------------------
// Arithmetic primitives
// ---------------------------------------------------------------------------

#define JXLT_DI __device__ __forceinline__
#define JXLT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)  // no instruction is scheduled across
// A use of a vector register value at this point of the program (no instruction: the compiler has to have
// waited for the load that produces it).  The CPU execution model of the tests defines this as nothing.
#ifndef JXLT_TOUCH_VGPR
#define JXLT_TOUCH_VGPR(x) asm volatile("" ::"v"(x))
#endif

JXLT_DI float fma32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
JXLT_DI float nfma32(float a, float b, float c) { return __builtin_fmaf(-a, b, c); }
// min(max(x, 0), 1): folds into the clamp modifier of the instruction that produces x.
JXLT_DI float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
JXLT_DI float zero_if_negative(float v) {
  // sign bit set -> +0: as a signed integer every such pattern is negative (one v_max_i32)
  const int bits = __float_as_int(v);
  return __int_as_float(bits < 0 ? 0 : bits);
}
// Value of lane K of the caller's aligned quad (quad_perm:[K,K,K,K]).
template <int K>
JXLT_DI float quad_lane(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xF, 0xF, true));
}
// Cross-lane moves inside an octet use DPP (data-parallel primitives: a VALU move with a
// lane permutation, no LDS round trip).  quad_perm covers xor 1 and xor 2; xor 4 is two
// row shifts by 4 whose bank masks pick the lanes that have a partner in that direction
// (a DPP bank = 4 lanes, a row = 16 lanes, octets never straddle a row).
constexpr int kDppXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowShl4 = 0x104;   // lane i reads lane i + 4
constexpr int kDppRowShr4 = 0x114;   // lane i reads lane i - 4

// value of lane (l ^ S) for S in {1, 2, 4}
template <int S>
JXLT_DI int octet_xor_i(int v) {
  if (S == 4) {
    int t = __builtin_amdgcn_update_dpp(0, v, kDppRowShl4, 0xF, 0x5, false);  // lanes 0-3 of each octet
    return __builtin_amdgcn_update_dpp(t, v, kDppRowShr4, 0xF, 0xA, false);   // lanes 4-7
  }
  return __builtin_amdgcn_update_dpp(0, v, S == 1 ? kDppXor1 : kDppXor2, 0xF, 0xF, true);
}
template <int S>
JXLT_DI float octet_xor(float v) {
  return __int_as_float(octet_xor_i<S>(__float_as_int(v)));
}

// SumOfLanes over the 8 lanes of an octet: (i)+(i^4), (i)+(i^2), (i)+(i^1).
JXLT_DI float octet_sum(float v) {
  v = v + octet_xor<4>(v);
  v = v + octet_xor<2>(v);
  v = v + octet_xor<1>(v);
  return v;
}
JXLT_DI int octet_sum_int(int v) {
  v = v + octet_xor_i<4>(v);
  v = v + octet_xor_i<2>(v);
  v = v + octet_xor_i<1>(v);
  return v;
}
JXLT_DI int ceil_log2_nonzero(uint32_t x) {
  const int fl = 31 - __clz((int)x);
  return (x & (x - 1)) == 0 ? fl : fl + 1;
}
JXLT_DI uint32_t pack_signed(int32_t v) {  // common.h:54-58
  return ((uint32_t)v << 1) ^ (((uint32_t)(~v) >> 31) - 1);
}

// The symbol alone (histograms).  For value >= 16 it is (n << 2) | (the two bits below the leading one) with
// n = floor(log2 value): exactly bits 21.. of the value as a float (exponent n + 127, then the top two
// mantissa bits; values below 2^24 convert exactly), minus 127 << 2.
JXLT_DI uint32_t hybrid_uint_symbol(uint32_t value) {
#ifdef JXLT_SYMBOL_BY_CLZ
  uint32_t sym, nb, eb;
  if (value < 16) return value;
  const uint32_t n = 31u - (uint32_t)__clz((int)value);
  return (n << 2) + ((value - (1u << n)) >> (n - 2));
#else
  const uint32_t hi = (__float_as_uint((float)value) >> 21) - (127u << 2);
  return value < 16 ? value : hi;
#endif
}

// token.h:32-48 (UintCoder::Encode): symbol, number of extra bits, extra bits
JXLT_DI void hybrid_uint(uint32_t value, uint32_t* sym, uint32_t* nbits, uint32_t* bits) {
  if (value < 16) {
    *sym = value;
    *nbits = 0;
    *bits = 0;
  } else {
    const uint32_t n = 31u - (uint32_t)__clz((int)value);
    const uint32_t m = value - (1u << n);
    *sym = (n << 2) + (m >> (n - 2));
    *nbits = n - 2;
    *bits = value & ((1u << (n - 2)) - 1);
  }
}

// Correctly rounded sqrtf for x == 0 or x in [2^-64, 2^64]: the hardware root (<= 1 ulp off)
// plus the usual neighbour test -- the residuals x - s_down*s and x - s_up*s tell whether a
// neighbour is the rounded root.  This is the generic sqrtf expansion minus its input scaling
// and its zero/infinity fix-up, which these argument ranges do not need.  (x == 0: s = 0, the
// "down" neighbour is a NaN pattern and the "up" residual is -0, both tests fail, s stays 0.)
JXLT_DI float sqrt_exact_midrange(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __int_as_float(__float_as_int(s) - 1);
  const float s_up = __int_as_float(__float_as_int(s) + 1);
  const float r_dn = nfma32(s_dn, s, x);
  const float r_up = nfma32(s_up, s, x);
  float r = (r_dn <= 0.0f) ? s_dn : s;
  r = (r_up > 0.0f) ? s_up : r;
  return r;
}

// Correctly rounded 1.0f / q for integer-valued q (0 < |q| <= 2^31): the hardware reciprocal
// (1 ulp) plus one residual correction.  On gfx950 this equals IEEE division for every such q
// (tools/rcp_probe.hip checks all 2^32 - 1 of them; tests/test_gpu_parity.py runs it), at 3
// instructions instead of the 11 of the generic division expansion.
JXLT_DI float rcp_int_exact(float q) {
  const float r0 = __builtin_amdgcn_rcpf(q);
  const float e0 = nfma32(q, r0, 1.0f);
  return fma32(e0, r0, r0);
}

// IEEE-correct num / den where operands and quotient are far from the overflow / underflow
// thresholds: the hardware reciprocal and the refinement steps of the generic expansion, without
// that expansion's operand scaling (v_div_scale x 2) and special-case fix-up (v_div_fixup) -- 8
// instructions instead of 11, the three dropped ones full-rate.  Every division of the kernels is
// of this kind for finite input of ordinary magnitude (denominators between 1e-3 and 1e6;
// DESIGN.md "domain of the guarantee").  tools/div_probe.hip compares it with the compiler's
// division on 2^32 operand pairs of magnitudes 2^-40 .. 2^40 (tests/test_gpu_parity.py runs it).
JXLT_DI float div_normal(float num, float den) {
  const float r0 = __builtin_amdgcn_rcpf(den);
  const float e0 = nfma32(den, r0, 1.0f);
  const float r1 = fma32(e0, r0, r0);
  const float q0 = num * r1;
  const float e1 = nfma32(den, q0, num);
  const float q1 = fma32(e1, r1, q0);
  const float e2 = nfma32(den, q1, num);
  return fma32(e2, r1, q1);
}

// fast_math-inl.h:113-133 + :74-108
JXLT_DI float fast_log2f(float x) {
  const float p0 = -1.8503833400518310E-06f, p1 = 1.4287160470083755E+00f,
              p2 = 7.4245873327820566E-01f;
  const float q0 = 9.9032814277590719E-01f, q1 = 1.0096718572241148E+00f,
              q2 = 1.7409343003366853E-01f;
  const int32_t x_bits = __float_as_int(x);
  const int32_t exp_bits = x_bits - 0x3f2aaaab;
  const int32_t exp_shifted = exp_bits >> 23;
  const float mantissa = __int_as_float(x_bits - (int32_t)((uint32_t)exp_shifted << 23));
  const float exp_val = (float)exp_shifted;
  const float t = mantissa - 1.0f;
  float yp = p2, yq = q2;
  yp = fma32(yp, t, p1);
  yq = fma32(yq, t, q1);
  yp = fma32(yp, t, p0);
  yq = fma32(yq, t, q0);
  return div_normal(yp, yq) + exp_val;
}

// fast_math-inl.h:137-151
JXLT_DI float fast_pow2f(float x) {
  const float floorx = floorf(x);
  const float e = __int_as_float((int32_t)((uint32_t)((int32_t)floorx + 127) << 23));
  const float frac = x - floorx;
  float num = frac + (float)1.01749063e+01;
  num = fma32(num, frac, (float)4.88687798e+01);
  num = fma32(num, frac, (float)9.85506591e+01);
  num = num * e;
  float den = fma32(frac, (float)2.10242958e-01, (float)-2.22328856e-02);
  den = fma32(den, frac, (float)-1.94414990e+01);
  den = fma32(den, frac, (float)9.85506633e+01);
  return div_normal(num, den);
}

// fast_math-inl.h:178-213
JXLT_DI float cube_root_and_add(float x, float add) {
  const float k1_3 = 1.0f / 3, k4_3 = 4.0f / 3;
  const float xa_3 = k1_3 * x;
  const int32_t m1 = __float_as_int(x);
  const int32_t m2 = (m1 == 0) ? 0 : (int32_t)(0x54800000u - (uint32_t)(m1 >> 23) * 0x002AAAAAu);
  float r = __int_as_float(m2);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float r2 = r * r;
    r = nfma32(xa_3, r2 * r2, k4_3 * r);
  }
  float r2 = r * r;
  r = fma32(k1_3, nfma32(x, r2 * r2, r), r);
  r2 = r * r;
  r = fma32(r2, x, add);
  return r;
}

// ZeroIfNegative (enc_xyb.cc:73-75) + CubeRootAndAdd in one: `mixed` is the biased mix BEFORE the clamp.
// The reference's result for an input clamped to zero is exactly `add` (seed 0 -> r stays 0 -> 0 * 0 + add), so
// the clamp, the zero test of the seed and its select collapse into ONE compare + select at the end; what
// the arithmetic in between produces for mixed <= 0 is never used.  The seed itself is a bit-field extract
// and a 24-bit multiply-add: e * -0x2AAAAA + 0x54800000 with the biased exponent e < 256 -- the same integer
// as 0x54800000 - (bits >> 23) * 0x2AAAAA for every positive input (denormals included: e = 0).
JXLT_DI float clamped_cube_root_and_add(float mixed, float add) {
#ifdef JXLT_CBRT_REFERENCE_SHAPE
  return cube_root_and_add(zero_if_negative(mixed), add);
#else
  const float k1_3 = 1.0f / 3, k4_3 = 4.0f / 3;
  const float x = mixed;
  const float xa_3 = k1_3 * x;
  const int32_t e = (int32_t)__builtin_amdgcn_ubfe(__float_as_uint(x), 23, 8);
  float r = __int_as_float(e * -0x002AAAAA + 0x54800000);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float r2 = r * r;
    r = nfma32(xa_3, r2 * r2, k4_3 * r);
  }
  float r2 = r * r;
  r = fma32(k1_3, nfma32(x, r2 * r2, r), r);
  r2 = r * r;
  r = fma32(r2, x, add);
  return mixed > 0.0f ? r : add;
#endif
}

// enc_xyb.cc:30-81
template <bool kNeedB = true>
JXLT_DI void linear_to_xyb(float r, float g, float b, float* ox, float* oy, float* ob) {
  const float kM02 = 0.078f, kM00 = 0.30f, kM01 = 1.0f - kM02 - kM00;
  const float kM12 = 0.078f, kM10 = 0.23f, kM11 = 1.0f - kM12 - kM10;
  const float kM20 = 0.24342268924547819f, kM21 = 0.20476744424496821f,
              kM22 = 1.0f - kM20 - kM21;
  const float bias = 0.0037930732552754493f;
  const float neg_bias_cbrt = -0.15595420054f;
  const float mixed0 = fma32(kM00, r, fma32(kM01, g, fma32(kM02, b, bias)));
  const float mixed1 = fma32(kM10, r, fma32(kM11, g, fma32(kM12, b, bias)));
  const float tm0 = clamped_cube_root_and_add(mixed0, neg_bias_cbrt);
  const float tm1 = clamped_cube_root_and_add(mixed1, neg_bias_cbrt);
  *ox = 0.5f * (tm0 - tm1);
  *oy = 0.5f * (tm0 + tm1);
  if (kNeedB) {  // (the halo columns only feed the adaptive quantisation, which reads X and Y)
    const float mixed2 = fma32(kM20, r, fma32(kM21, g, fma32(kM22, b, bias)));
    *ob = clamped_cube_root_and_add(mixed2, neg_bias_cbrt);
  }
}

// ---------------------------------------------------------------------------
// 1-D DCTs held in registers (enc_transforms-inl.h:292-425, dct_scales.h:82-107)
// ---------------------------------------------------------------------------

#define JXLT_SQRT2 1.41421356237f

JXLT_DI void dct4(float& m0, float& m1, float& m2, float& m3) {
  const float kW0 = (float)0.541196100146197, kW1 = (float)1.3065629648763764;
  const float t0 = m0 + m3, t1 = m1 + m2;
  const float u0 = t0 + t1, u1 = t0 - t1;
  const float t2 = (m0 - m3) * kW0, t3 = (m1 - m2) * kW1;
  float w0 = t2 + t3;
  const float w1 = t2 - t3;
  w0 = fma32(w0, JXLT_SQRT2, w1);
  m0 = u0;
  m1 = w0;
  m2 = u1;
  m3 = w1;
}

JXLT_DI void dct8(float* m) {
  const float kW[4] = {(float)0.5097955791041592, (float)0.6013448869350453,
                       (float)0.8999762231364156, (float)2.5629154477415055};
  float a0 = m[0] + m[7], a1 = m[1] + m[6], a2 = m[2] + m[5], a3 = m[3] + m[4];
  dct4(a0, a1, a2, a3);
  float b0 = (m[0] - m[7]) * kW[0], b1 = (m[1] - m[6]) * kW[1], b2 = (m[2] - m[5]) * kW[2],
        b3 = (m[3] - m[4]) * kW[3];
  dct4(b0, b1, b2, b3);
  b0 = fma32(b0, JXLT_SQRT2, b1);
  b1 = b1 + b2;
  b2 = b2 + b3;
  m[0] = a0; m[1] = b0; m[2] = a1; m[3] = b1;
  m[4] = a2; m[5] = b2; m[6] = a3; m[7] = b3;
}

JXLT_DI void dct16(float* m) {
  const float kW[8] = {(float)0.5024192861881557, (float)0.5224986149396889,
                       (float)0.5669440348163577, (float)0.6468217833599901,
                       (float)0.7881546234512502, (float)1.060677685990347,
                       (float)1.7224470982383342, (float)5.101148618689155};
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = m[i] + m[15 - i];
  dct8(a);
#pragma unroll
  for (int i = 0; i < 8; i++) b[i] = (m[i] - m[15 - i]) * kW[i];
  dct8(b);
  b[0] = fma32(b[0], JXLT_SQRT2, b[1]);
#pragma unroll
  for (int i = 1; i < 7; i++) b[i] = b[i] + b[i + 1];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    m[2 * i] = a[i];
    m[2 * i + 1] = b[i];
  }
}

// One butterfly exchange of an octet transpose: lanes with (l & S) == 0 keep `a` and receive
// their partner's `a` into `b`; the other lanes keep `b` and receive their partner's `b` into
// `a` (partner = lane l ^ S).
template <int S>
JXLT_DI void octet_exchange(float& a, float& b, int l) {
  const int ai = __float_as_int(a), bi = __float_as_int(b);
  if (S == 4) {
    // bank-masked row shifts do the select and the move in one instruction each
    a = __int_as_float(__builtin_amdgcn_update_dpp(ai, bi, kDppRowShr4, 0xF, 0xA, false));
    b = __int_as_float(__builtin_amdgcn_update_dpp(bi, ai, kDppRowShl4, 0xF, 0x5, false));
  } else {
    const bool hi = (l & S) != 0;
    const int pa = octet_xor_i<S>(ai), pb = octet_xor_i<S>(bi);
    a = hi ? __int_as_float(pb) : a;
    b = hi ? b : __int_as_float(pa);
  }
}

// 8x8 transpose across the 8 lanes of an octet: lane l holds v[j] = M[j][l] and
// ends with v[j] = M[l][j].
// JXLT_LDS_TRANSPOSE = 1: through a private LDS scratch of the octet (kTransposePitch floats,
// 8 more than the 64 it holds so that the eight octets of a wave fall into different banks):
// eight dword writes (immediate offsets j * 32 bytes), two 16-byte reads of the lane's row.  The
// wave's LDS operations execute in order (tools/lds_order_probe.hip checks exactly this on the
// GPU), so no barrier is needed between them; what this buys is
// VALU issue slots -- the register variant below costs 40 "full-rate" instructions (24 DPP moves +
// 16 selects, ~190 cycles per wave and transpose, tools/op_probe.hip), this one 10 LDS
// instructions that other waves' VALU work overlaps.
// JXLT_LDS_TRANSPOSE = 0: three butterfly stages in registers, static register indices.
// (Measured alternatives, both slower on gfx950: one assembly block of 24 fused
// v_cndmask_b32_dpp -- a VOP2 select whose mask does not come from a VALU compare is very slow,
// tools/op_probe.hip.)
#ifndef JXLT_LDS_TRANSPOSE
#define JXLT_LDS_TRANSPOSE 1
#endif
constexpr int kTransposePitch = 72;
// No instruction: the wave's LDS operations execute in order.  What has to be stopped is the
// compiler -- the stores and the loads of a transpose go through different types (float / float4),
// which type-based alias analysis treats as independent -- hence the memory clobber.
// (An execution model in which lanes are not lock-stepped defines its own JXLT_OCTET_SYNC before
// including this header: tests/hipsim does.)
#ifndef JXLT_OCTET_SYNC
#define JXLT_OCTET_SYNC()                  \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_wave_barrier();       \
    asm volatile("" ::: "memory");         \
  } while (0)
#endif
// The same for a whole wave (every lane of the wave reaches it).
#ifndef JXLT_WAVE_SYNC
#define JXLT_WAVE_SYNC() JXLT_OCTET_SYNC()
#endif
JXLT_DI void octet_transpose(float* v, float* sc, int l) {
#if JXLT_LDS_TRANSPOSE
  // Element (row r, column c) lives at (c >> 2) * 36 + r * 4 + (c & 3): the two 16-byte halves of
  // the rows form two dense 128-byte runs (the reads of the eight lanes are consecutive 16-byte
  // chunks), and the 4-dword gap between the runs puts the eight dwords a store instruction
  // writes per octet (column l of row j) into eight consecutive banks.
  float* const w = sc + (l >> 2) * 36 + (l & 3);
#pragma unroll
  for (int j = 0; j < 8; j++) w[j * 4] = v[j];
  JXLT_OCTET_SYNC();
  const float4 a = *reinterpret_cast<const float4*>(sc + l * 4);
  const float4 b = *reinterpret_cast<const float4*>(sc + 36 + l * 4);
  JXLT_OCTET_SYNC();  // (the next transpose overwrites the scratch)
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#else
  (void)sc;
  octet_exchange<4>(v[0], v[4], l);
  octet_exchange<4>(v[1], v[5], l);
  octet_exchange<4>(v[2], v[6], l);
  octet_exchange<4>(v[3], v[7], l);
  octet_exchange<2>(v[0], v[2], l);
  octet_exchange<2>(v[1], v[3], l);
  octet_exchange<2>(v[4], v[6], l);
  octet_exchange<2>(v[5], v[7], l);
  octet_exchange<1>(v[0], v[1], l);
  octet_exchange<1>(v[2], v[3], l);
  octet_exchange<1>(v[4], v[5], l);
  octet_exchange<1>(v[6], v[7], l);
#endif
}

// Block transforms.  `px` points at the block's top-left sample in an LDS plane
// of row pitch `pitch`; l = lane within the octet.  Results are the lane's
// "rows of 8": coefficient index i = r*8 + l (the reference's SIMD layout).

// The reference scales by 1/N after each 1-D pass (StoreToBlockAndScale, :387-390).  Those
// factors are powers of two, and scaling by a power of two commutes exactly with every
// rounded add/mul/fma of the second pass (no over/underflow at these magnitudes: pixel
// differences are 0 or >= 1 ulp of O(0.1) values), so both are applied once at the end.

// ComputeScaledDCT<8,8> (enc_transforms-inl.h:527-546): i = h*8 + v
JXLT_DI void block_dct8x8(const float* px, int pitch, int l, float* sc, float* c) {
#pragma unroll
  for (int y = 0; y < 8; y++) c[y] = px[y * pitch + l];
  dct8(c);
  octet_transpose(c, sc, l);  // lane v now holds 8*A[v][x], x = 0..7
  dct8(c);
#pragma unroll
  for (int y = 0; y < 8; y++) c[y] = (1.0f / 64) * c[y];  // c[h] = C[h][v=l]
}

// ComputeScaledDCT<16,8>: 16 rows x 8 cols, i = h*16 + v; r = 2h + (v>=8), lane = v&7
JXLT_DI void block_dct16x8(const float* px, int pitch, int l, float* sc, float* c) {
  float col[16];
#pragma unroll
  for (int y = 0; y < 16; y++) col[y] = px[y * pitch + l];
  dct16(col);
  float lo[8], hi[8];
#pragma unroll
  for (int v = 0; v < 8; v++) {
    lo[v] = col[v];
    hi[v] = col[v + 8];
  }
  octet_transpose(lo, sc, l);  // lane t: A[t][x]
  octet_transpose(hi, sc, l);  // lane t: A[t+8][x]
  dct8(lo);
  dct8(hi);
#pragma unroll
  for (int h = 0; h < 8; h++) {
    c[2 * h] = (1.0f / 128) * lo[h];
    c[2 * h + 1] = (1.0f / 128) * hi[h];
  }
}

// ComputeScaledDCT<8,16>: 8 rows x 16 cols, i = v*16 + h; r = 2v + (h>=8), lane = h&7
JXLT_DI void block_dct8x16(const float* px, int pitch, int l, float* sc, float* c) {
  float lo[8], hi[8];
#pragma unroll
  for (int y = 0; y < 8; y++) {
    lo[y] = px[y * pitch + l];
    hi[y] = px[y * pitch + l + 8];
  }
  dct8(lo);
  dct8(hi);
  octet_transpose(lo, sc, l);  // lane v: A[v][x], x < 8
  octet_transpose(hi, sc, l);  // lane v: A[v][x], x >= 8
  float row[16];
#pragma unroll
  for (int x = 0; x < 8; x++) {
    row[x] = lo[x];
    row[x + 8] = hi[x];
  }
  dct16(row);
#pragma unroll
  for (int h = 0; h < 8; h++) {
    lo[h] = (1.0f / 128) * row[h];
    hi[h] = (1.0f / 128) * row[h + 8];
  }
  octet_transpose(lo, sc, l);  // lane t: C[v][h=t], v = 0..7
  octet_transpose(hi, sc, l);  // lane t: C[v][h=t+8]
#pragma unroll
  for (int v = 0; v < 8; v++) {
    c[2 * v] = lo[v];
    c[2 * v + 1] = hi[v];
  }
}

// ---------------------------------------------------------------------------
// Adaptive quantisation helpers (enc_adaptive_quantization.cc)
// ---------------------------------------------------------------------------

// :78-104
JXLT_DI float ratio_of_derivatives(float v, bool invert) {
  const float kSGmul = 226.0480446705883f;
  const float kSGmul2 = 1.0f / 73.377132366608819f;
  const float kLog2 = 0.693147181f;
  const float kSGRetMul = kSGmul2 * 18.6580932135f * kLog2;
  const float kSGVOffset = 7.14672470003f;
  const float kEpsilon = (float)1e-2;
  v = zero_if_negative(v);
  const float kNumMul = kSGRetMul * 3 * kSGmul;
  const float kVOffset = kSGVOffset * kLog2 + kEpsilon;
  const float kDenMul = kLog2 * kSGmul;
  const float v2 = v * v;
  const float num = fma32(kNumMul, v2, kEpsilon);
  const float den = fma32(kDenMul * v, v2, kVOffset);
  return invert ? div_normal(num, den) : div_normal(den, num);
}

// :287-294.  sqrt(float(kMul * 1e8)) is a constant of the model; it is passed in
// so that it is computed once (correctly rounded) per thread.
JXLT_DI float masking_sqrt(float v, float sqrt_mul) {
  const float kLogOffset = 26.481471032459346f;
  return 0.25f * sqrt_exact_midrange(fma32(v, sqrt_mul, kLogOffset));  // argument >= kLogOffset
}
JXLT_DI float masking_sqrt_mul() {
  const float kMul = 211.50759899638012f;
  const float mul_v = (float)(kMul * 1e8);
  return sqrtf(mul_v);
}

// :52-75
JXLT_DI float compute_mask(float out_val) {
  const float kBase = -0.74174993f, kMul4 = 3.2353257320940401f, kMul2 = 12.906028311180409f,
              kOffset2 = 305.04035728311436f, kMul3 = 5.0220313103171232f,
              kOffset3 = 2.1925739705298404f, kMul0 = 0.74760422233706747f;
  const float kOffset4 = 0.25f * kOffset3;
  const float v1 = fmaxf(out_val * kMul0, 1e-3f);
  const float v2 = div_normal(1.0f, v1 + kOffset2);
  const float v3 = div_normal(1.0f, fma32(v1, v1, kOffset3));
  const float v4 = div_normal(1.0f, fma32(v1, v1, kOffset4));
  return kBase + fma32(kMul4, v4, fma32(kMul2, v2, kMul3 * v3));
}

// :296-320
// Keeps the four smallest of {min0<=min1<=min2<=min3, v}, sorted.  Same result as the
// reference's branchy insertion for non-NaN inputs (equal values are interchangeable).
JXLT_DI void store_min4(float v, float& min0, float& min1, float& min2, float& min3) {
  float t = fmaxf(min0, v);
  min0 = fminf(min0, v);
  float u = fmaxf(min1, t);
  min1 = fminf(min1, t);
  t = fmaxf(min2, u);
  min2 = fminf(min2, u);
  min3 = fminf(min3, t);
}

// ---------------------------------------------------------------------------
// Tile kernel
// ---------------------------------------------------------------------------

constexpr int kTileThreads = 512;
constexpr int kHalo = 5;                 // AQ: +-4 px window, +-1 px Laplacian tap
constexpr int kXYPitch = 64 + 2 * kHalo + 1;  // 75 floats (odd: conflict-free columns)
constexpr int kBPitch = 65;
constexpr int kPrePitch = 19;
constexpr int kCflTermFloats = 64 * 64 * 4;  // LDS floats overlaid by the CfL terms
// float stride between the blocks of the coefficient staging area (3 x 64 values each): 200 = 8 (mod 64), so
// the eight 32-byte runs the octets of a wave store at a time land in different banks
constexpr int kStageStrideF = 200;

struct alignas(16) TileShared {
  float x[64 * kXYPitch];
  float y[64 * kXYPitch];
  float b[64 * kBPitch];
  float rowsum[16 * 72];   // AQ: per 4-row band, per column
  float pre_erosion[16 * kPrePitch];
  float erosion[16 * 16];
  float cfl_pad[kCflTermFloats - (2 * 64 * kXYPitch + 64 * kBPitch + 16 * 72 + 16 * kPrePitch + 16 * 16)];
  // ^ x..cfl_pad (64 KB) are overlaid by the chroma-from-luma terms once every pixel
  //   read is done: 64 blocks x 64 coefficients x (a_x, b_x, a_b, b_b).
  // rowsum..transpose_pad: during the transforms (the AQ buffers are dead by then) the octets'
  // transpose scratch, 64 x kTransposePitch floats.
  float transpose_pad[64 * 72 - (kCflTermFloats - (2 * 64 * kXYPitch + 64 * kBPitch))];
  // (its first 128 floats hold the candidate entropies of the 2x2 cells during the strategy search, "ent8")
  float sqrt_lut[kSqrtLutSize];  // sqrtf of the quantised magnitudes below kSqrtLutSize
  float inv_w[576];
  float aq[64];            // quant field (tile-local 8x8)
  float mask[64];
  float cfl_sum[4];        // ca_x, cb_x, ca_b, cb_b
  int cmap[2];             // ytox, ytob
  uint8_t raw_quant[64];
  uint8_t strat[64];
  uint32_t ntok;
  uint32_t nfirst;
};
// After the last pixel read the XYB planes are dead and are reused: chroma-from-luma terms, the parked
// DCT8 coefficients of the entropy estimate, then the staging area of the selected transforms' coefficients
// (64 blocks x 3 channels x 64 floats).

JXLT_DI int imin(int a, int b) { return a < b ? a : b; }
JXLT_DI int imax(int a, int b) { return a > b ? a : b; }

// Per-octet entropy estimate of one transform (enc_ac_strategy.cc:51-146).
// cy/cx/cb: the lane's rows of the Y/X/B coefficients; NR rows (8 or 16).
// kLut: the roots come from the LDS table S.sqrt_lut (a multiply, a convert, a mask and an LDS
// read instead of v_sqrt + the exact-rounding fix-up, ~40 cycles); *qmax then receives the
// largest magnitude seen, and the caller redoes the estimate with kLut = false if it is beyond
// the table (quantised coefficients >= 1024: practically never, but results must not depend on it).
template <int NR, bool kLut>
JXLT_DI float estimate_entropy(const float* cx, const float* cy, const float* cb, const float* inv_x,
                               const float* inv_y, const float* inv_b, int l, float quant,
                               float masking, float cmap_x, float cmap_b, float distance,
                               const float* sqrt_lut, float* qmax) {
  const float num_blocks = (float)(NR / 8);
  const float kInfoLossMultiplier = 138.0f;
  const float kInfoLossMultiplier2 = (float)50.46839691767866;
  const float kCost2 = 4.4628149885273363f;
  const float kCostDelta = 5.3359184934516337f;
  const float kZerosMul = 7.565053364251793f;
  const float slope = fminf(1.0f, distance * (1.0f / 3));
  const float cost_of_1 = 1 + slope * 8.8703248061477744f;
  float entropy = 0.0f;
  float info_loss = 0.0f, info_loss2 = 0.0f;
  uint32_t qbits = 0;  // OR of the offset words before masking (a v_or is cheaper than a v_max)
  // One copy of the body per channel (no per-coefficient operand selects); the scheduling
  // fences keep the channels from being interleaved, which would spill.
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float* inv = c == 0 ? inv_x : c == 1 ? inv_y : inv_b;
    const float* cin = c == 0 ? cx : c == 1 ? cy : cb;
    const float cmap_factor = c == 0 ? cmap_x : c == 1 ? 0.0f : cmap_b;
    float entropy_v = 0.0f, nzeros_v = 0.0f;
    JXLT_SCHED_FENCE();
    // Selects and compares are the expensive kind of VALU instruction on gfx950
    // (tools/op_probe.hip), multiply-adds with the (free) clamp modifier are not.  With q a
    // non-negative integer:  [q >= 2] = clamp01(q - 1),  [q >= 1] = clamp01(4 * q),  and
    // x + (c ? k : 0) == fma(c, k, x) for c in {0, 1}.
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const float in = cin[r];
      const float im = inv[r * 8 + l];
      // (skipping the subtraction of cy * 0 for the Y channel saves two instructions per
      // coefficient on paper; the register allocator then spills 90 VGPRs)
      const float val = (in - cy[r] * cmap_factor) * (im * quant);
      const float rval = rintf(val);
      const float diff = fabsf(val - rval);
      info_loss = info_loss + diff;
      info_loss2 = fma32(diff, diff, info_loss2);
      const float q = fabsf(rval);
      entropy_v = fma32(clamp01(q - 1.0f), kCost2, entropy_v);  // + (q >= 1.5 ? kCost2 : 0)
      float root;
      if (kLut) {
        // byte offset 4 * q, wrapped into the table (a wrapped read is redone by the caller)
        // 4 * q + 2^23 is exact for q < 2^21 and its bit pattern is 0x4B000000 + 4 * q: the byte
        // offset comes out of a multiply-add and a mask, no float -> int conversion (a full-rate
        // instruction, tools/op_probe.hip).  Larger q (or NaN) disturb the bits above the offset
        // field, which the OR below keeps for the caller's overflow test.
        const uint32_t off_raw = __float_as_uint(fma32(q, 4.0f, 8388608.0f));
        const uint32_t off = off_raw & (uint32_t)(kSqrtLutSize * 4 - 4);
        root = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sqrt_lut) + off);
        qbits |= off_raw;
      } else {
        // (skipping the root with a branch where a whole wave has q <= 1 was tried: control flow
        // inside this loop makes the register allocator spill)
        root = sqrt_exact_midrange(q);  // q is 0 or an integer >= 1
      }
      entropy_v = fma32(root, kCostDelta, entropy_v);
      nzeros_v = nzeros_v + clamp01(4.0f * q);  // + (q == 0 ? 0 : 1)
    }
    entropy_v = fma32(nzeros_v, cost_of_1, entropy_v);
    entropy += octet_sum(entropy_v);
    const uint32_t num_nzeros = (uint32_t)octet_sum(nzeros_v);
    const uint32_t nbits = (uint32_t)ceil_log2_nonzero(num_nzeros + 1) + 1;
    entropy += kZerosMul * (float)(ceil_log2_nonzero(nbits + 17) + nbits);
  }
  const float infoloss = octet_sum(info_loss);
  const float infoloss2 = sqrtf(num_blocks * octet_sum(info_loss2));
  const float info_loss_score = (kInfoLossMultiplier * infoloss + kInfoLossMultiplier2 * infoloss2);
  // every offset stayed inside the table <=> nothing above the offset field differs from 2^23's pattern
  if (kLut) *qmax = ((qbits | 0x4B000000u) & ~(uint32_t)(kSqrtLutSize * 4 - 1)) != 0x4B000000u ? (float)kSqrtLutSize : 0.0f;
  return entropy + masking * info_loss_score;
}

// enc_group.cc:186-218 for channel 1.  `quant` is the quantised coefficient as a float: 0 (+0, the
// thresholded case) or an integer of magnitude >= 1.  |quant| <= 1: the reference selects
// +-kBias1 by sign, 0 for 0 -- which is quant * kBias1 exactly; otherwise quant - kBias3 / quant.
JXLT_DI float adjust_quant_bias_y(float quant) {
  const float kBias1 = 1.0f - 0.07005449891748593f;  // kDefaultQuantBias[1]
  const float kBias3 = 0.145f;
  const float small = quant * kBias1;
  const float bias = nfma32(kBias3, rcp_int_exact(quant), quant);  // (quant == 0: selected away below)
  return fabsf(quant) < 1.125f ? small : bias;
}

// kDebug: the variant that serves the A.dbg_* outputs (per-phase clocks, intermediate planes for
// the parity tests); the production variant has none of their tests, branches and registers.
template <bool kLutRoots, bool kDebug>
JXLT_DI void tile_kernel_body(const TileArgs& A) {
  __shared__ TileShared S;
  const int tid = (int)threadIdx.x;
  const int l = tid & 7;    // lane within octet
  const int oct = tid >> 3;  // octet index == block index within tile (0..63)
  const DeviceTables* T = A.tab;
  long long t_prev = (kDebug && A.dbg_phase) ? clock64() : 0;
  // Profiling builds (-DJXLT_PHASE_STOPS, tools/phase_pmc.py) can truncate the kernel after
  // phase i; the early exits perturb code generation, so production builds leave them out.
#ifdef JXLT_PHASE_STOPS
#define JXLT_STOP(i) if (((A.flags >> 8) & 15u) == (unsigned)(i) + 1u) return;
#else
#define JXLT_STOP(i)
#endif
#define JXLT_MARK(i)                                                        \
  if (kDebug && A.dbg_phase && tid == 0) {                                  \
    const long long t_now = clock64();                                      \
    atomicAdd(&A.dbg_phase[i], (unsigned long long)(t_now - t_prev));       \
    t_prev = t_now;                                                         \
  }                                                                         \
  JXLT_STOP(i)

  // ---- geometry (enc_frame.cc:716-751) ------------------------------------
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed placement, used for
  // speed only), and each XCD has its own L2.  Give every XCD one contiguous raster range
  // of tiles so that horizontally adjacent tiles -- which share the +-5 px halo columns and
  // the partially covered 128-byte lines -- are served by the same L2.
  int tile_id;
  {
    const int n = A.g.xsize_tiles * A.g.ysize_tiles;
    const int b = (int)blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int q = n >> 3, r = n & 7;
    tile_id = xcd * q + (xcd < r ? xcd : r) + idx;
  }
  const int tx_img = tile_id % A.g.xsize_tiles, ty_img = tile_id / A.g.xsize_tiles;
  const int gx = tx_img >> 2;
  const int sx0 = gx * 256, sy0 = ty_img * 64;            // stripe origin (pixels)
  const int sw = imin(256, A.g.xsize - sx0), sh = imin(64, A.g.ysize - sy0);
  const int swp = (sw + 7) & ~7, shp = (sh + 7) & ~7;       // padded stripe size
  const int tbx0 = (tx_img & 3) * 8;                        // tile origin in stripe blocks
  const int nbx = imin(8, swp / 8 - tbx0), nby = shp / 8;   // tile size in blocks
  const int px0 = tbx0 * 8;                                 // tile origin in stripe pixels
  const int bx_img0 = gx * 32 + tbx0, by_img0 = ty_img * 8; // image-absolute block origin
  const int obx = oct & 7, oby = oct >> 3;                  // octet's block in the tile
  const bool blk_valid = obx < nbx && oby < nby;
  const uint32_t bstride = (uint32_t)A.g.xsize_blocks;

  // ---- P0: tables -> LDS; load + XYB (enc_frame.cc:597-617, enc_xyb.cc) -----
  // The table values are REQUESTED here (every lane, clamped indices: no branches) and stored to LDS behind the
  // pixel requests below, so that all of the tile's global loads are in flight together.  (Loops of "load, wait,
  // store to LDS" in front of the pixel loads cost six serial round trips to L2 per tile.)
  static_assert(kTileThreads == 512 && (kSqrtLutSize <= 512 || kSqrtLutSize == 1024), "table staging below");
  const float tab_inv0 = T->inv_weights[tid];
  const float tab_inv1 = T->inv_weights[512 + (tid & 63)];
  const float tab_root0 = T->sqrt_lut[tid & (kSqrtLutSize - 1)];
  const float tab_root1 = T->sqrt_lut[(512 + tid) & (kSqrtLutSize - 1)];
  if (tid == 0) {
    S.ntok = 0;
    S.nfirst = 0;
  }
  {
    // 16 lanes along x, 32 rows per pass: a thread owns 5 columns x 2 rows of the
    // (64 + 2*kHalo)-wide window, so the row and column clamps are shared and all thirty loads
    // are in flight before the first use.  Column slots 0-3 cover the 64 interior columns; slot 4
    // takes the ten halo columns (lanes 0-4 left, 5-9 right), which need X and Y only.
    constexpr int kWin = 64 + 2 * kHalo;
    const int base = px0 - kHalo;  // stripe x of LDS column 0
    const int lx = tid & 15, ly = tid >> 4;
    const float* rowp[2][3];
    bool yok[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int y = ly + 32 * h;
      yok[h] = y < shp;
      const ptrdiff_t off = (ptrdiff_t)(sy0 + imin(y, sh - 1)) * A.pitch + (ptrdiff_t)sx0 * A.pix_stride;
      rowp[h][0] = A.planes[0] + off;
      rowp[h][1] = A.planes[1] + off;
      rowp[h][2] = A.planes[2] + off;
    }
    // (scalar arithmetic on purpose: on gfx950 a packed v_pk_*_f32 costs at least as much as its
    // two scalar halves -- tools/pk_probe.hip -- and the packed variant of this loop measured
    // 3 % slower for the whole kernel)
    float pr[5][2], pg[5][2], pb[5][2];
    bool xok[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int cx = j < 4 ? kHalo + lx + 16 * j : (lx < kHalo ? lx : lx < 2 * kHalo ? 64 + lx : kWin);
      const int x = base + cx;
      xok[j] = cx < kWin && x >= 0 && x < swp && x < px0 + nbx * 8 + kHalo;
      const int xs = (xok[j] ? imin(x, sw - 1) : 0) * A.pix_stride;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        pr[j][h] = rowp[h][0][xs];
        pg[j][h] = rowp[h][1][xs];
        pb[j][h] = rowp[h][2][xs];
      }
    }
    // (table values -> LDS while the pixels are on their way; they were requested first, so the wait is theirs only)
    S.inv_w[tid] = tab_inv0;
    if (tid < 64) S.inv_w[512 + tid] = tab_inv1;
    if (tid < kSqrtLutSize) S.sqrt_lut[tid] = tab_root0;
    if (kSqrtLutSize > 512) S.sqrt_lut[(512 + tid) & (kSqrtLutSize - 1)] = tab_root1;
    if (A.byteswap) {  // big-endian PFM payload (BSwapFloat, read_pfm.cc:206)
#pragma unroll
      for (int j = 0; j < 5; j++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          pr[j][h] = __uint_as_float(__builtin_bswap32(__float_as_uint(pr[j][h])));
          pg[j][h] = __uint_as_float(__builtin_bswap32(__float_as_uint(pg[j][h])));
          pb[j][h] = __uint_as_float(__builtin_bswap32(__float_as_uint(pb[j][h])));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
      if (!xok[j]) continue;
      const int cx = j < 4 ? kHalo + lx + 16 * j : (lx < kHalo ? lx : 64 + lx);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if (!yok[h]) continue;
        const int y = ly + 32 * h;
        float px_, py_, pb_ = 0.0f;
        if (j < 4) linear_to_xyb<true>(pr[j][h], pg[j][h], pb[j][h], &px_, &py_, &pb_);
        else linear_to_xyb<false>(pr[j][h], pg[j][h], pb[j][h], &px_, &py_, &pb_);
        S.x[y * kXYPitch + cx] = px_;
        S.y[y * kXYPitch + cx] = py_;
        if (j < 4) S.b[y * kBPitch + cx - kHalo] = pb_;
        if (kDebug && j < 4 && A.dbg_xyb[0] && cx < kHalo + nbx * 8) {
          const size_t d = (size_t)(by_img0 * 8 + y) * ((size_t)bstride * 8) + (size_t)(bx_img0 * 8 + cx - kHalo);
          A.dbg_xyb[0][d] = px_;
          A.dbg_xyb[1][d] = py_;
          A.dbg_xyb[2][d] = pb_;
        }
      }
    }
  }
  __syncthreads();
  JXLT_MARK(0);
  // LDS column of stripe pixel x is (x - px0 + kHalo).
#define SX(yy, xx) S.x[(yy) * kXYPitch + ((xx) - px0 + kHalo)]
#define SY(yy, xx) S.y[(yy) * kXYPitch + ((xx) - px0 + kHalo)]

  // ---- P1: AQ per-pixel masked Laplacian energy, summed over 4-row bands ----
  // (enc_adaptive_quantization.cc:376-483)
  int aq_x0 = px0, aq_x1 = px0 + nbx * 8;
  if (aq_x0 != 0) aq_x0 -= 4;
  if (aq_x1 != swp) aq_x1 += 4;
  const int aq_w = aq_x1 - aq_x0;  // <= 72
  {
    const float match_gamma_offset = (float)0.019;
    const float kXMul = 23.426802998210313f;
    const float sqrt_mul = masking_sqrt_mul();
    // Positions handled by the reference's 8-lane vector loop: [vs, ve).
    const int vs = aq_x0 == 0 ? 1 : aq_x0;
    const int nvec = (aq_x1 - 10 >= vs) ? ((aq_x1 - 10 - vs) / 8 + 1) : 0;
    const int ve = vs + 8 * nvec;
    const int nbands = nby * 2;
    // One pixel's term from its own value, the sum of its vertical neighbours and its horizontal
    // neighbours; both association orders are computed and selected (a branch per pixel would wait
    // for the LDS before and after each arm).
    auto pixel_term = [&](bool vec, float in, float du, float in_l, float in_r, float ix, float dux, float ix_l,
                          float ix_r) {
      const float base = 0.25f * (vec ? (in_r + in_l) + du : (du + in_l) + in_r);
      const float gammac = ratio_of_derivatives(in + match_gamma_offset, false);
      float diff = gammac * (in - base);
      diff = diff * diff;
      const float base_x = 0.25f * (vec ? (ix_r + ix_l) + dux : (dux + ix_l) + ix_r);
      float diff_x = gammac * (ix - base_x);
      diff_x = diff_x * diff_x;
      const float fused = fma32(kXMul, diff_x, diff), unfused = diff + kXMul * diff_x;
      return masking_sqrt(vec ? fused : unfused, sqrt_mul);
    };
    // (Spreading the last, partly filled pass over all threads row by row changes nothing: the
    // other resident workgroup takes the issue slots the idle waves leave.)
    for (int i = tid; i < nbands * aq_w; i += kTileThreads) {
      const int q = i / aq_w, x = aq_x0 + i % aq_w;
      const bool vec = x >= vs && x < ve;
      const int xl = x > 0 ? x - 1 : x, xr = x + 1 < swp ? x + 1 : x;
      // The band's column, rows y0-1 .. y0+4 (clamped to the stripe: only the first and the
      // last entry can clamp, shp = 8 nby), is read once.
      const int y0 = q * 4;
      const int yu0 = y0 > 0 ? y0 - 1 : y0, yd3 = y0 + 4 < shp ? y0 + 4 : y0 + 3;
      float cy[6], cxx[6], ly4[4], ry4[4], lx4[4], rx4[4];
      cy[0] = SY(yu0, x);
      cxx[0] = SX(yu0, x);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        cy[k + 1] = SY(y0 + k, x);
        cxx[k + 1] = SX(y0 + k, x);
        ly4[k] = SY(y0 + k, xl);
        ry4[k] = SY(y0 + k, xr);
        lx4[k] = SX(y0 + k, xl);
        rx4[k] = SX(y0 + k, xr);
      }
      cy[5] = SY(yd3, x);
      cxx[5] = SX(yd3, x);
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float diff = pixel_term(vec, cy[k + 1], cy[k + 2] + cy[k], ly4[k], ry4[k], cxx[k + 1],
                                      cxx[k + 2] + cxx[k], lx4[k], rx4[k]);
        acc = (k == 0) ? diff : acc + diff;
      }
      // P2, the 4-column average (:484-491), inside the quad: aq_w is a multiple of 4 and so
      // is the stride, so the four columns of one average sit in one aligned quad of lanes;
      // summed in the reference's order ((c0 + c1) + c2) + c3 by lane 0 of the quad.
      float s4 = acc + quad_lane<1>(acc);
      s4 = s4 + quad_lane<2>(acc);
      s4 = s4 + quad_lane<3>(acc);
      if ((i & 3) == 0) S.pre_erosion[q * kPrePitch + ((x - aq_x0) >> 2)] = s4 * 0.25f;
    }
  }
  __syncthreads();
  JXLT_MARK(1);
  const int pre_xs = aq_w / 4, pre_ys = nby * 2;
  // ---- P3: fuzzy erosion (:322-374) ------------------------------------------
  {
    const int rx0 = (aq_x0 % 8 == 0) ? 0 : 1;
    // The four cells of one block share a quad of lanes (cell c = lane & 3, row-major); quad e of
    // the first 256 threads has block e of the 8x8 grid, blocks outside the tile get aq = mask = 0.
    if (tid < 256) {
      const int i = tid, ebx = (tid >> 2) & 7, eby = tid >> 5;
      const bool eb_valid = ebx < nbx && eby < nby;
      const int fy = eb_valid ? 2 * eby + ((i >> 1) & 1) : 0, fx = eb_valid ? 2 * ebx + (i & 1) : 0;
      const int y = fy, x = fx + rx0;
      const int ym1 = y >= 1 ? y - 1 : y, yp1 = y + 1 < pre_ys ? y + 1 : y;
      const int xm1 = x >= 1 ? x - 1 : x, xp1 = x + 1 < pre_xs ? x + 1 : x;
      const float* rowt = &S.pre_erosion[ym1 * kPrePitch];
      const float* row = &S.pre_erosion[y * kPrePitch];
      const float* rowb = &S.pre_erosion[yp1 * kPrePitch];
      float min0 = row[x], min1 = row[xm1], min2 = row[xp1], min3 = rowt[xm1], t;
#define JXLT_SWAP_GT(a, b) { t = fminf(a, b); b = fmaxf(a, b); a = t; }
      JXLT_SWAP_GT(min0, min1);
      JXLT_SWAP_GT(min0, min2);
      JXLT_SWAP_GT(min0, min3);
      JXLT_SWAP_GT(min1, min2);
      JXLT_SWAP_GT(min1, min3);
      JXLT_SWAP_GT(min2, min3);
#undef JXLT_SWAP_GT
      store_min4(rowt[x], min0, min1, min2, min3);
      store_min4(rowt[xp1], min0, min1, min2, min3);
      store_min4(rowb[xm1], min0, min1, min2, min3);
      store_min4(rowb[x], min0, min1, min2, min3);
      store_min4(rowb[xp1], min0, min1, min2, min3);
      const float kMul = 0.05f;
      const float ev = kMul * row[x] + kMul * min0 + kMul * min1 + kMul * min2 + kMul * min3;
      // Block value (:366-373): ((e00 + e01) + e10) + e11, by lane 0 of the quad.
      float v = ev + quad_lane<1>(ev);
      v = v + quad_lane<2>(ev);
      v = v + quad_lane<3>(ev);
      if ((i & 3) == 0) {
        S.aq[eby * 8 + ebx] = eb_valid ? v : 0.0f;
        S.mask[eby * 8 + ebx] = eb_valid ? div_normal(1.0f, v + 0.001f) : 0.0f;  // ComputeMaskForAcStrategyUse (:46-50)
      }
    }
  }
  __syncthreads();
  JXLT_MARK(2);
  // ---- P4: per-block modulations, one octet per block (:114-285) -------------
  {
    float out_val = 0.0f;
    const int bxp = px0 + obx * 8, byp = oby * 8;  // block origin (stripe pixels)
    if (blk_valid) {
      out_val = compute_mask(S.aq[oct]);
    }
    // HfModulation (:209-247): lane l = column l of the block
    float hf = 0.0f, red = 0.0f, blue = 0.0f, gam = 0.0f;
    const float kBias = 0.16f;
    const float kRedRampStart = (float)0.0073200141118951231;
    const float kRedRampLength = (float)0.019421555948474039;
    const float kBlueRampLength = (float)0.086890611400405895;
    const float kBlueRampStart = (float)0.26973418507870539;
    const int right_step = l < 7 ? 1 : 0;
    if (blk_valid) {
#pragma unroll
      for (int dy = 0; dy < 8; dy++) {
        const int yy = byp + dy, xx = bxp + l;
        const float p = SY(yy, xx);
        // column 7 has no right neighbour inside the block: it reads itself (|p - p| = 0, as the
        // reference adds) instead of branching around the read
        hf = hf + fabsf(p - SY(yy, xx + right_step));
        const float pd = (dy == 7) ? p : SY(yy + 1, xx);
        hf = hf + fabsf(p - pd);
        // ColorModulation (:146-207)
        const float vx = SX(yy, xx);
        const float vb = S.b[yy * kBPitch + obx * 8 + l];
        const float pixel_x = fmaxf(0.0f, vx - kRedRampStart);
        const float pixel_b = fmaxf(0.0f, vb - (p + kBlueRampStart));
        red = red + fminf(pixel_x, kRedRampLength);
        blue = blue + fminf(pixel_b, kBlueRampLength);
        // GammaModulation (:114-144)
        const float iny = p + kBias;
        const float rr = iny - vx, gg = iny + vx;
        const float ratio_r = ratio_of_derivatives(rr, true);
        const float ratio_g = ratio_of_derivatives(gg, true);
        gam = gam + 0.5f * (ratio_r + ratio_g);
      }
    }
    hf = octet_sum(hf);
    red = octet_sum(red);
    blue = octet_sum(blue);
    gam = octet_sum(gam);
    if (blk_valid) {
      out_val = fma32(hf, -2.0052193233688884f / 112, out_val);
      {
        const float kStrengthMul = (float)2.177823400325309;
        const double butteraugli_target = (double)A.distance;
        const float strength = (float)(kStrengthMul * (1.0f - 0.25f * butteraugli_target));
        if (!(strength < 0)) {
          const float red_strength = strength * 5.992297772961519f;
          const float blue_strength = strength;
          const float offset = strength * -0.009174542291185913f;
          out_val = out_val + offset;
          const float ratio = 30.610615782142737f;
          float overall_red = fminf(red, ratio * kRedRampLength);
          overall_red = overall_red * (red_strength / ratio);
          float overall_blue = fminf(blue, ratio * kBlueRampLength);
          overall_blue = overall_blue * (blue_strength / ratio);
          out_val = overall_red + (overall_blue + out_val);
        }
      }
      {
        const float overall_ratio = gam * (1.0f / 64);
        const float kGam = -0.15526878023684174f * 0.693147180559945f;
        out_val = fma32(kGam, fast_log2f(overall_ratio), out_val);
      }
      // PerBlockModulations tail (:249-285) + raw quant (:518-534)
      const float kAcQuant = 0.8294f;
      const float scale = div_normal(kAcQuant, A.distance);
      const float base_level = 0.5f * scale;
      float dampen = 1.0f;
      if (A.distance >= 7.0f) {
        dampen = 1.0f - ((A.distance - 7.0f) / (14.0f - 7.0f));
        if (dampen < 0) dampen = 0;
      }
      const float mul = scale * dampen;
      const float add = (1.0f - dampen) * base_level;
      const float qf = fast_pow2f(out_val * 1.442695041f) * mul + add;
      if (l == 0) {
        S.aq[oct] = qf;
        int v = (int)(qf * A.inv_scale + 0.5f);
        v = v < 1 ? 1 : v > 255 ? 255 : v;
        S.raw_quant[oct] = (uint8_t)v;
        S.strat[oct] = 1;  // DCT8, first block (FillDCT8)
        if (kDebug && A.dbg_qf) {
          const uint32_t pos = (uint32_t)(by_img0 + oby) * bstride + (uint32_t)(bx_img0 + obx);
          A.dbg_qf[pos] = qf;
          A.dbg_mask[pos] = S.mask[oct];
        }
      }
    }
  }
  __syncthreads();
  JXLT_MARK(3);

  // ---- P6a: candidate two-block transforms (enc_ac_strategy.cc:62-66) -------
  // Waves 0-3 take the 32 DCT16X8 candidates, waves 4-7 the 32 DCT8X16 candidates.
  // Done before chroma-from-luma so that afterwards no pixel is needed any more; the
  // coefficients stay in registers for the entropy estimate and for P8.
  float* const tsc = &S.rowsum[0] + oct * kTransposePitch;  // octet's transpose scratch (AQ buffers are dead)
  float c16x[16], c16y[16], c16b[16];
  const bool search = (A.flags & 1u) == 0;
  const int cand = oct & 31;           // candidate index within its type
  const int cell = cand >> 1;          // 2x2 cell index (4x4 cells per tile)
  const int ccx = (cell & 3) * 2, ccy = (cell >> 2) * 2;  // cell origin (tile blocks)
  const bool is_tall = oct < 32;       // DCT16X8 (16 rows x 8 cols)
  const int cbx = is_tall ? ccx + (cand & 1) : ccx;       // candidate's first block
  const int cby = is_tall ? ccy : ccy + (cand & 1);
  const bool cell_valid = search && (ccx + 1 < nbx) && (ccy + 1 < nby);
  if (cell_valid) {
    const float* pxp = &S.x[(cby * 8) * kXYPitch + cbx * 8 + kHalo];
    const float* pyp = &S.y[(cby * 8) * kXYPitch + cbx * 8 + kHalo];
    const float* pbp = &S.b[(cby * 8) * kBPitch + cbx * 8];
    // (scheduling fences: interleaving the three independent transforms would triple the
    // live registers and spill)
    if (is_tall) {
      block_dct16x8(pxp, kXYPitch, l, tsc, c16x);
      JXLT_SCHED_FENCE();
      block_dct16x8(pyp, kXYPitch, l, tsc, c16y);
      JXLT_SCHED_FENCE();
      block_dct16x8(pbp, kBPitch, l, tsc, c16b);
    } else {
      block_dct8x16(pxp, kXYPitch, l, tsc, c16x);
      JXLT_SCHED_FENCE();
      block_dct8x16(pyp, kXYPitch, l, tsc, c16y);
      JXLT_SCHED_FENCE();
      block_dct8x16(pbp, kBPitch, l, tsc, c16b);
    }
    JXLT_SCHED_FENCE();
  }
  JXLT_MARK(4);
  // ---- P5: DCT8 of every block (kept in registers) + chroma-from-luma -------
  // (enc_chroma_from_luma.cc:40-131)
  float c8x[8], c8y[8], c8b[8];
  {
    const float* pxp = &S.x[(oby * 8) * kXYPitch + obx * 8 + kHalo];
    const float* pyp = &S.y[(oby * 8) * kXYPitch + obx * 8 + kHalo];
    const float* pbp = &S.b[(oby * 8) * kBPitch + obx * 8];
    if (blk_valid) {
      block_dct8x8(pxp, kXYPitch, l, tsc, c8x);
      JXLT_SCHED_FENCE();
      block_dct8x8(pyp, kXYPitch, l, tsc, c8y);
      JXLT_SCHED_FENCE();
      block_dct8x8(pbp, kBPitch, l, tsc, c8b);
      JXLT_SCHED_FENCE();
    } else {
      // (cross-lane traffic never leaves an octet, so idle octets may skip it)
#pragma unroll
      for (int r = 0; r < 8; r++) c8x[r] = c8y[r] = c8b[r] = 0.0f;
    }
  }
  __syncthreads();  // all pixel reads done: the planes are dead from here on
  JXLT_MARK(5);
  // ---- P5b: chroma-from-luma (enc_chroma_from_luma.cc:40-131) ----------------
  {
    // Every octet publishes the terms of its block, a = m/84 and b = base*m - s with
    // m = Y*qm, s = C*qm (:49-53,117-120); then four sequential per-lane fma chains
    // (ca = sum a*a, cb = sum a*b, for X and for B) run over the blocks in raster order.
    // Term layout: [block][lane l][chunk], a chunk = (a, b) of two consecutive rows of one
    // chroma channel (chunk = channel * 4 + row / 2), i.e. the eight rows a chain lane needs
    // from a block are four 16-byte reads.  The chunk slot is XORed with l so that the eight
    // lanes of an octet hit different banks.
    float* terms = &S.x[0];
    // (swizzle key: l for lanes 0-3, l ^ 1 for lanes 4-7 -- a 16-byte LDS load is serviced in 16-lane
    // groups that pair lanes 0-3 of the X chain with lanes 4-7 of the B chain, MI355X_MICROARCH.md;
    // with the plain key those read the same banks)
    const int lsw = l ^ (l >> 2);
    const float* qm_x = S.inv_w + 0;    // InvMatrix(DCT, 0)
    const float* qm_b = S.inv_w + 128;  // InvMatrix(DCT, 2)
    const int nblk = nbx * nby;
    const float kInvColorFactor = 1.0f / 84;
    if (blk_valid) {
      float* dst = &terms[(oby * nbx + obx) * 256 + l * 32];
#pragma unroll
      for (int r = 0; r < 8; r += 2) {
        float4 tx, tb;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int rr = r + h;
          const bool dc = (rr == 0 && l == 0);  // block_*[0] = 0 (:109-111)
          const float by_ = dc ? 0.0f : c8y[rr], bx_ = dc ? 0.0f : c8x[rr], bb_ = dc ? 0.0f : c8b[rr];
          const float qx = qm_x[rr * 8 + l], qb = qm_b[rr * 8 + l];
          const float m_x = by_ * qx, s_x = bx_ * qx, m_b = by_ * qb, s_b = bb_ * qb;
          const float ax = kInvColorFactor * m_x, bx2 = 0.0f * m_x - s_x;
          const float ab = kInvColorFactor * m_b, bb2 = 1.0f * m_b - s_b;
          if (h == 0) { tx.x = ax; tx.y = bx2; tb.x = ab; tb.y = bb2; }
          else { tx.z = ax; tx.w = bx2; tb.z = ab; tb.w = bb2; }
        }
        *(float4*)&dst[(((r >> 1)) ^ lsw) * 4] = tx;
        *(float4*)&dst[((4 + (r >> 1)) ^ lsw) * 4] = tb;
      }
    }
    __syncthreads();
    // Chain lanes: wave 0 lanes 0-15 run ca (X: 0-7, B: 8-15), wave 1 lanes 0-15 run cb.
    // The 512 fused multiply-adds of a chain are strictly sequential, so these two waves are
    // the critical path of the workgroup: they run at raised issue priority, and the reads
    // of block blk + 1 are issued before the arithmetic of block blk.
    float acc = 0.0f;
    const int cw = tid >> 6, cl = tid & 63;
#ifndef JXLT_CFL_PINGPONG
    // The chains as a RELAY over the four 16-lane rows of the wave.  A row = the 16 chain lanes (X: 8, B: 8); row
    // k handles every fourth block, and the terms of the next four blocks are requested a whole round of four
    // blocks ahead of their use -- in registers the other rows' lanes have anyway.  The
    // sixteen accumulators travel from row to row (0 -> 1 -> 3 -> 2 -> 0) with one v_permlane16_swap /
    // v_permlane32_swap per block (tools/permlane_probe.hip).  With ping-pong buffers in sixteen lanes the terms
    // of block blk + 1 were requested only eight dependent multiply-adds before their use: less than an LDS round
    // trip, and the chains are the workgroup's critical path.
    const int relay_row = cl >> 4;
    const int relay_pos = relay_row == 0 ? 0 : relay_row == 1 ? 1 : relay_row == 3 ? 2 : 3;  // place in the relay
    if (cw < 2) {
      __builtin_amdgcn_s_setprio(3);
      const int ch = (cl >> 3) & 1;  // 0: X, 1: B
      const float* src = terms + l * 32;
      int slot[4];
#pragma unroll
      for (int q = 0; q < 4; q++) slot[q] = ((ch * 4 + q) ^ lsw) * 4;
      const int last = nblk - 1;
      // Two register sets, used in turn by ROUNDS of four blocks (one per row): at the start of a round every row
      // requests the block it will handle in the NEXT round -- one wave-wide set of four 16-byte loads, a whole
      // round (32 dependent multiply-adds and four hops) ahead of its use.
      float4 ta[4], tb[4];
#pragma unroll
      for (int q = 0; q < 4; q++) ta[q] = *(const float4*)&src[imin(relay_pos, last) * 256 + slot[q]];
      auto round4 = [&](const float4* t, int first) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (first + j >= nblk) break;  // (wave-uniform)
          // every lane runs the eight steps; only the row that holds the accumulators has meaningful ones
          if (cw == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              acc = fma32(t[q].x, t[q].x, acc);
              acc = fma32(t[q].z, t[q].z, acc);
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              acc = fma32(t[q].x, t[q].y, acc);
              acc = fma32(t[q].z, t[q].w, acc);
            }
          }
          // the accumulators move on: rows 0 -> 1 and 3 -> 2 with a 16-lane swap, 1 -> 3 and 2 -> 0 with a 32-lane one
          const unsigned bits = __float_as_uint(acc);
          if (j == 0) acc = __uint_as_float(__builtin_amdgcn_permlane16_swap(bits, bits, false, false)[0]);
          if (j == 1) acc = __uint_as_float(__builtin_amdgcn_permlane32_swap(bits, bits, false, false)[0]);
          if (j == 2) acc = __uint_as_float(__builtin_amdgcn_permlane16_swap(bits, bits, false, false)[1]);
          if (j == 3) acc = __uint_as_float(__builtin_amdgcn_permlane32_swap(bits, bits, false, false)[1]);
        }
      };
#pragma clang loop unroll(disable)
      for (int blk = 0; blk < nblk; blk += 8) {
#pragma unroll
        for (int q = 0; q < 4; q++) tb[q] = *(const float4*)&src[imin(blk + 4 + relay_pos, last) * 256 + slot[q]];
        round4(ta, blk);
#pragma unroll
        for (int q = 0; q < 4; q++) ta[q] = *(const float4*)&src[imin(blk + 8 + relay_pos, last) * 256 + slot[q]];
        round4(tb, blk + 4);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    const int final_pos = nblk & 3;  // where the accumulators are after the last hop
    const bool chain_lane = cw < 2 && relay_pos == final_pos;
    const int chain_ch = (cl >> 3) & 1;
#else
    if (cw < 2 && cl < 16) {
      __builtin_amdgcn_s_setprio(3);
      const int ch = cl >> 3;  // 0: X, 1: B
      const float* src = terms + l * 32;
      int slot[4];
#pragma unroll
      for (int q = 0; q < 4; q++) slot[q] = ((ch * 4 + q) ^ lsw) * 4;
      // two blocks per iteration, ping-pong buffers (no register copies in the loop)
      float4 ta[4], tb[4];
#pragma unroll
      for (int q = 0; q < 4; q++) ta[q] = *(const float4*)&src[slot[q]];
      const int last = nblk - 1;
      if (cw == 0) {
#pragma clang loop unroll(disable)
        for (int blk = 0; blk < nblk; blk += 2) {
          const int n1 = imin(blk + 1, last), n2 = imin(blk + 2, last);
#pragma unroll
          for (int q = 0; q < 4; q++) tb[q] = *(const float4*)&src[n1 * 256 + slot[q]];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            acc = fma32(ta[q].x, ta[q].x, acc);
            acc = fma32(ta[q].z, ta[q].z, acc);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) ta[q] = *(const float4*)&src[n2 * 256 + slot[q]];
          if (blk + 1 < nblk) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              acc = fma32(tb[q].x, tb[q].x, acc);
              acc = fma32(tb[q].z, tb[q].z, acc);
            }
          }
        }
      } else {
#pragma clang loop unroll(disable)
        for (int blk = 0; blk < nblk; blk += 2) {
          const int n1 = imin(blk + 1, last), n2 = imin(blk + 2, last);
#pragma unroll
          for (int q = 0; q < 4; q++) tb[q] = *(const float4*)&src[n1 * 256 + slot[q]];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            acc = fma32(ta[q].x, ta[q].y, acc);
            acc = fma32(ta[q].z, ta[q].w, acc);
          }
#pragma unroll
          for (int q = 0; q < 4; q++) ta[q] = *(const float4*)&src[n2 * 256 + slot[q]];
          if (blk + 1 < nblk) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              acc = fma32(tb[q].x, tb[q].y, acc);
              acc = fma32(tb[q].z, tb[q].w, acc);
            }
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    const bool chain_lane = cw < 2 && cl < 16;
    const int chain_ch = cl >> 3;
#endif
    const float total = octet_sum(acc);
    // cfl_sum: ca_x, cb_x, ca_b, cb_b
    if (chain_lane && l == 0) S.cfl_sum[chain_ch * 2 + cw] = total;
    __syncthreads();
    if (tid < 2) {  // FindBestMultiplier tail (:56-61)
      const float kDistanceMultiplierAC = 1e-3f;
      const float num = (float)(nblk * 64);
      float xq = -S.cfl_sum[tid * 2 + 1] / (S.cfl_sum[tid * 2] + num * kDistanceMultiplierAC * 0.5f);
      xq = fmaxf(-128.0f, fminf(127.0f, roundf(xq)));
      S.cmap[tid] = (int)xq;
    }
  }
  __syncthreads();
  JXLT_MARK(10);
  const int ytox = S.cmap[0], ytob = S.cmap[1];
  const float kInvColorFactorF = 1.0f / 84;
  const float cmap_x = (float)ytox * kInvColorFactorF;           // YtoXRatio
  const float cmap_b = 1.0f + (float)ytob * kInvColorFactorF;    // YtoBRatio
  if (tid == 0) {
    A.ytox[(size_t)ty_img * A.g.xsize_tiles + tx_img] = (int8_t)ytox;
    A.ytob[(size_t)ty_img * A.g.xsize_tiles + tx_img] = (int8_t)ytob;
  }

  // ---- P6b: entropy estimates (enc_ac_strategy.cc:68-146,187-212) -----------
  float qmax = 0.0f;  // largest quantised magnitude whose root was taken from the table
  if (search) {
    // DCT8 estimate for this octet's own block
    if (blk_valid) {
      const float e = estimate_entropy<8, kLutRoots>(c8x, c8y, c8b, S.inv_w + 0, S.inv_w + 64, S.inv_w + 128, l,
                                                     fmaxf(0.0f, S.aq[oct]), fmaxf(0.0f, S.mask[oct]), cmap_x,
                                                     cmap_b, A.distance, S.sqrt_lut, &qmax);
      const float k8x8mul1 = (float)(-0.55 * 0.75f);
      const float k8x8mul2 = 1.0735757687292623f * 0.75f;
      const float k8x8base = (float)1.4;
      const float mul8x8 = k8x8mul2 + div_normal(k8x8mul1, A.strategy_distance + k8x8base);
      float e8 = 3.0f * mul8x8;
      e8 += mul8x8 * e;
      if (l == 0) S.transpose_pad[((oby >> 1) * 4 + (obx >> 1)) * 8 + (oby & 1) * 2 + (obx & 1)] = e8;
    }
    // The DCT8 coefficients are needed again in P8; they wait in the (now dead) term area
    // while the two-block estimate runs, which would otherwise spill.
    float* park = &S.x[0] + tid;
#pragma unroll
    for (int r = 0; r < 8; r++) {
      park[(r)*kTileThreads] = c8x[r];
      park[(8 + r) * kTileThreads] = c8y[r];
      park[(16 + r) * kTileThreads] = c8b[r];
    }
    JXLT_SCHED_FENCE();
    if (cell_valid) {
      const int o2 = is_tall ? 8 : 1;  // second covered block in the 8x8 tile grid
      const int bi = cby * 8 + cbx;
      const float quant = fmaxf(fmaxf(0.0f, S.aq[bi]), S.aq[bi + o2]);
      const float masking = fmaxf(fmaxf(0.0f, S.mask[bi]), S.mask[bi + o2]);
      const int toff = is_tall ? 3 : 6;
      float qmax16 = 0.0f;
      const float e = estimate_entropy<16, kLutRoots>(c16x, c16y, c16b, S.inv_w + quant_table_offset(toff),
                                                      S.inv_w + quant_table_offset(toff + 1),
                                                      S.inv_w + quant_table_offset(toff + 2), l, quant, masking,
                                                      cmap_x, cmap_b, A.distance, S.sqrt_lut, &qmax16);
      qmax = fmaxf(qmax, qmax16);
      const float k8X16mul1 = (float)-0.55, k8X16mul2 = (float)0.9019587899705066,
                  k8X16base = (float)1.6;
      const float mul16x8 = k8X16mul2 + div_normal(k8X16mul1, A.strategy_distance + k8X16base);
      if (l == 0) S.transpose_pad[cell * 8 + (is_tall ? 4 : 6) + (cand & 1)] = mul16x8 * e;
    }
    JXLT_SCHED_FENCE();
#pragma unroll
    for (int r = 0; r < 8; r++) {
      c8x[r] = park[(r)*kTileThreads];
      c8y[r] = park[(8 + r) * kTileThreads];
      c8b[r] = park[(16 + r) * kTileThreads];
    }
  }
  // A magnitude beyond the root table invalidates this tile's estimates: the frame is then
  // redone by the kernel variant that computes every root (jxlt_capi.hip; practically never).
  if (kLutRoots && (qmax >= (float)kSqrtLutSize || (A.flags & 0x1000u) != 0)) A.lut_overflow[0] = 1u;  // (0x1000: test hook)
  __syncthreads();
  JXLT_MARK(6);
  // ---- P7: decision (:213-237) + AdjustQuantField (:240-266) ------------------
  if (search && tid < 16) {
    const int cx = (tid & 3) * 2, cy = (tid >> 2) * 2;
    if (cx + 1 < nbx && cy + 1 < nby) {
      const float* e = &S.transpose_pad[tid * 8];
      const float e00 = e[0], e01 = e[1], e10 = e[2], e11 = e[3];
      const float l16 = e[4], r16 = e[5], t16 = e[6], b16 = e[7];
      const float cost16x8 = fminf(l16, e00 + e10) + fminf(r16, e01 + e11);
      const float cost8x16 = fminf(t16, e00 + e01) + fminf(b16, e10 + e11);
      const int b00 = cy * 8 + cx;
      if (cost16x8 < cost8x16) {
        if (l16 < e00 + e10) { S.strat[b00] = (1 << 1) | 1; S.strat[b00 + 8] = (1 << 1); }
        if (r16 < e01 + e11) { S.strat[b00 + 1] = (1 << 1) | 1; S.strat[b00 + 9] = (1 << 1); }
      } else {
        if (t16 < e00 + e01) { S.strat[b00] = (2 << 1) | 1; S.strat[b00 + 1] = (2 << 1); }
        if (b16 < e10 + e11) { S.strat[b00 + 8] = (2 << 1) | 1; S.strat[b00 + 9] = (2 << 1); }
      }
      if (kDebug && A.dbg_ent8) {
        const size_t cells_x = (size_t)A.g.xsize_blocks / 2 + 1;
        float* d = A.dbg_ent8 + (((size_t)(by_img0 + cy) / 2) * cells_x + (size_t)(bx_img0 + cx) / 2) * 8;
        for (int k = 0; k < 8; k++) d[k] = e[k];
      }
      // AdjustQuantField for the cell's transforms
      for (int k = 0; k < 4; k++) {
        const int bi = b00 + (k >> 1) * 8 + (k & 1);
        const uint8_t a = S.strat[bi];
        if (!(a & 1) || (a >> 1) == 0) continue;
        const int o2 = (a >> 1) == 1 ? 8 : 1;
        const uint8_t m = S.raw_quant[bi] > S.raw_quant[bi + o2] ? S.raw_quant[bi] : S.raw_quant[bi + o2];
        S.raw_quant[bi] = m;
        S.raw_quant[bi + o2] = m;
      }
    }
  }
  __syncthreads();
  if (tid < 64 && (tid & 7) < nbx && (tid >> 3) < nby) {
    const uint32_t pos = (uint32_t)(by_img0 + (tid >> 3)) * bstride + (uint32_t)(bx_img0 + (tid & 7));
    A.strategy[pos] = S.strat[tid];
    if (S.strat[tid] & 1) atomicAdd(&S.nfirst, 1u);
    A.raw_quant[pos] = S.raw_quant[tid];
  }
  // All pixel reads were done before P5b (the transforms live in registers): from here on the
  // XYB planes are reused as the quantised-coefficient staging area.  No barrier is needed
  // between the stores above (they read S.strat / S.raw_quant, final since the barrier before
  // them) and P8; S.nfirst is read after later barriers.
#ifdef JXLT_P7_SECOND_BARRIER
  __syncthreads();
#endif
  JXLT_MARK(7);
  // ---- P8a: the coefficients of the selected transforms -> LDS ------------------
  // The transforms that the decision kept are quantised in SCAN ORDER by other lanes than the ones that hold
  // them: per tile every block belongs to exactly one selected transform, so "one wave pass = the 64 scan
  // positions of one block and channel" always fills its lanes, whatever the mix of strategies -- while the
  // octets that hold the coefficients are, by construction, idle for every candidate that lost (half of
  // the two-block candidates at best).  Natural layout [block][channel x, y, b][64] of floats, the second
  // half of a two-block transform in its second block's slot.
  float* const stagef = &S.x[0];
  {
    // (within a block and channel the slot of coefficient (row r, column l) is l * 8 + r: a lane's eight rows
    // are two 16-byte stores)
    auto put8 = [&](float* d, const float* v) {
      float4 lo, hi;
      lo.x = v[0]; lo.y = v[1]; lo.z = v[2]; lo.w = v[3];
      hi.x = v[4]; hi.y = v[5]; hi.z = v[6]; hi.w = v[7];
      *reinterpret_cast<float4*>(d) = lo;
      *reinterpret_cast<float4*>(d + 4) = hi;
    };
    if (blk_valid && S.strat[oct] == 1) {  // this octet's own block stayed DCT8
      float* d = stagef + oct * kStageStrideF + l * 8;
      put8(d, c8x);
      put8(d + 64, c8y);
      put8(d + 128, c8b);
    }
    const int bi = cby * 8 + cbx;
    if (cell_valid && S.strat[bi] == (uint8_t)(((is_tall ? 1 : 2) << 1) | 1)) {  // its candidate was selected
      float* da = stagef + bi * kStageStrideF + l * 8;
      float* db = stagef + (bi + (is_tall ? 8 : 1)) * kStageStrideF + l * 8;
      put8(da, c16x);
      put8(da + 64, c16y);
      put8(da + 128, c16b);
      put8(db, c16x + 8);
      put8(db + 64, c16y + 8);
      put8(db + 128, c16b + 8);
    }
  }
  __syncthreads();
  JXLT_MARK(8);

  // ---- P8b + P9: quantise, DC, nzeros, scan-order store (enc_group.cc:166-443) --
  // One wave pass = one selected transform: lane = scan position (the lane's natural coefficient index, and
  // with it its quantisation weights and thresholds, are per-lane constants of the strategy class).  The
  // tile's transforms are dealt out to the waves round robin (in raster order of their first blocks), so every
  // wave has the same number of them whatever the mix of strategies.  Per transform only the per-coefficient
  // work is done at once; the DC values and the per-block outputs are collected per lane (lane j = the wave's
  // j-th transform) and finished in one pass at the end.  Everything else is wave-uniform (scalar unit).
  {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    uint32_t wave_tokens = 0;
    struct LaneConsts {
      float inv[3];  // InvMatrix of x, y, b at the lane's coefficient
      float ydq;     // dequantisation weight of y
      float thr[3];  // zeroing threshold of x, y, b (enc_group.cc:227-242)
    };
    // (per scan position and position class: tables built by the host, DeviceTables::scan_consts)
    auto consts_of = [&](int cls) {
      LaneConsts k;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        k.inv[c] = A.tab->scan_consts[cls][c][lane];
        k.thr[c] = A.tab->scan_consts[cls][4 + c][lane];
      }
      k.ydq = A.tab->scan_consts[cls][3][lane];
      return k;
    };
    const LaneConsts k8 = consts_of(0), k16a = consts_of(1), k16b = consts_of(2);
    const int slot8 = A.tab->scan_slot[0][lane], slot16a = A.tab->scan_slot[1][lane], slot16b = A.tab->scan_slot[2][lane];
    // lane b knows block b of the tile; the first blocks of the tile's transforms as a mask
    const bool lane_blk_valid = (lane & 7) < nbx && (lane >> 3) < nby;
    const int strat_of_lane = lane_blk_valid ? (int)S.strat[lane] : 0;
    const int quant_of_lane = (int)S.raw_quant[lane];
    const float inv_qac_of_lane = A.tab->inv_qac[quant_of_lane];  // (one vector load: no scalar load per transform)
    // (transform number t, in raster order of the first blocks, goes to wave t mod 8: lane b finds its block's
    // number as the count of first blocks below it, and the wave's own blocks come out of one more ballot)
    const unsigned long long firsts = __ballot(strat_of_lane & 1);
    const int rank_of_lane =
        (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(firsts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)firsts, 0u));
    unsigned long long todo = __ballot((strat_of_lane & 1) != 0 && (rank_of_lane & 7) == wave);
    // staged coefficients of the transform whose first block is b: [half a / b][channel x, y, b]
    auto fetch = [&](int b, int st, float (*v)[3]) {
      const int o2 = st == 1 ? 8 : 1;
      const int i0 = st == 0 ? slot8 : slot16a, i1 = slot16b;  // (bit 6: the transform's second block)
      const int src0 = (i0 < 64 ? b : b + o2) * kStageStrideF + (i0 & 63);
      const int src1 = (i1 < 64 ? b : b + o2) * kStageStrideF + (i1 & 63);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        v[0][c] = stagef[src0 + c * 64];
        v[1][c] = st != 0 ? stagef[src1 + c * 64] : 0.0f;
      }
    };
    auto scalar_lane = [&](int v, int l_) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l_)); };
    // collected per transform (lane j = the wave's j-th transform)
    // (the two lowest frequencies of what was quantised go through LDS: lanes 0 and 1 store them, lane j reads
    // its transform's six values at the end -- [wave][transform][channel][2] floats behind the staging area)
    float* const dc_stage = stagef + 64 * kStageStrideF;            // [transform][channel][2]
    int* const tr_info = reinterpret_cast<int*>(dc_stage + 64 * 6);  // [3][transform]: block | strategy << 8, nzeros, nscan
    float* const lane_dump = dc_stage + 64 * 6 + 3 * 64 + lane;      // where the stores of the lanes that have nothing to say go
    // (lane 0 files a wave-uniform value under the transform's number)
    auto file_int = [&](int which, int t, int v) {
      (lane == 0 ? tr_info + which * 64 + t : reinterpret_cast<int*>(lane_dump))[0] = v;
    };
    // (a use of the loaded value here: the wait for it belongs in front of the loop -- inside, where loads and
    // stores share one counter, it would wait for the previous transform's coefficient stores every time)
    JXLT_TOUCH_VGPR(inv_qac_of_lane);
    for (const LaneConsts* k : {&k8, &k16a, &k16b}) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        JXLT_TOUCH_VGPR(k->inv[c]);
        JXLT_TOUCH_VGPR(k->thr[c]);
      }
      JXLT_TOUCH_VGPR(k->ydq);
    }
    JXLT_TOUCH_VGPR(slot8);
    JXLT_TOUCH_VGPR(slot16a);
    JXLT_TOUCH_VGPR(slot16b);
    int ntrans = 0;
    float next_v[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    int next_b = todo != 0 ? (int)__builtin_ctzll(todo) : -1, next_st = 0;
    if (next_b >= 0) {
      next_st = scalar_lane(strat_of_lane, next_b) >> 1;
      fetch(next_b, next_st, next_v);
    }
    while (next_b >= 0) {
      const int b = __builtin_amdgcn_readfirstlane(next_b), st = __builtin_amdgcn_readfirstlane(next_st);
      float in[2][3];
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 3; c++) in[h][c] = next_v[h][c];
      todo &= todo - 1;
      next_b = todo != 0 ? (int)__builtin_ctzll(todo) : -1;
      if (next_b >= 0) {
        next_st = scalar_lane(strat_of_lane, next_b) >> 1;
        fetch(next_b, next_st, next_v);  // (requested before this transform is worked on)
      }
      const bool two = st != 0;
      const int covered = two ? 2 : 1;
      const int quant_ac = scalar_lane(quant_of_lane, b);
      const float qac = A.scale * quant_ac;
      const float inv_qac = __int_as_float(scalar_lane(__float_as_int(inv_qac_of_lane), b));
      const uint32_t pos0 = (uint32_t)(by_img0 + (b >> 3)) * bstride + (uint32_t)(bx_img0 + (b & 7));
      const uint32_t pos1 = pos0 + (st == 1 ? bstride : 1u);
      // per scan position: y first (its round trip feeds the chroma channels, :392-425)
      float quant[2][3], cur0[3];  // quantised values (integer-valued); first half of what was quantised
      auto half = [&](const LaneConsts& k, const float* v, float* q, float* cur) {
        auto quantise = [&](int c, float x, float quantv) {
          const float qq = k.inv[c] * quantv;
          const float val = qq * x;
          return fabsf(val) >= k.thr[c] ? rintf(val) : 0.0f;
        };
        q[1] = quantise(1, v[1], qac * 1.0f);
        const float y_back = (adjust_quant_bias_y(q[1]) * k.ydq) * inv_qac;
        const float cx_ = nfma32(cmap_x, y_back, v[0]), cb_ = nfma32(cmap_b, y_back, v[2]);
        q[0] = quantise(0, cx_, qac * A.x_qm_mul);
        q[2] = quantise(2, cb_, qac * (float)1.0);
        if (cur) {
          cur[0] = cx_;
          cur[1] = v[1];
          cur[2] = cb_;
        }
      };
      if (two) {
        half(k16a, in[0], quant[0], cur0);
        half(k16b, in[1], quant[1], nullptr);
      } else {
        half(k8, in[0], quant[0], cur0);
        quant[1][0] = quant[1][1] = quant[1][2] = 0.0f;
      }
      const int t = wave + 8 * ntrans;  // the transform's number in the tile
      file_int(0, t, b | (st << 8));
      int nz_packed = 0, nscan_packed = 0;
      // nzeros (enc_group.cc:51-148) and the scan position behind the last nonzero coefficient: the six ballots
      // first, then the scalar arithmetic on them, then the stores (no compare -> scalar -> compare round trip
      // per channel)
      const unsigned long long llf_mask = covered == 2 ? 3ull : 1ull;  // scan positions < covered: coded as DC
      unsigned long long m0[3], m1[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        (lane < 2 ? dc_stage + (t * 3 + c) * 2 + lane : lane_dump)[0] = cur0[c];
        m0[c] = __ballot(quant[0][c] != 0.0f) & ~llf_mask;
        m1[c] = two ? __ballot(quant[1][c] != 0.0f) : 0ull;
      }
      int nscan[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int nzeros = __popcll(m0[c]) + __popcll(m1[c]);
        nscan[c] = m1[c] != 0 ? 128 - __clzll((long long)m1[c]) : m0[c] != 0 ? 64 - __clzll((long long)m0[c]) : 0;
        nz_packed |= nzeros << (8 * c);
        nscan_packed |= nscan[c] << (8 * c);
        wave_tokens += 1 + (nscan[c] > covered ? nscan[c] - covered : 0);
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        // (the tokeniser takes "nonzeros still to come" and "previous coefficient nonzero" from these masks:
        // lanes 0 and 1 store the two words)
        if (lane < 2) A.blk_nzmask[(size_t)(pos0 * 3 + c) * 2 + lane] = lane == 0 ? m0[c] : m1[c];
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        // only scan positions below nscan (= up to the last nonzero) are ever read again
        int16_t* const out0 = A.coef_scan + (size_t)(pos0 * 3 + c) * 64;
        int16_t* const out1 = A.coef_scan + (size_t)(pos1 * 3 + c) * 64;
        if (lane < nscan[c]) out0[lane] = (int16_t)(int)quant[0][c];
        if (64 + lane < nscan[c]) out1[lane] = (int16_t)(int)quant[1][c];
      }
      file_int(1, t, nz_packed);
      file_int(2, t, nscan_packed);
      ntrans++;
    }
    if (lane == 0 && wave_tokens) atomicAdd(&S.ntok, wave_tokens);
  }
  __syncthreads();
  // the tile's transforms side by side, one per lane of wave 0: DC of the covered blocks (:392-443) and the
  // per-block outputs
  if (tid < 64) {
    const int lane = tid;
    float* const dc_stage = stagef + 64 * kStageStrideF;
    const int* const tr_info = reinterpret_cast<const int*>(dc_stage + 64 * 6);
    const bool lane_blk_valid = (lane & 7) < nbx && (lane >> 3) < nby;
    const int ntrans = __popcll(__ballot(lane_blk_valid && (S.strat[lane] & 1) != 0));
    const int col_block = tr_info[lane], col_nz = tr_info[64 + lane], col_nscan = tr_info[128 + lane];
    if (lane < ntrans) {
      const int b = col_block & 0xFF, st = col_block >> 8;
      const bool two = st != 0;
      const uint32_t pos0 = (uint32_t)(by_img0 + (b >> 3)) * bstride + (uint32_t)(bx_img0 + (b & 7));
      const uint32_t pos1 = pos0 + (st == 1 ? bstride : 1u);
      const float kScale1 = (float)0.901764195028874394;
      const float kInvDCQuant[3] = {4096.0f, 512.0f, 256.0f};
      int16_t dcy_a = 0, dcy_b = 0;
#pragma unroll
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;  // y first: the chroma DC is coded relative to it
        const float c0 = dc_stage[(lane * 3 + c) * 2], c1 = dc_stage[(lane * 3 + c) * 2 + 1];
        const float b0 = c0 * 1.0f * 1.0f, b1 = c1 * 1.0f * kScale1;
        const float d_a = two ? b0 + b1 : c0, d_b = two ? b0 - b1 : 0.0f;
        int16_t qdc_a, qdc_b;
        if (c == 1) {
          const float inv_factor_y = kInvDCQuant[1] * A.scale_dc;
          qdc_a = dcy_a = (int16_t)roundf(inv_factor_y * d_a);
          qdc_b = dcy_b = (int16_t)roundf(inv_factor_y * d_b);
        } else {
          const float inv_factor = (c == 0 ? kInvDCQuant[0] : kInvDCQuant[2]) * A.scale_dc;
          const float cfl_factor = c == 0 ? 0.0f : kInvDCQuant[2] * (1.0f / kInvDCQuant[1]);
          qdc_a = (int16_t)roundf(d_a * inv_factor - dcy_a * cfl_factor);
          qdc_b = (int16_t)roundf(d_b * inv_factor - dcy_b * cfl_factor);
        }
        // (select, not A.nzgrid[c] / A.quant_dc[c]: indexing a kernel-argument array by a runtime value
        // would force the argument block into scratch memory)
        uint8_t* nzg = c == 0 ? A.nzgrid[0] : c == 1 ? A.nzgrid[1] : A.nzgrid[2];
        int16_t* qdc = c == 0 ? A.quant_dc[0] : c == 1 ? A.quant_dc[1] : A.quant_dc[2];
        const int nzeros = (col_nz >> (8 * c)) & 0xFF;
        qdc[pos0] = qdc_a;
        A.blk_nz[pos0 * 3 + c] = (uint8_t)nzeros;
        A.blk_nscan[pos0 * 3 + c] = (uint8_t)((col_nscan >> (8 * c)) & 0xFF);
        if (!two) {
          nzg[pos0] = (uint8_t)nzeros;
        } else {
          qdc[pos1] = qdc_b;
          const uint8_t shifted = (uint8_t)((nzeros + 1) >> 1);
          nzg[pos0] = shifted;
          nzg[pos1] = shifted;
        }
      }
    }
  }
  JXLT_MARK(9);
  if (tid == 0) {
    const int group = (ty_img >> 2) * A.g.xsize_groups + gx;
    atomicAdd(&A.group_ntok[group], S.ntok);
    const int dcg = (ty_img >> 5) * ((A.g.xsize + 2047) / 2048) + (tx_img >> 5);
    atomicAdd(&A.dc_nac[dcg], S.nfirst);
  }
#undef JXLT_MARK
#undef JXLT_STOP
#undef SX
#undef SY
}

// tile_kernel: roots of the entropy estimate from the LDS table (the product path);
// tile_kernel_exact_roots: every root computed -- the same results, needed only for frames in
// which tile_kernel met a quantised magnitude beyond the table.
__global__ void __launch_bounds__(kTileThreads, 4) tile_kernel(const TileArgs A) { tile_kernel_body<true, false>(A); }
__global__ void __launch_bounds__(kTileThreads, 4) tile_kernel_debug(const TileArgs A) { tile_kernel_body<true, true>(A); }
__global__ void __launch_bounds__(kTileThreads, 4) tile_kernel_exact_roots(const TileArgs A) {
  tile_kernel_body<false, true>(A);
}

// ---------------------------------------------------------------------------
// Exclusive scan of up to a few ten thousand 32-bit counts into 64-bit offsets (single workgroup):
// offsets[i] = counts[0] + ... + counts[i - 1], offsets[n] = the total.  Every thread owns a contiguous run
// (all of its loads in flight together), one wave scan + one barrier for the runs' totals.
// ---------------------------------------------------------------------------
constexpr int kScanThreads = 1024;
constexpr int kScanMaxPerThread = 32;  // n <= 32 768 (the C ABI's frames have <= 16 448 sections of a kind)
__global__ void __launch_bounds__(kScanThreads) group_scan_kernel(const uint32_t* counts, uint64_t* offsets, int n) {
  __shared__ uint64_t wave_total[kScanThreads / 64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + kScanThreads - 1) / kScanThreads;
  const int beg = tid * per, end = imin(n, beg + per);
  uint32_t v[kScanMaxPerThread];
  uint64_t mine = 0;
#pragma unroll
  for (int k = 0; k < kScanMaxPerThread; k++) {
    v[k] = (k < per && beg + k < end) ? counts[beg + k] : 0u;
    mine += v[k];
  }
  uint64_t incl = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  uint64_t run = incl - mine;
  for (int w = 0; w < wave; w++) run += wave_total[w];
#pragma unroll
  for (int k = 0; k < kScanMaxPerThread; k++) {
    if (k < per && beg + k < end) {
      offsets[beg + k] = run;
      run += v[k];
    }
  }
  if (tid == kScanThreads - 1) offsets[n] = run;
}

// ---------------------------------------------------------------------------
// Token kernel: one workgroup per 256x256 group (enc_group.cc:444-494)
// ---------------------------------------------------------------------------
constexpr int kTokenThreads = 512;

// A lane per COEFFICIENT TOKEN of the group's stream (window w = tokens 64 w .. 64 w + 63, whatever entries
// they belong to): a lane finds its block through the per-window index of the block that holds the window's
// first token plus the block starts inside the window, and needs nothing from its neighbours -- "nonzeros still
// to come" and "previous coefficient nonzero" come from the nonzero masks tile_kernel leaves per entry.  (A
// 64-lane pass per entry, the structure until the end of round 2, filled 18 % of its lane slots on ordinary
// content -- 39 tokens per Y entry, 0.6 per chroma entry -- and kept the CU's one scalar unit busy with
// per-entry bookkeeping.)  The nzeros tokens (one per entry) are written by a thread-per-block pass.
__global__ void __launch_bounds__(kTokenThreads) token_kernel(const TokenArgs A) {
  // per block, two words: strategy byte | nzeros y << 8 | nscan y << 16 | nzeros x << 24, nscan x | nzeros b << 8 |
  // nscan b << 16 (the entries in stream order y, x, b; 40 KB of LDS in all: four workgroups per CU)
  __shared__ uint2 meta[1024];
  // per block: coefficient tokens in front of it (bits 0-19) | first blocks in front of it (bits 20-30)
  __shared__ uint32_t bstart[1024 + 1];
  __shared__ uint16_t first_blk[3072 + 8];  // per window: the block that holds its first coefficient token
  __shared__ uint32_t wsum[kTokenThreads / 64];
  __shared__ uint32_t hist[64 * 64];
  __shared__ uint16_t s_nnz_ctx[64], s_freq_ctx[64];
  __shared__ uint8_t s_ctx_map[1980];
  // nzeros grid of the group (PredictFromTopAndLeft input of the nzeros tokens); once those are written its first
  // kilobyte is `boundary`: per wave, the block that starts at a position of its window
  __shared__ alignas(4) uint8_t s_nzg[3 * 1024];
  uint16_t (*const boundary)[64] = reinterpret_cast<uint16_t (*)[64]>(&s_nzg[0]);
  __shared__ uint64_t s_group_base;
  __shared__ uint64_t gsum[kTokenThreads / 64];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DeviceTables* T = A.tab;
  const int group = A.group_first + (int)blockIdx.x;
  const bool do_hist = A.histogram != nullptr;
  // LDS histogram slot of (pre-clustered context, symbol): the symbol is rotated by the context, so that the
  // small symbols nearly all tokens have do not land in the same few banks for every context
#ifdef JXLT_HIST_PLAIN
  auto hist_slot = [](uint32_t cm, uint32_t sym) { return cm * 64u + sym; };
#else
  auto hist_slot = [](uint32_t cm, uint32_t sym) { return cm * 64u + ((sym + cm) & 63u); };
#endif
  if (do_hist)
    for (int i = tid; i < 64 * 64; i += kTokenThreads) hist[i] = 0;
  for (int i = tid; i < 1980; i += kTokenThreads) s_ctx_map[i] = T->ac_context_map[i];
  if (tid < 64) {
    s_nnz_ctx[tid] = T->nnz_context[tid];
    s_freq_ctx[tid] = T->freq_context[tid];
  }
  // Where the group's tokens start: the sum of the counts of all groups before it (<= 16 384 counts, 64 KB,
  // one round of loads; a scan kernel in front of this one cost 22 us of the step for the same numbers).
  {
    uint64_t part = 0;
    for (int i = tid; i < group; i += kTokenThreads) part += A.group_ntok[i];
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    if (lane == 0) gsum[wave] = part;
  }
  const int ggx = group % A.g.xsize_groups, ggy = group / A.g.xsize_groups;
  const int bx0 = ggx * 32, by0 = ggy * 32;
  const int nbx = imin(32, A.g.xsize_blocks - bx0), nby = imin(32, A.g.ysize_blocks - by0);
  // 32-bit block / record indices (the C ABI limits a frame to 2^24 blocks, a group's tokens to
  // 196 608 records): addresses are scalar base + 32-bit lane offset, no 64-bit vector arithmetic
  const uint32_t bstride = (uint32_t)A.g.xsize_blocks;
  const uint32_t nbx_magic = 65536u / (uint32_t)nbx + 1u;  // b / nbx == (b * magic) >> 16 for b < 1024, nbx <= 32

  // metadata per entry, coefficient-token and first-block counts per block, predicted-nzeros grid -> LDS
  const int nblk = nbx * nby;
  // (a group has at most 1024 blocks, two per thread: all ten bytes of both are requested before the first is
  // used, whether the block turns out to be a first block or not -- one round trip instead of four)
  static_assert(2 * kTokenThreads >= 1024, "two blocks per thread");
  {
    uint32_t ld_a[2], ld_nscan[2][3], ld_nz[2][3], ld_grid[2][3];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int b = imin(tid + r * kTokenThreads, nblk - 1);
      const int by = (int)(((uint32_t)b * nbx_magic) >> 16), bx = b - by * nbx;
      const uint32_t pos = (uint32_t)(by0 + by) * bstride + (uint32_t)(bx0 + bx);
      ld_a[r] = A.strategy[pos];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        ld_nscan[r][c] = A.blk_nscan[pos * 3 + c];
        ld_nz[r][c] = A.blk_nz[pos * 3 + c];
        ld_grid[r][c] = A.nzgrid[c][pos];
      }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int b = tid + r * kTokenThreads;
      if (b >= nblk) continue;
      const uint32_t a = ld_a[r];
      const int covered = (a >> 1) == 0 ? 1 : 2;
      uint32_t ncoef = 0;
      uint32_t nzs[3] = {0, 0, 0}, nscans[3] = {0, 0, 0};  // in stream order y, x, b
#pragma unroll
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
        if (a & 1) {
          nscans[ci] = ld_nscan[r][c];
          nzs[ci] = ld_nz[r][c];
          ncoef += nscans[ci] > (uint32_t)covered ? nscans[ci] - covered : 0;
        }
        s_nzg[c * 1024 + b] = (uint8_t)ld_grid[r][c];
      }
      uint2 mw;
      mw.x = a | (nzs[0] << 8) | (nscans[0] << 16) | (nzs[1] << 24);
      mw.y = nscans[1] | (nzs[2] << 8) | (nscans[2] << 16);
      meta[b] = mw;
      bstart[b + 1] = ncoef | ((a & 1) << 20);
    }
  }
  if (tid == 0) bstart[0] = 0;
  __syncthreads();
  if (tid == 0) {
    uint64_t base = 0;
    for (int w = 0; w < kTokenThreads / 64; w++) base += gsum[w];
    s_group_base = base;
    A.group_tok_offset[group] = base;
    const int ngroups = A.g.xsize_groups * A.g.ysize_groups;
    if (group + 1 == ngroups) A.group_tok_offset[ngroups] = base + A.group_ntok[group];
  }
  // inclusive scan over bstart[1..nblk] (blocked: each thread owns a contiguous run; both fields at once: the
  // sums stay inside their bit ranges, <= 387 072 coefficient tokens and <= 1024 first blocks)
  {
    const int per = (nblk + kTokenThreads - 1) / kTokenThreads;
    const int beg = 1 + tid * per, end = imin(1 + nblk, beg + per);
    uint32_t sum = 0;
    for (int i = beg; i < end; i++) sum += bstart[i];
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wsum[w];
    uint32_t run = wbase + incl - sum;
    for (int i = beg; i < end; i++) {
      run += bstart[i];
      bstart[i] = run;
    }
  }
  __syncthreads();
  // bstart[b] is now what lies in front of block b (bstart[nblk]: the totals)
  const uint32_t kLow = (1u << 20) - 1u;
  const uint32_t ncoef_total = bstart[nblk] & kLow;
  const int nwin = (int)((ncoef_total + 63u) >> 6);
  uint8_t* out = A.tokens + 3 * s_group_base;  // (written before the two barriers of the scan above)
  // per block: its windows' index entries and the nzeros tokens of its three entries
  for (int b = tid; b < nblk; b += kTokenThreads) {
    const uint32_t here = bstart[b], next = bstart[b + 1];
    const uint32_t s0 = here & kLow, s1 = next & kLow;
    for (uint32_t q = (s0 + 63u) >> 6; (q << 6) < s1; q++) first_blk[q] = (uint16_t)b;  // (windows that start in b)
    const uint2 mw = meta[b];
    if (!(mw.x & 1)) continue;  // not the first block of a transform: no entries
    // (per entry: nzeros << 8 | nscan << 16)
    const uint32_t mb[3] = {mw.x & 0xFFFF00u, ((mw.x >> 16) & 0xFF00u) | ((mw.y & 0xFFu) << 16), (mw.y & 0xFFFF00u)};
    const int st = (int)((mw.x >> 1) & 0x7F);
    const int covered = st == 0 ? 1 : 2;
    const int bctx_y = st == 0 ? 0 : 1, bctx_c = 2 + bctx_y;  // (ac_context.h:64-114, see below)
    const int cby = (int)(((uint32_t)b * nbx_magic) >> 16), cbx = b - cby * nbx;
    uint32_t tl = s0 + 3u * (here >> 20);  // tokens in front of the block: coefficient tokens + three per first block
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
      const int nzl = (int)((mb[ci] >> 8) & 0xFF), nsc = (int)(mb[ci] >> 16);
      // PredictFromTopAndLeft (enc_group.cc:150-160), default 32
      int pred;
      const uint8_t* nzg = &s_nzg[c * 1024 + b];
      if (cbx == 0) pred = cby == 0 ? 32 : nzg[-nbx];
      else if (cby == 0) pred = nzg[-1];
      else pred = (nzg[-nbx] + nzg[-1] + 1) / 2;
      const int bucket = pred < 8 ? pred : pred >= 64 ? 36 : 4 + pred / 2;
      const int ctx = bucket * 4 + (ci == 0 ? bctx_y : bctx_c);
      uint8_t* o = out + 3u * tl;
      const uint8_t cm = s_ctx_map[ctx];
      o[0] = cm;
      o[1] = (uint8_t)(nzl & 0xFF);
      o[2] = (uint8_t)(nzl >> 8);
      if (do_hist) {
        atomicAdd(&hist[hist_slot(cm, hybrid_uint_symbol((uint32_t)nzl))], 1u);
      }
      tl += 1 + (nsc > covered ? nsc - covered : 0);
    }
  }
  __syncthreads();  // the window index is complete

  // ---- the coefficient tokens: wave w takes windows w, w + 8, ... ---------------------------------------------
  // Three stages per window: "locate" finds the lane's block, entry and scan position (LDS only), "request" asks
  // for its coefficient and its entry's nonzero masks, "emit" (when the values have arrived) derives the context
  // and stores the record.  The lanes behind the stream's end (last window only) repeat
  // the last token, so that every emit issues the same three stores -- the wait for the next window's loads
  // can then be a count (loads and stores share one counter) instead of "everything".
  constexpr int kWaves = kTokenThreads / 64;
  uint16_t* const bnd = &boundary[wave][0];
  struct Located {
    uint32_t out_index;  // the token's place in the group's stream
    int k;               // scan position
    int nzeros;          // of its entry
    int st_ci;           // strategy code | channel in stream order << 8 | counts (a real token) << 16
    uint32_t coef_at, mask_at;  // where its coefficient / its entry's nonzero masks are (element / word index)
    int coef;            // requested
    uint32_t nz[4];      // requested: the entry's nonzero masks, positions covered .. 127
  };
  auto locate = [&](int q, Located& t) {
    const uint32_t w0 = (uint32_t)q << 6;  // the window's first token
    const bool real = w0 + (uint32_t)lane < ncoef_total;
    const uint32_t i = real ? w0 + (uint32_t)lane : ncoef_total - 1u;
    const int b0 = (int)first_blk[q];      // (wave-uniform)
    // the blocks that start inside the window, filed under the position of their first token
    bnd[lane] = 0;
    JXLT_WAVE_SYNC();
    for (int base = b0 + 1; base < nblk; base += 64) {
      const int bb = base + lane;
      if (bb < nblk) {
        const uint32_t s0 = bstart[bb] & kLow, s1 = bstart[bb + 1] & kLow;
        if (s1 > s0 && s0 > w0 && s0 < w0 + 64u) bnd[s0 - w0] = (uint16_t)bb;
      }
      const int last = imin(base + 63, nblk - 1);  // (beyond the window from here on?)
      if ((bstart[last + 1] & kLow) >= w0 + 64u) break;
    }
    JXLT_WAVE_SYNC();  // (the wave's LDS operations execute in order)
    const unsigned long long bm = __ballot(bnd[lane] != 0);
    // the lane's block: the one filed at the highest position <= its token's, else the window's first block
    int blk;
    {
      const int at_most = (int)(i - w0);
      const uint32_t lo = (uint32_t)bm, hi = (uint32_t)(bm >> 32);
      const uint32_t below_lo = at_most < 32 ? lo & ((2u << at_most) - 1u) : lo;
      const uint32_t below_hi = at_most < 32 ? 0u : hi & ((2u << (at_most - 32)) - 1u);
      const int at = below_hi ? 63 - __clz((int)below_hi) : below_lo ? 31 - __clz((int)below_lo) : -1;
      blk = at >= 0 ? (int)bnd[at] : b0;
    }
    JXLT_WAVE_SYNC();  // (read before the next window's entries are filed)
    const uint32_t here = bstart[blk];
    const uint32_t in_block = i - (here & kLow);  // coefficient token of the block
    const uint2 mw = meta[blk];
    // (per entry: nzeros << 8 | nscan << 16)
    const uint32_t m_y = mw.x & 0xFFFF00u, m_x = ((mw.x >> 16) & 0xFF00u) | ((mw.y & 0xFFu) << 16), m_b = mw.y & 0xFFFF00u;
    const int st = (int)((mw.x >> 1) & 0x7F);
    const int covered = st == 0 ? 1 : 2;
    const uint32_t n_y = imax((int)(m_y >> 16) - covered, 0), n_x = imax((int)(m_x >> 16) - covered, 0);
    const int ci = (in_block >= n_y ? 1 : 0) + (in_block >= n_y + n_x ? 1 : 0);  // y, x, b in stream order
    const uint32_t m_e = ci == 0 ? m_y : ci == 1 ? m_x : m_b;
    t.k = covered + (int)(in_block - (ci == 0 ? 0u : ci == 1 ? n_y : n_y + n_x));
    t.nzeros = (int)((m_e >> 8) & 0xFF);
    t.st_ci = st | (ci << 8) | ((real ? 1 : 0) << 16);
    // tokens in front of the block (coefficient tokens + three per first block), the nzeros tokens of this
    // and the earlier entries of the block, the coefficient tokens of the block in front of this one
    t.out_index = (here & kLow) + 3u * (here >> 20) + (uint32_t)(ci + 1) + in_block;
    const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
    const int cby = (int)(((uint32_t)blk * nbx_magic) >> 16), cbx = blk - cby * nbx;
    const uint32_t pos = (uint32_t)(by0 + cby) * bstride + (uint32_t)(bx0 + cbx);
    const uint32_t pos1 = pos + (st == 1 ? bstride : 1u);
    t.coef_at = t.k < 64 ? (pos * 3 + (uint32_t)c) * 64 + (uint32_t)t.k : (pos1 * 3 + (uint32_t)c) * 64 + (uint32_t)(t.k - 64);
    t.mask_at = (pos * 3 + (uint32_t)c) * 4;
  };
  auto request = [&](Located& t) {
    t.coef = (int)A.coef_scan[t.coef_at];
    const uint32_t* nzw = reinterpret_cast<const uint32_t*>(A.blk_nzmask) + t.mask_at;
#pragma unroll
    for (int j = 0; j < 4; j++) t.nz[j] = nzw[j];
  };
  auto emit = [&](const Located& t) {
    const int st = t.st_ci & 0xFF, ci = (t.st_ci >> 8) & 0xFF;
    const bool real = (t.st_ci >> 16) != 0;
    const int covered = st == 0 ? 1 : 2;
    const int log2c = covered == 1 ? 0 : 1;
    const int size = covered * 64;
    const int k = t.k;
    const unsigned long long nz0 = (unsigned long long)t.nz[0] | ((unsigned long long)t.nz[1] << 32);
    const unsigned long long nz1 = (unsigned long long)t.nz[2] | ((unsigned long long)t.nz[3] << 32);
    // nonzeros at the scan positions in front of k (the masks hold positions covered .. 127), previous one
    const int below = k <= 64 ? __popcll(k == 64 ? nz0 : nz0 & ((1ull << k) - 1ull))
                              : __popcll(nz0) + __popcll(nz1 & ((1ull << (k - 64)) - 1ull));
    const int prev = k <= 64 ? (int)((nz0 >> (k - 1)) & 1ull) : (int)((nz1 >> (k - 65)) & 1ull);
    const int left = t.nzeros - below;  // nzeros still to come at this position
    const int nl = (left + covered - 1) >> log2c;
    const int zidx = s_nnz_ctx[nl] + s_freq_ctx[k >> log2c];
    const int pp = k == covered ? ((t.nzeros > size / 16) ? 0 : 1) : prev;
    // block context (ac_context.h:64-114): kBlockContextMap[c*27 + code] is 0/1 for Y and
    // 2/3 for X,B, the odd value for the two-block strategy codes 6 and 7
    const int bctx = (st == 0 ? 0 : 1) + (ci == 0 ? 0 : 2);
    const int ctx = 4 * 37 + 458 * bctx + zidx * 2 + pp;
    const uint8_t cm = s_ctx_map[ctx];
    const uint32_t val = pack_signed((int32_t)t.coef);
    uint8_t* o = out + 3u * t.out_index;
    o[0] = cm;
    o[1] = (uint8_t)(val & 0xFF);
    o[2] = (uint8_t)((val >> 8) & 0xFF);
    if (do_hist && real) {
      const uint32_t slot = hist_slot(cm, hybrid_uint_symbol(val & 0xFFFFu));
      atomicAdd(&hist[slot], 1u);
    }
  };
  // Three windows in flight per wave: window q is emitted (the first use of its requested values: with loads
  // and stores on one counter that is a wait for everything the wave has issued), window q + 8 is requested,
  // window q + 16 is located -- so that a wait comes a whole "locate" (LDS round trips only) behind the last
  // request and the last stores.  Two sets of registers used in turn, no copies (a copy would be a use).
  Located even, odd;
  if (wave < nwin) {
    locate(wave, even);
    request(even);
    if (wave + kWaves < nwin) locate(wave + kWaves, odd);
  }
  for (int q = wave; q < nwin; q += 2 * kWaves) {
    emit(even);
    if (q + kWaves >= nwin) break;
    request(odd);
    if (q + 2 * kWaves < nwin) locate(q + 2 * kWaves, even);
    emit(odd);
    if (q + 2 * kWaves >= nwin) break;
    request(even);
    if (q + 3 * kWaves < nwin) locate(q + 3 * kWaves, odd);
  }
  if (do_hist) {
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += kTokenThreads) {
      const uint32_t n = hist[hist_slot((uint32_t)i >> 6, (uint32_t)i & 63u)];
      if (n) atomicAdd(&A.histogram[i], n);
    }
  }
}

// ---------------------------------------------------------------------------
// Section bit packing: one workgroup per section (enc_frame.cc:784-800 with
// WriteToken, enc_entropy_code.h:34-42).  Tiles of kPackTile records: every
// thread owns kPackPerThread consecutive records, a block scan of their bit
// lengths gives its bit offset, bits are OR-ed into an LDS window that is then
// flushed with coalesced dword stores.
// ---------------------------------------------------------------------------
constexpr int kPackThreads = 512;
constexpr int kPackPerThread = 8;
constexpr int kPackTile = kPackThreads * kPackPerThread;        // 4096 records
constexpr int kPackWindowWords = kPackTile * 28 / 32 + 4;        // <= 28 bits per record

// ---------------------------------------------------------------------------
// Copy-free packing at tile granularity (kPackTile records per workgroup, whatever section they
// belong to): DC-group sections hold ~100 tiles each, AC-group sections <= 48, so per-section
// workgroups leave most of the machine idle on the 64 DC sections of a 16384^2 frame.
//   pack_tile_count_kernel    tiles per section            (+ group_scan_kernel -> tile_base)
//   pack_tile_plan_kernel     per section: the record range of each of its tiles
//   pack_tile_measure_kernel  bit length of every tile
//   pack_tile_offsets_kernel  per section: bit offset of each tile, section bits / bytes
//                             (+ group_scan_kernel -> byte offset of each section)
//   pack_tile_finalize_kernel per tile: absolute bit positions
//   pack_tile_write_kernel    entropy-codes a tile at its final bit position of the blob
// (a tile's workgroup finds everything it needs in one 32-byte PackTileInfo: no dependent
// global loads in front of the record loads)
// Two tiles of a section, or two byte-aligned sections, meet inside a dword: those dwords (a tile's
// first and last) are zeroed by pack_tile_finalize_kernel and OR-ed into by both neighbours; every
// other dword is stored by exactly one workgroup with plain stores.  (Round 1 had the later tile
// re-derive its predecessor's trailing bits instead -- one wave walking back through up to 64 records
// while seven waited, 64 extra records staged per tile, two more barriers.)  Up to 3 bytes behind a
// blob's last section are zeroed.
// ---------------------------------------------------------------------------
struct alignas(16) PackTileInfo {
  uint64_t rec_first;      // absolute index of the tile's first record
  uint64_t bit_pos;        // bit position of the tile in the blob (section-relative until finalised)
  uint64_t sec_start_bit;  // bit position of the tile's section in the blob (section index until finalised)
  uint32_t n_last;         // records in the tile | last tile of its section << 31
  uint32_t before;         // records of the section in front of the tile
};

struct PackTileArgs {
  const uint8_t* records;           // 3-byte records
  const uint64_t* sec_rec_offset;   // [nsec (+1)] first record of each section
  const uint32_t* sec_rec_count;    // optional [nsec] (else offset[s+1] - offset[s])
  int nsec;
  const uint32_t* code_table;       // [64][64]: (depth << 16) | bits
  uint32_t* sec_tiles;              // [nsec] tiles per section
  const uint64_t* tile_base;        // [nsec + 1] exclusive scan of sec_tiles
  uint32_t* tile_bits;              // [tiles] bit length of each tile
  PackTileInfo* tile_info;          // [tiles] where each tile's records and bits are
  uint32_t* sec_bits;               // [nsec]
  uint32_t* sec_bytes;              // [nsec]
  const uint64_t* sec_byte_offset;  // [nsec + 1] exclusive scan of sec_bytes
  uint8_t* out;                     // blob (4-byte aligned)
  uint32_t tile_first;              // first tile of this launch
  uint32_t tile_end;                // one past the last tile of this launch (clamped to the tile count)
};

JXLT_DI uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
JXLT_DI uint32_t pack_section_records(const PackTileArgs& A, int sec) {
  return A.sec_rec_count ? A.sec_rec_count[sec] : (uint32_t)(A.sec_rec_offset[sec + 1] - A.sec_rec_offset[sec]);
}

__global__ void __launch_bounds__(256) pack_tile_count_kernel(const PackTileArgs A) {
  const int s = (int)(blockIdx.x * 256 + threadIdx.x);
  if (s < A.nsec) A.sec_tiles[s] = (pack_section_records(A, s) + kPackTile - 1) / kPackTile;
}

__global__ void __launch_bounds__(256) pack_tile_plan_kernel(const PackTileArgs A) {
  const int s = (int)(blockIdx.x * 256 + threadIdx.x);
  if (s >= A.nsec) return;
  const uint32_t t0 = (uint32_t)A.tile_base[s], t1 = (uint32_t)A.tile_base[s + 1];
  const uint32_t cnt = pack_section_records(A, s);
  const uint64_t rec0 = A.sec_rec_offset[s];
  for (uint32_t t = t0; t < t1; t++) {
    const uint32_t before = (t - t0) * kPackTile;
    const uint32_t n = cnt - before < (uint32_t)kPackTile ? cnt - before : (uint32_t)kPackTile;
    PackTileInfo info;
    info.rec_first = rec0 + before;
    info.bit_pos = 0;
    info.sec_start_bit = (uint64_t)s;
    info.n_last = n | (t + 1 == t1 ? 0x80000000u : 0u);
    info.before = before;
    A.tile_info[t] = info;
  }
}

// Records of a tile -> registers -> LDS, so that the first record starts at stage[0].  The records start at
// any byte: unaligned dword loads (one instruction each on gfx950).  Fixed trip count, every load issued
// before the first use (a loop over a run-time count waits for each load in turn); the two halves are
// separate so that a tile's records can be requested while the previous tile is being packed.
constexpr int kPackStageIters = (kPackTile * 3 / 4 + kPackThreads - 1) / kPackThreads;
struct PackStagedLoads {
  uint32_t w[kPackStageIters];
};
JXLT_DI void pack_request_tile(const uint8_t* src, int n, int tid, PackStagedLoads* r) {
  const int nw = (3 * n + 3) >> 2;
#pragma unroll
  for (int k = 0; k < kPackStageIters; k++) {
    const int i = tid + k * kPackThreads;
    uint32_t v = 0;
    if (i < nw) __builtin_memcpy(&v, src + 4 * (size_t)i, 4);
    r->w[k] = v;
  }
}
JXLT_DI void pack_store_tile(const PackStagedLoads& r, int n, uint32_t* stage, int tid) {
  const int nw = (3 * n + 3) >> 2;
#pragma unroll
  for (int k = 0; k < kPackStageIters; k++) {
    const int i = tid + k * kPackThreads;
    if (i < nw) stage[i] = r.w[k];
  }
}

// The kPackPerThread consecutive records of thread `tid` (3 * kPackPerThread bytes = 12 dwords,
// dword aligned in the staged tile) with wide LDS reads; record j is the 24 bits at byte 3 * j.
struct PackThreadRecords {
  uint32_t w[kPackPerThread * 3 / 4 + 1];
};
JXLT_DI void pack_load_thread_records(const uint32_t* stage_tile, int tid, PackThreadRecords* out) {
  static_assert(kPackPerThread * 3 % 4 == 0, "whole dwords per thread");
  const uint32_t* p = stage_tile + tid * (kPackPerThread * 3 / 4);
#pragma unroll
  for (int q = 0; q < kPackPerThread * 3 / 4; q++) out->w[q] = p[q];
  out->w[kPackPerThread * 3 / 4] = 0;
}
JXLT_DI uint32_t pack_thread_record(const PackThreadRecords& r, int j) {  // ctx | value << 8
  const int byte = 3 * j;
  return __builtin_amdgcn_alignbyte(r.w[(byte >> 2) + 1], r.w[byte >> 2], (uint32_t)(byte & 3)) & 0xFFFFFFu;
}
JXLT_DI void pack_bits_of(uint32_t rec24, const uint32_t* table, uint32_t* nb, uint32_t* data) {
  const uint32_t ctx = rec24 & 0xFFu, value = rec24 >> 8;
  if (ctx >= 128) {
    *nb = ctx - 128;
    *data = value;
  } else {
    uint32_t sym, nbits, extra;
    hybrid_uint(value, &sym, &nbits, &extra);
    const uint32_t e = table[ctx * 64 + sym];
    const uint32_t depth = e >> 16;
    *nb = depth + nbits;
    *data = (e & 0xFFFFu) | (extra << depth);
  }
}

constexpr int kPackTilesPerGroup = 4;  // consecutive tiles per workgroup (amortises the table load)

__global__ void __launch_bounds__(kPackThreads) pack_tile_measure_kernel(const PackTileArgs A) {
  __shared__ uint8_t depth[64 * 64];
  __shared__ alignas(16) uint32_t stage[kPackTile * 3 / 4 + 4];
  __shared__ uint32_t total[kPackTilesPerGroup];
  const int tid = (int)threadIdx.x;
  const uint32_t ntiles_all = umin32((uint32_t)A.tile_base[A.nsec], A.tile_end);
  const uint32_t first = A.tile_first + blockIdx.x * kPackTilesPerGroup;
  if (first >= ntiles_all) return;
  for (int i = tid; i < 64 * 64; i += kPackThreads) depth[i] = (uint8_t)(A.code_table[i] >> 16);
  if (tid < kPackTilesPerGroup) total[tid] = 0;
  // The records of tile k + 1 are requested before tile k is summed: its descriptor one tile earlier still.
  PackTileInfo info = A.tile_info[first];
  PackTileInfo next_info = A.tile_info[first + 1 < ntiles_all ? first + 1 : first];
  PackStagedLoads loads;
  pack_request_tile(A.records + 3 * info.rec_first, (int)(info.n_last & 0x7FFFFFFFu), tid, &loads);
  for (int k = 0; k < kPackTilesPerGroup; k++) {
    const uint32_t tile = first + k;
    if (tile >= ntiles_all) break;
    __syncthreads();  // previous tile's stage consumed; tables loaded
    const int n = (int)(info.n_last & 0x7FFFFFFFu);
    pack_store_tile(loads, n, stage, tid);
    if (k + 1 < kPackTilesPerGroup && tile + 1 < ntiles_all) {
      info = next_info;
      next_info = A.tile_info[tile + 2 < ntiles_all ? tile + 2 : tile + 1];
      pack_request_tile(A.records + 3 * info.rec_first, (int)(info.n_last & 0x7FFFFFFFu), tid, &loads);
    }
    __syncthreads();
    PackThreadRecords recs;
    pack_load_thread_records(stage, tid, &recs);
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kPackPerThread; j++) {
      const int r = tid * kPackPerThread + j;
      if (r < n) {
        const uint32_t rec24 = pack_thread_record(recs, j);
        const uint32_t ctx = rec24 & 0xFFu, value = rec24 >> 8;
        if (ctx >= 128) {
          mine += ctx - 128;
        } else {
          uint32_t sym, nbits, extra;
          hybrid_uint(value, &sym, &nbits, &extra);
          mine += depth[ctx * 64 + sym] + nbits;
        }
      }
    }
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((tid & 63) == 0) atomicAdd(&total[k], mine);
  }
  __syncthreads();
  if (tid < kPackTilesPerGroup && first + tid < ntiles_all) A.tile_bits[first + tid] = total[tid];
}

__global__ void __launch_bounds__(256) pack_tile_offsets_kernel(const PackTileArgs A) {
  const int s = (int)(blockIdx.x * 256 + threadIdx.x);
  if (s >= A.nsec) return;
  const uint32_t t0 = (uint32_t)A.tile_base[s], t1 = (uint32_t)A.tile_base[s + 1];
  uint32_t off = 0;
  for (uint32_t t = t0; t < t1; t++) {
    A.tile_info[t].bit_pos = off;
    off += A.tile_bits[t];
  }
  A.sec_bits[s] = off;
  A.sec_bytes[s] = (off + 7) >> 3;
}

__global__ void __launch_bounds__(256) pack_tile_finalize_kernel(const PackTileArgs A) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint32_t)A.tile_base[A.nsec]) return;
  PackTileInfo info = A.tile_info[t];
  const uint32_t sec = (uint32_t)info.sec_start_bit;
  const uint64_t start = 8 * A.sec_byte_offset[sec];
  info.bit_pos += start;
  info.sec_start_bit = start;
  A.tile_info[t] = info;
  // The dwords in which two tiles (or two sections) meet are OR-ed into by both: zero them here, before the
  // writing pass -- the dword a tile starts in, and the dword a section ends in.
  uint32_t* outw = reinterpret_cast<uint32_t*>(A.out);
  outw[info.bit_pos >> 5] = 0u;
  if (info.n_last >> 31) {
    const uint64_t end = start + A.sec_bits[sec];
    if (end & 31u) outw[end >> 5] = 0u;
  }
}

__global__ void __launch_bounds__(kPackThreads) pack_tile_write_kernel(const PackTileArgs A) {
  __shared__ uint32_t table[64 * 64];
  __shared__ alignas(16) uint32_t stage[kPackTile * 3 / 4 + 4];
  __shared__ alignas(16) uint32_t window[kPackWindowWords];
  __shared__ uint32_t wave_sum[kPackThreads / 64];
  const int tid = (int)threadIdx.x;
  const uint32_t ntiles_all = umin32((uint32_t)A.tile_base[A.nsec], A.tile_end);
  const uint32_t first_tile = A.tile_first + blockIdx.x * kPackTilesPerGroup;
  if (first_tile >= ntiles_all) return;
  {  // (all eight loads of the code table in flight before the first LDS store)
    uint32_t tl[64 * 64 / kPackThreads];
#pragma unroll
    for (int q = 0; q < 64 * 64 / kPackThreads; q++) tl[q] = A.code_table[tid + q * kPackThreads];
#pragma unroll
    for (int q = 0; q < 64 * 64 / kPackThreads; q++) table[tid + q * kPackThreads] = tl[q];
  }
  uint32_t* outw = reinterpret_cast<uint32_t*>(A.out);
  // The records of tile kt + 1 are requested before tile kt is packed: its descriptor one tile earlier still.
  PackTileInfo info = A.tile_info[first_tile];
  PackTileInfo next_info = A.tile_info[first_tile + 1 < ntiles_all ? first_tile + 1 : first_tile];
  PackStagedLoads loads;
  pack_request_tile(A.records + 3 * info.rec_first, (int)(info.n_last & 0x7FFFFFFFu), tid, &loads);
  for (int kt = 0; kt < kPackTilesPerGroup; kt++) {
    const uint32_t tile = first_tile + kt;
    if (tile >= ntiles_all) break;
    __syncthreads();  // previous tile's window stored, its records consumed; table loaded
    const int n = (int)(info.n_last & 0x7FFFFFFFu);
    pack_store_tile(loads, n, stage, tid);
    for (int i = tid; i < kPackWindowWords; i += kPackThreads) window[i] = 0u;
    const uint64_t pos_bit = info.bit_pos;  // where this tile's bits start
    const uint32_t lead = (uint32_t)(pos_bit & 31u);
    const uint64_t word0 = pos_bit >> 5;
    if (kt + 1 < kPackTilesPerGroup && tile + 1 < ntiles_all) {
      info = next_info;
      next_info = A.tile_info[tile + 2 < ntiles_all ? tile + 2 : tile + 1];
      pack_request_tile(A.records + 3 * info.rec_first, (int)(info.n_last & 0x7FFFFFFFu), tid, &loads);
    }
    __syncthreads();  // stage complete, window clear
    // pass 1: bit length of this thread's records
    PackThreadRecords recs;
    pack_load_thread_records(stage, tid, &recs);
    uint32_t nb[kPackPerThread];
    uint32_t data[kPackPerThread];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < kPackPerThread; j++) {
      const int r = tid * kPackPerThread + j;
      nb[j] = 0;
      data[j] = 0;
      if (r < n) pack_bits_of(pack_thread_record(recs, j), table, &nb[j], &data[j]);
      mine += nb[j];
    }
    // exclusive prefix of `mine` over the workgroup: wave scan + per-wave totals
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    uint32_t wave_base = 0, tile_bits = 0;
#pragma unroll
    for (int w = 0; w < kPackThreads / 64; w++) {
      const uint32_t v = wave_sum[w];
      if (w < (tid >> 6)) wave_base += v;
      tile_bits += v;
    }
    // pass 2: OR the bits into the window
    {
      const uint32_t pos = lead + wave_base + incl - mine;
      uint32_t w = pos >> 5;
      uint32_t fill = pos & 31u;
      unsigned long long acc = 0;
#pragma unroll
      for (int j = 0; j < kPackPerThread; j++) {
        acc |= (unsigned long long)data[j] << fill;
        fill += nb[j];
        if (fill >= 32) {
          atomicOr(&window[w], (uint32_t)acc);
          acc >>= 32;
          fill -= 32;
          w++;
        }
      }
      if (fill) atomicOr(&window[w], (uint32_t)acc);
    }
    __syncthreads();
    // stores: the dwords the tile covers completely with plain stores; its first and its last dword, which it
    // may share with its neighbours (tiles of the same section, or the byte-aligned neighbour sections), are
    // OR-ed into memory that pack_tile_finalize_kernel zeroed
    const uint32_t end_bits = lead + tile_bits;
    const uint32_t nwords = (end_bits + 31) >> 5;  // dwords the tile touches
    for (uint32_t i = tid; i < nwords; i += kPackThreads) {
      const uint32_t v = window[i];
      if (i == 0 || (i + 1 == nwords && (end_bits & 31u) != 0)) {
        if (v) atomicOr(&outw[word0 + i], v);
      } else {
        outw[word0 + i] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// DC-group sections as raw records (enc_frame.cc:287-424, 536-570):
//   [esc 6 bits][DC tokens Y,X,B][esc nb_bits][esc 4 bits][ytox][ytob]
//   [strategy per first block][quant field per first block][EPF per block]
// Contexts are the reference's DC context ids (identity pre-clustering).
// dc_elementwise_kernel: every token whose position needs no scan.
// dc_chain_kernel: the two per-first-block token runs ("left" = previous first block).
// ---------------------------------------------------------------------------
struct DcArgs {
  FrameGeom g;
  const DeviceTables* tab;
  const int16_t* quant_dc[3];
  const uint8_t* raw_quant;
  const uint8_t* strategy;
  const int8_t* ytox;
  const int8_t* ytob;
  const uint32_t* dc_nac;         // [ndc] first blocks per DC group
  const uint64_t* dc_rec_offset;  // [ndc] start of each DC group's records (fixed stride)
  uint8_t* records;
  uint32_t* dc_count;             // [ndc] records per DC group
  uint32_t* histogram;            // [64 * 64]
  uint32_t* chain_summary;        // [ndc * kDcChainChunks]: first blocks in the chunk | last one's (code << 8 | qf - 1) << 16
  int dcg_first;                  // the launch covers DC groups dcg_first .. (one row of DC groups at a time)
};

struct DcGeom {
  int bx0, by0, nbx, nby, nb;      // block rect of the DC group
  int tx0, ty0, ntx, nty, nt;      // tile rect
  uint32_t pos_dc, pos_esc, pos_cmap, pos_strategy, pos_qf, pos_epf, total;
};

JXLT_DI DcGeom dc_geom(const FrameGeom& g, int dcg, uint32_t nac) {
  DcGeom d;
  const int xdc = (g.xsize + 2047) / 2048;
  const int gx = dcg % xdc, gy = dcg / xdc;
  d.bx0 = gx * 256;
  d.by0 = gy * 256;
  d.nbx = imin(256, g.xsize_blocks - d.bx0);
  d.nby = imin(256, g.ysize_blocks - d.by0);
  d.nb = d.nbx * d.nby;
  d.tx0 = gx * 32;
  d.ty0 = gy * 32;
  d.ntx = (d.nbx * 8 + 63) / 64;
  d.nty = (d.nby * 8 + 63) / 64;
  d.nt = d.ntx * d.nty;
  d.pos_dc = 1;
  d.pos_esc = 1 + 3 * (uint32_t)d.nb;
  d.pos_cmap = d.pos_esc + (d.nb > 1 ? 2 : 1);
  d.pos_strategy = d.pos_cmap + 2 * (uint32_t)d.nt;
  d.pos_qf = d.pos_strategy + nac;
  d.pos_epf = d.pos_qf + nac;
  d.total = d.pos_epf + (uint32_t)d.nb;
  return d;
}

JXLT_DI int clamped_gradient(int n, int w, int l) {  // enc_frame.cc:158-176
  const int m = n < w ? n : w, M = n < w ? w : n;
  const int grad = (int)((uint32_t)n + (uint32_t)w - (uint32_t)l);
  const int grad_clamp_M = (l < m) ? M : grad;
  return (l > M) ? m : grad_clamp_M;
}

JXLT_DI void put_record(uint8_t* rec, uint32_t pos, uint32_t ctx, uint32_t value, uint32_t* hist) {
  uint8_t* o = rec + 3 * (size_t)pos;
  o[0] = (uint8_t)ctx;
  o[1] = (uint8_t)(value & 0xFF);
  o[2] = (uint8_t)((value >> 8) & 0xFF);
  if (ctx < 128) {
    uint32_t sym, nb, eb;
    hybrid_uint(value & 0xFFFFu, &sym, &nb, &eb);
    atomicAdd(&hist[ctx * 64 + sym], 1u);
  }
}

constexpr int kDcParts = 32;  // workgroups per DC group in dc_elementwise_kernel

__global__ void __launch_bounds__(256) dc_elementwise_kernel(const DcArgs A) {
  __shared__ uint32_t hist[64 * 64];
  const int tid = (int)threadIdx.x;
  const int dcg = A.dcg_first + (int)blockIdx.x / kDcParts, part = (int)blockIdx.x % kDcParts;
  for (int i = tid; i < 64 * 64; i += 256) hist[i] = 0;
  __syncthreads();
  const uint32_t nac = A.dc_nac[dcg];
  const DcGeom d = dc_geom(A.g, dcg, nac);
  uint8_t* rec = A.records + 3 * A.dc_rec_offset[dcg];
  const size_t bstride = (size_t)A.g.xsize_blocks;
  if (part == 0 && tid == 0) {
    put_record(rec, 0, 128 + 6, 12, hist);  // extra_dc_precision = 0, global tree / default wp
    uint32_t p = d.pos_esc;
    if (d.nb > 1) put_record(rec, p++, 128 + (uint32_t)ceil_log2_nonzero((uint32_t)d.nb), nac - 1, hist);
    put_record(rec, p, 128 + 4, 3, hist);
    A.dc_count[dcg] = d.total;
  }
  // DC tokens (WriteDCTokens, enc_frame.cc:287-316): the part's share of the block rows, a thread per block
  // column (a DC group is at most 256 blocks wide: no index divisions), the three channels in turn.
  {
    const int rows_per = (d.nby + kDcParts - 1) / kDcParts;
    const int y0 = part * rows_per, y1 = imin(d.nby, y0 + rows_per);
    const int x = tid;
    if (x < d.nbx) {
      for (int y = y0; y < y1; y++) {
#pragma unroll
        for (int ci = 0; ci < 3; ci++) {
          const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
          const int16_t* q = A.quant_dc[c] + (size_t)(d.by0 + y) * bstride + d.bx0 + x;
          const int left = x ? q[-1] : y ? q[-(ptrdiff_t)bstride] : 0;
          const int top = y ? q[-(ptrdiff_t)bstride] : left;
          const int topleft = (x && y) ? q[-(ptrdiff_t)bstride - 1] : left;
          const int guess = clamped_gradient(top, left, topleft);
          int gp = 512 + top + left - topleft;
          gp = gp < 0 ? 0 : gp > 1023 ? 1023 : gp;
          const int residual = (int)q[0] - guess;
          put_record(rec, d.pos_dc + (uint32_t)(ci * d.nb + y * d.nbx + x), A.tab->gradient_lut[gp],
                     pack_signed(residual), hist);
        }
      }
    }
  }
  // YtoX / YtoB tokens (enc_frame.cc:339-362)
  const int perc = (2 * d.nt + kDcParts - 1) / kDcParts;
  for (int i = part * perc + tid; i < imin(2 * d.nt, (part + 1) * perc); i += 256) {
    const int c = i / d.nt, r = i % d.nt;
    const int y = r / d.ntx, x = r % d.ntx;
    const int8_t* m = (c == 0 ? A.ytox : A.ytob) + (size_t)(d.ty0 + y) * A.g.xsize_tiles + d.tx0 + x;
    const ptrdiff_t ts = A.g.xsize_tiles;
    const int left = x ? m[-1] : y ? m[-ts] : 0;
    const int top = y ? m[-ts] : left;
    const int topleft = (x && y) ? m[-ts - 1] : left;
    const int residual = (int)m[0] - clamped_gradient(top, left, topleft);
    put_record(rec, d.pos_cmap + (uint32_t)i, 2u - (uint32_t)c, pack_signed(residual), hist);
  }
  // EPF tokens (enc_frame.cc:410-423)
  const int pere = (d.nb + kDcParts - 1) / kDcParts;
  for (int i = part * pere + tid; i < imin(d.nb, (part + 1) * pere); i += 256)
    put_record(rec, d.pos_epf + (uint32_t)i, 0, pack_signed(4), hist);
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256)
    if (hist[i]) atomicAdd(&A.histogram[i], hist[i]);
}

constexpr int kDcChainThreads = 1024;

// The two per-first-block token runs need, for every first block, its rank among the DC group's
// first blocks and the previous first block's (strategy code, quant field).  One workgroup per
// chunk of kDcChainThreads blocks (64 chunks per full DC group, all of them in parallel):
// dc_chain_summary_kernel records each chunk's first-block count and its last first block's
// values; dc_chain_kernel derives a chunk's carry from the summaries of its predecessors.
constexpr int kDcChainChunks = 65536 / kDcChainThreads;  // per DC group (256 x 256 blocks)

struct DcChunkBlock {
  bool first;
  int code, qfm1;
};
JXLT_DI DcChunkBlock dc_chunk_block(const DcArgs& A, const DcGeom& d, int i) {
  DcChunkBlock b = {false, 0, 0};
  if (i < d.nb) {
    const size_t pos = (size_t)(d.by0 + i / d.nbx) * (size_t)A.g.xsize_blocks + d.bx0 + i % d.nbx;
    const uint8_t a = A.strategy[pos];
    b.first = (a & 1) != 0;
    b.code = (a >> 1) == 0 ? 0 : (a >> 1) == 1 ? 6 : 7;
    b.qfm1 = (int)A.raw_quant[pos] - 1;
  }
  return b;
}

__global__ void __launch_bounds__(kDcChainThreads) dc_chain_summary_kernel(const DcArgs A) {
  __shared__ uint32_t count;
  __shared__ int last_idx;
  __shared__ uint32_t last_val;
  const int tid = (int)threadIdx.x;
  const int dcg = A.dcg_first + (int)blockIdx.x / kDcChainChunks, chunk = (int)blockIdx.x % kDcChainChunks;
  const DcGeom d = dc_geom(A.g, dcg, 0);
  if (tid == 0) {
    count = 0;
    last_idx = -1;
    last_val = 0;
  }
  __syncthreads();
  const DcChunkBlock b = dc_chunk_block(A, d, chunk * kDcChainThreads + tid);
  const unsigned long long m = __ballot(b.first);
  if ((tid & 63) == 0 && m != 0) {
    atomicAdd(&count, (uint32_t)__popcll(m));
    atomicMax(&last_idx, (tid & ~63) + 63 - __clzll((long long)m));
  }
  __syncthreads();
  if (b.first && tid == last_idx) last_val = (uint32_t)((b.code << 8) | b.qfm1);
  __syncthreads();
  if (tid == 0) A.chain_summary[dcg * kDcChainChunks + chunk] = count | (last_val << 16);
}

__global__ void __launch_bounds__(kDcChainThreads) dc_chain_kernel(const DcArgs A) {
  __shared__ uint32_t hist[16 * 64];  // the two runs only use contexts 3..10
  __shared__ uint32_t wsum[kDcChainThreads / 64];
  __shared__ uint16_t compact[kDcChainThreads + 1];  // (code << 8) | (qf - 1) of the chunk's first blocks
  __shared__ uint32_t carry_rank;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int dcg = A.dcg_first + (int)blockIdx.x / kDcChainChunks, chunk = (int)blockIdx.x % kDcChainChunks;
  const uint32_t nac = A.dc_nac[dcg];
  const DcGeom d = dc_geom(A.g, dcg, nac);
  if (chunk * kDcChainThreads >= d.nb) return;  // (partial DC groups have fewer chunks)
  for (int i = tid; i < 16 * 64; i += kDcChainThreads) hist[i] = 0;
  uint8_t* rec = A.records + 3 * A.dc_rec_offset[dcg];
  const size_t bstride = (size_t)A.g.xsize_blocks;
  // "left" before the first first-block: 0 for the strategy run, StrategyCode(acs(0,0)) for
  // the quant-field run (sic, enc_frame.cc:386)
  const uint8_t a00 = A.strategy[(size_t)d.by0 * bstride + d.bx0];
  const int code00 = (a00 >> 1) == 0 ? 0 : (a00 >> 1) == 1 ? 6 : 7;
  if (tid < 64) {
    // carry from the preceding chunks: their first-block counts, and the values of the last
    // first block before this chunk (lane c looks at chunk c; 64 chunks = one wave)
    const uint32_t sm = (tid < chunk) ? A.chain_summary[dcg * kDcChainChunks + tid] : 0u;
    uint32_t cnt = sm & 0xFFFFu;
    const unsigned long long nonempty = __ballot(cnt != 0);
    for (int dd = 32; dd >= 1; dd >>= 1) cnt += __shfl_xor(cnt, dd);
    const int src = nonempty ? 63 - __clzll((long long)nonempty) : 0;
    const uint32_t prev = __shfl(sm >> 16, src);
    if (tid == 0) {
      carry_rank = cnt;
      compact[0] = nonempty ? (uint16_t)prev : (uint16_t)((0 << 8) | code00);  // predecessor of the chunk's first entry
    }
  }
  __syncthreads();
  {
    const DcChunkBlock b = dc_chunk_block(A, d, chunk * kDcChainThreads + tid);
    const bool first = b.first;
    const int code = b.code, qfm1 = b.qfm1;
    // exclusive rank of first blocks inside the chunk
    const unsigned long long m = __ballot(first);
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < kDcChainThreads / 64; w++)
      if (w < wave) wbase += wsum[w];
    const uint32_t r = wbase + in_wave;  // rank within chunk
    if (first) compact[1 + r] = (uint16_t)((code << 8) | qfm1);
    __syncthreads();
    if (first) {
      const uint16_t prev = compact[r];  // previous first block (or the carried one)
      const uint32_t grank = carry_rank + r;
      // strategy token (enc_frame.cc:364-383): left = previous code (0 for the very first)
      const int left_s = (grank == 0) ? 0 : (prev >> 8);
      const uint32_t ctx_s = left_s > 11 ? 7 : left_s > 5 ? 8 : left_s > 3 ? 9 : 10;
      put_record(rec, d.pos_strategy + grank, ctx_s, pack_signed(code), hist);
      // quant-field token (:384-408): left = previous (qf-1), initially code of block (0,0)
      const int left_q = (grank == 0) ? code00 : (prev & 0xFF);
      const uint32_t ctx_q = left_q > 11 ? 3 : left_q > 5 ? 4 : left_q > 3 ? 5 : 6;
      put_record(rec, d.pos_qf + grank, ctx_q, pack_signed(qfm1 - left_q), hist);
    }
  }
  __syncthreads();
  for (int i = tid; i < 16 * 64; i += kDcChainThreads)
    if (hist[i]) atomicAdd(&A.histogram[i], hist[i]);
}

}  // namespace jxlt_dev

#endif  // JXLT_DEVICE_H_
