/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  See jxl_tiny_oracle.h.
 *
 * CPU restatement of the libjxl-tiny per-group hot path, written as plain
 * scalar C that spells out the canonical 8-lane / fused-MulAdd arithmetic model.
 * PARITY UNPINNED (no reference golden vectors exist; reference not buildable
 * here) -- see the header.
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -mfma (see oracle/Makefile).
 * -ffp-contract=off is REQUIRED: only fma32()/nfma32() may fuse.
 *
 * All "ref:" citations are relative to /root/reference/encoder/.
 */
#include "jxl_tiny_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "orc_tables.h"

/* ------------------------------------------------------------------------- */
/* Canonical arithmetic primitives                                           */
/* ------------------------------------------------------------------------- */

#define LANES 8 /* HWY_FULL(float) in the canonical model */

/* ORC_FMA=0 builds the "unfused" variant of the model (hwy MulAdd = mul, add),
 * used only by tests/test_oracle_known_answers.py to compare against the
 * size-only known answers of SURVEY.md Appendix C. */
#ifndef ORC_FMA
#define ORC_FMA 1
#endif
static inline float fma32(float a, float b, float c) { /* hwy MulAdd */
#if ORC_FMA
  return __builtin_fmaf(a, b, c);
#else
  return a * b + c;
#endif
}
static inline float nfma32(float a, float b, float c) { /* hwy NegMulAdd: c-a*b */
#if ORC_FMA
  return __builtin_fmaf(-a, b, c);
#else
  return c - a * b;
#endif
}
static inline uint32_t f2u(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
/* hwy ZeroIfNegative on x86 tests the sign bit. */
static inline float zero_if_negative(float v) {
  return (f2u(v) & 0x80000000u) ? 0.0f : v;
}
/* hwy SumOfLanes for 8 lanes: (i)+(i+4), then +2, then +1. */
static inline float sum_of_lanes8(const float* t) {
  float a0 = t[0] + t[4], a1 = t[1] + t[5], a2 = t[2] + t[6], a3 = t[3] + t[7];
  float b0 = a0 + a2, b1 = a1 + a3;
  return b0 + b1;
}
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline float minf(float a, float b) { return a < b ? a : b; }
static inline size_t div_ceil(size_t a, size_t b) { return (a + b - 1) / b; }

static inline int floor_log2_nonzero(uint64_t x) { return 63 - __builtin_clzll(x); }
/* ref: base/bits.h:122-132 */
static inline int ceil_log2_nonzero(uint64_t x) {
  int fl = floor_log2_nonzero(x);
  return (x & (x - 1)) == 0 ? fl : fl + 1;
}
/* ref: common.h:54-58 */
static inline uint32_t pack_signed(int32_t v) {
  return ((uint32_t)v << 1) ^ (((uint32_t)(~v) >> 31) - 1);
}

/* ------------------------------------------------------------------------- */
/* fast_math-inl.h                                                           */
/* ------------------------------------------------------------------------- */

/* ref: fast_math-inl.h:113-133 (FastLog2f) + :74-108 (EvalRationalPolynomial) */
float orc_fast_log2f(float x) {
  const float p0 = -1.8503833400518310E-06f, p1 = 1.4287160470083755E+00f,
              p2 = 7.4245873327820566E-01f;
  const float q0 = 9.9032814277590719E-01f, q1 = 1.0096718572241148E+00f,
              q2 = 1.7409343003366853E-01f;
  int32_t x_bits = (int32_t)f2u(x);
  int32_t exp_bits = x_bits - 0x3f2aaaab;
  int32_t exp_shifted = exp_bits >> 23; /* arithmetic */
  float mantissa = u2f((uint32_t)(x_bits - (int32_t)((uint32_t)exp_shifted << 23)));
  float exp_val = (float)exp_shifted;
  float t = mantissa - 1.0f;
  float yp = p2, yq = q2;
  yp = fma32(yp, t, p1);
  yq = fma32(yq, t, q1);
  yp = fma32(yp, t, p0);
  yq = fma32(yq, t, q0);
  return yp / yq + exp_val;
}

/* ref: fast_math-inl.h:137-151 (FastPow2f); constants are double literals
 * narrowed by Set(df, .) */
float orc_fast_pow2f(float x) {
  float floorx = floorf(x);
  float e = u2f((uint32_t)(((int32_t)floorx + 127)) << 23);
  float frac = x - floorx;
  float num = frac + (float)1.01749063e+01;
  num = fma32(num, frac, (float)4.88687798e+01);
  num = fma32(num, frac, (float)9.85506591e+01);
  num = num * e;
  float den = fma32(frac, (float)2.10242958e-01, (float)-2.22328856e-02);
  den = fma32(den, frac, (float)-1.94414990e+01);
  den = fma32(den, frac, (float)9.85506633e+01);
  return num / den;
}

/* ref: fast_math-inl.h:178-213 (CubeRootAndAdd) */
static inline float cube_root_and_add(float x, float add) {
  const float k1_3 = 1.0f / 3, k4_3 = 4.0f / 3;
  float xa_3 = k1_3 * x;
  int32_t m1 = (int32_t)f2u(x);
  int32_t m2 =
      (m1 == 0) ? 0
                : (int32_t)(0x54800000u - (uint32_t)(m1 >> 23) * 0x002AAAAAu);
  float r = u2f((uint32_t)m2);
  for (int i = 0; i < 3; i++) {
    float r2 = r * r;
    r = nfma32(xa_3, r2 * r2, k4_3 * r);
  }
  float r2 = r * r;
  r = fma32(k1_3, nfma32(x, r2 * r2, r), r);
  r2 = r * r;
  r = fma32(r2, x, add);
  return r;
}

/* ------------------------------------------------------------------------- */
/* enc_xyb.cc                                                                */
/* ------------------------------------------------------------------------- */

/* ref: enc_xyb.cc:30-81 (ToXYB), in place on three rows of n samples */
void orc_to_xyb(float* row0, float* row1, float* row2, size_t n) {
  const float kM02 = 0.078f, kM00 = 0.30f, kM01 = 1.0f - kM02 - kM00;
  const float kM12 = 0.078f, kM10 = 0.23f, kM11 = 1.0f - kM12 - kM10;
  const float kM20 = 0.24342268924547819f, kM21 = 0.20476744424496821f,
              kM22 = 1.0f - kM20 - kM21;
  const float bias = 0.0037930732552754493f;
  const float neg_bias_cbrt = -0.15595420054f;
  for (size_t x = 0; x < n; ++x) {
    float r = row0[x], g = row1[x], b = row2[x];
    float mixed0 = fma32(kM00, r, fma32(kM01, g, fma32(kM02, b, bias)));
    float mixed1 = fma32(kM10, r, fma32(kM11, g, fma32(kM12, b, bias)));
    float mixed2 = fma32(kM20, r, fma32(kM21, g, fma32(kM22, b, bias)));
    float tm0 = cube_root_and_add(zero_if_negative(mixed0), neg_bias_cbrt);
    float tm1 = cube_root_and_add(zero_if_negative(mixed1), neg_bias_cbrt);
    float tm2 = cube_root_and_add(zero_if_negative(mixed2), neg_bias_cbrt);
    row0[x] = 0.5f * (tm0 - tm1);
    row1[x] = 0.5f * (tm0 + tm1);
    row2[x] = tm2;
  }
}

/* ------------------------------------------------------------------------- */
/* enc_transforms-inl.h + dct_scales.h                                       */
/* ------------------------------------------------------------------------- */

static const float kSqrt2 = 1.41421356237f; /* ref: dct_scales.h:16 */
/* ref: dct_scales.h:82-107 (double literals narrowed to float) */
static const float kWc4[2] = {(float)0.541196100146197, (float)1.3065629648763764};
static const float kWc8[4] = {(float)0.5097955791041592, (float)0.6013448869350453,
                              (float)0.8999762231364156, (float)2.5629154477415055};
static const float kWc16[8] = {
    (float)0.5024192861881557, (float)0.5224986149396889, (float)0.5669440348163577,
    (float)0.6468217833599901, (float)0.7881546234512502, (float)1.060677685990347,
    (float)1.7224470982383342, (float)5.101148618689155};

/* ref: enc_transforms-inl.h:394-425 (DCT1DImpl) with CoeffBundle ops :292-392.
 * One column of N samples (the SIMD lanes are independent columns). */
static void dct1d(float* mem, int n) {
  if (n == 1) return;
  if (n == 2) {
    float a = mem[0], b = mem[1];
    mem[0] = a + b;
    mem[1] = a - b;
    return;
  }
  float tmp[16];
  const int h = n / 2;
  const float* wc = (n == 4) ? kWc4 : (n == 8) ? kWc8 : kWc16;
  for (int i = 0; i < h; i++) tmp[i] = mem[i] + mem[n - 1 - i]; /* AddReverse */
  dct1d(tmp, h);
  for (int i = 0; i < h; i++) tmp[h + i] = mem[i] - mem[n - 1 - i]; /* SubReverse */
  for (int i = 0; i < h; i++) tmp[h + i] = tmp[h + i] * wc[i];      /* Multiply */
  dct1d(tmp + h, h);
  /* B<N/2> on the odd half (:312-322) */
  tmp[h] = fma32(tmp[h], kSqrt2, tmp[h + 1]);
  for (int i = 1; i + 1 < h; i++) tmp[h + i] = tmp[h + i] + tmp[h + i + 1];
  /* InverseEvenOdd */
  for (int i = 0; i < h; i++) {
    mem[2 * i] = tmp[i];
    mem[2 * i + 1] = tmp[h + i];
  }
}

/* ref: ComputeScaledDCT<8,8> (:527-546): out[h*8+v] */
void orc_dct8x8(const float* px, size_t stride, float* out) {
  float a[8][8], col[8];
  for (int x = 0; x < 8; x++) {
    for (int y = 0; y < 8; y++) col[y] = px[y * stride + x];
    dct1d(col, 8);
    for (int v = 0; v < 8; v++) a[v][x] = (1.0f / 8) * col[v];
  }
  for (int v = 0; v < 8; v++) {
    for (int x = 0; x < 8; x++) col[x] = a[v][x];
    dct1d(col, 8);
    for (int hh = 0; hh < 8; hh++) out[hh * 8 + v] = (1.0f / 8) * col[hh];
  }
}

/* ref: ComputeScaledDCT<16,8>: 16 rows x 8 cols; out[h*16+v], h<8, v<16 */
void orc_dct16x8(const float* px, size_t stride, float* out) {
  float a[16][8], col[16];
  for (int x = 0; x < 8; x++) {
    for (int y = 0; y < 16; y++) col[y] = px[y * stride + x];
    dct1d(col, 16);
    for (int v = 0; v < 16; v++) a[v][x] = (1.0f / 16) * col[v];
  }
  for (int v = 0; v < 16; v++) {
    for (int x = 0; x < 8; x++) col[x] = a[v][x];
    dct1d(col, 8);
    for (int hh = 0; hh < 8; hh++) out[hh * 16 + v] = (1.0f / 8) * col[hh];
  }
}

/* ref: ComputeScaledDCT<8,16>: 8 rows x 16 cols; out[v*16+h], v<8, h<16 */
void orc_dct8x16(const float* px, size_t stride, float* out) {
  float a[8][16], col[16];
  for (int x = 0; x < 16; x++) {
    for (int y = 0; y < 8; y++) col[y] = px[y * stride + x];
    dct1d(col, 8);
    for (int v = 0; v < 8; v++) a[v][x] = (1.0f / 8) * col[v];
  }
  for (int v = 0; v < 8; v++) {
    for (int x = 0; x < 16; x++) col[x] = a[v][x];
    dct1d(col, 16);
    for (int hh = 0; hh < 16; hh++) out[v * 16 + hh] = (1.0f / 16) * col[hh];
  }
}

enum { STRAT_DCT = 0, STRAT_DCT16X8 = 1, STRAT_DCT8X16 = 2 };
static const int kCoveredX[3] = {1, 1, 2}; /* ref: ac_strategy.h:81-93 */
static const int kCoveredY[3] = {1, 2, 1};
static const uint8_t kStrategyCode[3] = {0, 6, 7}; /* ref: ac_strategy.h:59-62 */

/* ref: enc_transforms-inl.h:602-627 (TransformFromPixels) */
static void transform_from_pixels(int strategy, const float* px, size_t stride,
                                  float* coeffs) {
  if (strategy == STRAT_DCT16X8) orc_dct16x8(px, stride, coeffs);
  else if (strategy == STRAT_DCT8X16) orc_dct8x16(px, stride, coeffs);
  else orc_dct8x8(px, stride, coeffs);
}

/* ref: enc_transforms-inl.h:629-652 (DCFromLowestFrequencies) via
 * ReinterpretingIDCT (:572-600) with DCTResampleScales<16,2> (dct_scales.h:53-58) */
static void dc_from_lowest_frequencies(int strategy, const float* block, float* dc,
                                       size_t dc_stride) {
  const float kScale1 = (float)0.901764195028874394;
  if (strategy == STRAT_DCT) {
    dc[0] = block[0];
  } else {
    float b0 = block[0] * 1.0f * 1.0f;
    float b1 = block[1] * 1.0f * kScale1;
    if (strategy == STRAT_DCT16X8) {
      dc[0] = b0 + b1;
      dc[dc_stride] = b0 - b1;
    } else {
      dc[0] = b0 + b1;
      dc[1] = b0 - b1;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* quant_weights.cc                                                          */
/* ------------------------------------------------------------------------- */

typedef struct {
  float w[576];   /* dequant weights */
  float inv[576]; /* 1/w computed in double, LLF entries zeroed */
} dequant_matrices;

/* ref: quant_weights.cc:140-157 */
static void dequant_matrices_init(dequant_matrices* m) {
  for (int i = 0; i < 576; i++) {
    m->w[i] = u2f(ORC_kQuantWeightBits[i]);
    m->inv[i] = (float)(1.0 / m->w[i]);
  }
  for (int n = 0; n < 9; n++)
    for (int b = 0; b < ORC_kQuantTableLLF[n]; b++) m->inv[ORC_kQuantTableOffset[n] + b] = 0.0f;
}
static inline const float* dq_matrix(const dequant_matrices* m, int kind, int c) {
  return &m->w[ORC_kQuantTableOffset[kind * 3 + c]];
}
static inline const float* dq_inv_matrix(const dequant_matrices* m, int kind, int c) {
  return &m->inv[ORC_kQuantTableOffset[kind * 3 + c]];
}

/* ------------------------------------------------------------------------- */
/* enc_frame.cc: DistanceParams                                              */
/* ------------------------------------------------------------------------- */

static inline float clampf(float v, float lo, float hi) {
  return v < lo ? lo : v > hi ? hi : v;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* ref: enc_frame.cc:95-102 */
static float quant_dc_for_distance(float distance) {
  const float kDcQuantPow = 0.57f, kDcQuant = 1.12f, kDcMul = (float)2.9;
  float effective_dist = kDcMul * powf(distance / kDcMul, kDcQuantPow);
  effective_dist = clampf(effective_dist, 0.5f * distance, distance);
  return minf(kDcQuant / effective_dist, 50.f);
}

/* ref: enc_frame.cc:115-156 */
void orc_compute_distance_params(float distance, orc_distance_params* p) {
  p->distance = distance;
  const int kGlobalScaleDenom = 1 << 16, kGlobalScaleNumerator = 4096;
  const float kAcQuant = 0.8f, kQuantFieldTarget = 5;
  float quant_dc = quant_dc_for_distance(distance);
  float scale = kGlobalScaleDenom * kAcQuant / (distance * kQuantFieldTarget);
  scale = clampf(scale, 1.0f, 1.0f * (1 << 15));
  int scaled_quant_dc = (int)(quant_dc * kGlobalScaleNumerator * 1.6);
  p->global_scale = clampi((int)scale, 1, scaled_quant_dc);
  p->scale = p->global_scale * (1.0f / kGlobalScaleDenom);
  p->inv_scale = 1.0f / p->scale;
  p->quant_dc = (int)(quant_dc / p->scale + 0.5f);
  p->quant_dc = clampi(p->quant_dc, 1, 1 << 16);
  p->scale_dc = p->quant_dc * p->scale;
  p->x_qm_scale = 2;
  if (distance > 1.25f) p->x_qm_scale++;
  if (distance > 9.0f) p->x_qm_scale++;
  if (distance < 0.299f) p->x_qm_scale++;
  const float kEpf[3] = {(float)0.7, (float)1.5, (float)4.0};
  p->epf_iters = 0;
  for (int i = 0; i < 3; i++)
    if (distance >= kEpf[i]) p->epf_iters++;
}

/* ------------------------------------------------------------------------- */
/* Stripe scratch image (ref: enc_frame.cc:703 "Image3F stripe(256, 64)")     */
/* ------------------------------------------------------------------------- */

#define STRIPE_W 256
#define STRIPE_H 64
#define STRIPE_PITCH (STRIPE_W + 16)

typedef struct {
  size_t xsize, ysize; /* padded to x8 */
  float px[3][STRIPE_H][STRIPE_PITCH];
} stripe_t;

static inline const float* srow(const stripe_t* s, int c, size_t y) { return s->px[c][y]; }

/* ref: enc_frame.cc:597-617 (CopyAndPadImage) */
static void copy_and_pad(const float* const planes[3], size_t stride, size_t x0,
                         size_t y0, size_t w, size_t h, stripe_t* s) {
  size_t wp = div_ceil(w, 8) * 8, hp = div_ceil(h, 8) * 8;
  s->xsize = wp;
  s->ysize = hp;
  for (int c = 0; c < 3; c++) {
    for (size_t y = 0; y < h; y++) {
      float* dst = s->px[c][y];
      memcpy(dst, planes[c] + (y0 + y) * stride + x0, w * sizeof(float));
      float last = dst[w - 1];
      for (size_t x = w; x < wp; x++) dst[x] = last;
      for (size_t x = wp; x < STRIPE_PITCH; x++) dst[x] = 0.0f; /* never used unmasked */
    }
    for (size_t y = h; y < hp; y++) memcpy(s->px[c][y], s->px[c][h - 1], STRIPE_PITCH * sizeof(float));
  }
}

/* ------------------------------------------------------------------------- */
/* enc_adaptive_quantization.cc                                              */
/* ------------------------------------------------------------------------- */

static const float kSGmul = 226.0480446705883f;            /* :78-83 */
static const float kSGmul2 = 1.0f / 73.377132366608819f;
static const float kLog2c = 0.693147181f;
#define kSGRetMul (kSGmul2 * 18.6580932135f * kLog2c)
static const float kSGVOffset = 7.14672470003f;

/* ref: :85-104 (RatioOfDerivativesOfCubicRootToSimpleGamma) */
static inline float ratio_of_derivatives(float v, int invert) {
  const float kEpsilon = (float)1e-2;
  v = zero_if_negative(v);
  const float kNumMul = kSGRetMul * 3 * kSGmul;
  const float kVOffset = kSGVOffset * kLog2c + kEpsilon;
  const float kDenMul = kLog2c * kSGmul;
  float v2 = v * v;
  float num = fma32(kNumMul, v2, kEpsilon);
  float den = fma32(kDenMul * v, v2, kVOffset);
  return invert ? num / den : den / num;
}

/* ref: :287-294 (MaskingSqrt) */
static inline float masking_sqrt(float v) {
  const float kLogOffset = 26.481471032459346f;
  const float kMul = 211.50759899638012f;
  const float mul_v = (float)(kMul * 1e8);
  return 0.25f * sqrtf(fma32(v, sqrtf(mul_v), kLogOffset));
}

/* ref: :52-75 (ComputeMask) */
static inline float compute_mask(float out_val) {
  const float kBase = -0.74174993f, kMul4 = 3.2353257320940401f,
              kMul2 = 12.906028311180409f, kOffset2 = 305.04035728311436f,
              kMul3 = 5.0220313103171232f, kOffset3 = 2.1925739705298404f,
              kMul0 = 0.74760422233706747f;
  const float kOffset4 = 0.25f * kOffset3;
  float v1 = maxf(out_val * kMul0, 1e-3f);
  float v2 = 1.0f / (v1 + kOffset2);
  float v3 = 1.0f / fma32(v1, v1, kOffset3);
  float v4 = 1.0f / fma32(v1, v1, kOffset4);
  return kBase + fma32(kMul4, v4, fma32(kMul2, v2, kMul3 * v3));
}

/* ref: :209-247 (HfModulation), non-scalar branch (8 lanes incl. masked lane 7) */
static float hf_modulation(const stripe_t* s, size_t x, size_t y, float out_val) {
  float sum[8] = {0};
  for (int dy = 0; dy < 8; dy++) {
    const float* row = srow(s, 1, y + dy) + x;
    const float* next = (dy == 7) ? row : srow(s, 1, y + dy + 1) + x;
    for (int l = 0; l < 8; l++) {
      float p = row[l];
      float right = (l < 7) ? fabsf(p - row[l + 1]) : 0.0f; /* kMaskRight */
      sum[l] = sum[l] + right;
      sum[l] = sum[l] + fabsf(p - next[l]);
    }
  }
  float total = sum_of_lanes8(sum);
  return fma32(total, -2.0052193233688884f / 112, out_val);
}

/* ref: :146-207 (ColorModulation); butteraugli_target is a double parameter */
static float color_modulation(const stripe_t* s, size_t x, size_t y,
                              double butteraugli_target, float out_val) {
  const float kStrengthMul = (float)2.177823400325309;
  const float kRedRampStart = (float)0.0073200141118951231;
  const float kRedRampLength = (float)0.019421555948474039;
  const float kBlueRampLength = (float)0.086890611400405895;
  const float kBlueRampStart = (float)0.26973418507870539;
  const float strength = (float)(kStrengthMul * (1.0f - 0.25f * butteraugli_target));
  if (strength < 0) return out_val;
  const float red_strength = strength * 5.992297772961519f;
  const float blue_strength = strength;
  {
    const float offset = strength * -0.009174542291185913f;
    out_val = out_val + offset;
  }
  float blue[8] = {0}, red[8] = {0};
  for (int dy = 0; dy < 8; dy++) {
    const float* rx = srow(s, 0, y + dy) + x;
    const float* ry = srow(s, 1, y + dy) + x;
    const float* rb = srow(s, 2, y + dy) + x;
    for (int l = 0; l < 8; l++) {
      float pixel_x = maxf(0.0f, rx[l] - kRedRampStart);
      float pixel_y = ry[l];
      float pixel_b = maxf(0.0f, rb[l] - (pixel_y + kBlueRampStart));
      float blue_slope = minf(pixel_b, kBlueRampLength);
      float red_slope = minf(pixel_x, kRedRampLength);
      red[l] = red[l] + red_slope;
      blue[l] = blue[l] + blue_slope;
    }
  }
  const float ratio = 30.610615782142737f;
  float overall_red = sum_of_lanes8(red);
  overall_red = minf(overall_red, ratio * kRedRampLength);
  overall_red = overall_red * (red_strength / ratio);
  float overall_blue = sum_of_lanes8(blue);
  overall_blue = minf(overall_blue, ratio * kBlueRampLength);
  overall_blue = overall_blue * (blue_strength / ratio);
  return overall_red + (overall_blue + out_val);
}

/* ref: :114-144 (GammaModulation) */
static float gamma_modulation(const stripe_t* s, size_t x, size_t y, float out_val) {
  const float kBias = 0.16f;
  float overall[8] = {0};
  for (int dy = 0; dy < 8; dy++) {
    const float* rx = srow(s, 0, y + dy) + x;
    const float* ry = srow(s, 1, y + dy) + x;
    for (int l = 0; l < 8; l++) {
      float iny = ry[l] + kBias;
      float inx = rx[l];
      float r = iny - inx, g = iny + inx;
      float ratio_r = ratio_of_derivatives(r, 1);
      float ratio_g = ratio_of_derivatives(g, 1);
      float avg = 0.5f * (ratio_r + ratio_g);
      overall[l] = overall[l] + avg;
    }
  }
  float ratio = sum_of_lanes8(overall) * (1.0f / 64);
  const float kGam = -0.15526878023684174f * 0.693147180559945f;
  return fma32(kGam, orc_fast_log2f(ratio), out_val);
}

/* ref: :249-285 (PerBlockModulations); rect = (bx0, 0, nbx, nby) in stripe blocks,
 * aq_map is the tile-local 8x8 map (stride 8) */
static void per_block_modulations(float butteraugli_target, const stripe_t* s,
                                  float scale, size_t bx0, size_t nbx, size_t nby,
                                  float* aq_map) {
  float base_level = 0.5f * scale;
  float kDampenRampStart = 7.0f, kDampenRampEnd = 14.0f;
  float dampen = 1.0f;
  if (butteraugli_target >= kDampenRampStart) {
    dampen = 1.0f - ((butteraugli_target - kDampenRampStart) /
                     (kDampenRampEnd - kDampenRampStart));
    if (dampen < 0) dampen = 0;
  }
  const float mul = scale * dampen;
  const float add = (1.0f - dampen) * base_level;
  for (size_t iy = 0; iy < nby; iy++) {
    size_t y = iy * 8;
    for (size_t ix = 0; ix < nbx; ix++) {
      size_t x = (bx0 + ix) * 8;
      float out_val = aq_map[iy * 8 + ix];
      out_val = compute_mask(out_val);
      out_val = hf_modulation(s, x, y, out_val);
      out_val = color_modulation(s, x, y, butteraugli_target, out_val);
      out_val = gamma_modulation(s, x, y, out_val);
      aq_map[iy * 8 + ix] = orc_fast_pow2f(out_val * 1.442695041f) * mul + add;
    }
  }
}

/* ref: :296-320 (StoreMin4) */
static void store_min4(float v, float* min0, float* min1, float* min2, float* min3) {
  if (v < *min3) {
    if (v < *min0) {
      *min3 = *min2; *min2 = *min1; *min1 = *min0; *min0 = v;
    } else if (v < *min1) {
      *min3 = *min2; *min2 = *min1; *min1 = v;
    } else if (v < *min2) {
      *min3 = *min2; *min2 = v;
    } else {
      *min3 = v;
    }
  }
}

#define PRE_PITCH 18
/* ref: :322-374 (FuzzyErosion); from has logical size fx*fy (pitch PRE_PITCH) */
static void fuzzy_erosion(size_t rx0, size_t ry0, size_t rxs, size_t rys,
                          const float* from, size_t fxsize, size_t fysize, float* to) {
  for (size_t fy = 0; fy < rys; ++fy) {
    size_t y = fy + ry0;
    size_t ym1 = y >= 1 ? y - 1 : y;
    size_t yp1 = y + 1 < fysize ? y + 1 : y;
    const float* rowt = from + ym1 * PRE_PITCH;
    const float* row = from + y * PRE_PITCH;
    const float* rowb = from + yp1 * PRE_PITCH;
    float* row_out = to + (fy / 2) * 8;
    for (size_t fx = 0; fx < rxs; ++fx) {
      size_t x = fx + rx0;
      size_t xm1 = x >= 1 ? x - 1 : x;
      size_t xp1 = x + 1 < fxsize ? x + 1 : x;
      float min0 = row[x], min1 = row[xm1], min2 = row[xp1], min3 = rowt[xm1], t;
#define SWAP_IF_GT(a, b) if (a > b) { t = a; a = b; b = t; }
      SWAP_IF_GT(min0, min1);
      SWAP_IF_GT(min0, min2);
      SWAP_IF_GT(min0, min3);
      SWAP_IF_GT(min1, min2);
      SWAP_IF_GT(min1, min3);
      SWAP_IF_GT(min2, min3);
#undef SWAP_IF_GT
      store_min4(rowt[x], &min0, &min1, &min2, &min3);
      store_min4(rowt[xp1], &min0, &min1, &min2, &min3);
      store_min4(rowb[xm1], &min0, &min1, &min2, &min3);
      store_min4(rowb[x], &min0, &min1, &min2, &min3);
      store_min4(rowb[xp1], &min0, &min1, &min2, &min3);
      const float kMul = 0.05f;
      float v = kMul * row[x] + kMul * min0 + kMul * min1 + kMul * min2 + kMul * min3;
      if (fx % 2 == 0 && fy % 2 == 0) row_out[fx / 2] = v;
      else row_out[fx / 2] += v;
    }
  }
}

/* ref: :376-505 (ComputeAdaptiveQuantFieldTile, 8-lane model) + wrapper :518-534.
 * rect = (bx0, 0, nbx, nby) in stripe blocks.  Outputs tile-local aq_map and mask
 * (8x8, stride 8) and raw quant u8 (8x8, stride 8). */
static void compute_aq_tile(const stripe_t* s, size_t bx0, size_t nbx, size_t nby,
                            float distance, float inv_scale, float* aq_map,
                            float* mask, uint8_t* raw_quant) {
  const size_t xsize = s->xsize, ysize = s->ysize;
  const float kAcQuant = 0.8294f;
  const float scale = kAcQuant / distance;
  const float match_gamma_offset = (float)0.019;
  const float kXMul = 23.426802998210313f;
  float pre_erosion[PRE_PITCH * PRE_PITCH];
  float diff_buffer[64 + 8 + 8];

  size_t y_start = 0, y_end = nby * 8;
  size_t x0 = bx0 * 8, x1 = x0 + nbx * 8;
  if (x0 != 0) x0 -= 4;
  if (x1 != xsize) x1 += 4;
  if (y_start != 0) y_start -= 4;
  if (y_end != ysize) y_end += 4;
  const size_t pre_xs = (x1 - x0) / 4, pre_ys = (y_end - y_start) / 4;

  for (size_t y = y_start; y < y_end; ++y) {
    size_t y2 = y + 1 < ysize ? y + 1 : y;
    size_t y1 = y > 0 ? y - 1 : y;
    const float* row_in = srow(s, 1, y);
    const float* row_in1 = srow(s, 1, y1);
    const float* row_in2 = srow(s, 1, y2);
    const float* row_x_in = srow(s, 0, y);
    const float* row_x_in1 = srow(s, 0, y1);
    const float* row_x_in2 = srow(s, 0, y2);
    float* row_out = diff_buffer;

    size_t x = x0;
    /* scalar_pixel lambda (:420-441) */
#define SCALAR_PIXEL(X)                                                          \
  do {                                                                           \
    const size_t sx = (X);                                                       \
    const size_t sx2 = sx + 1 < xsize ? sx + 1 : sx;                             \
    const size_t sx1 = sx > 0 ? sx - 1 : sx;                                     \
    const float base =                                                           \
        0.25f * (row_in2[sx] + row_in1[sx] + row_in[sx1] + row_in[sx2]);         \
    const float gammac = ratio_of_derivatives(row_in[sx] + match_gamma_offset, 0); \
    float diff = gammac * (row_in[sx] - base);                                   \
    diff *= diff;                                                                \
    const float base_x =                                                         \
        0.25f * (row_x_in2[sx] + row_x_in1[sx] + row_x_in[sx1] + row_x_in[sx2]); \
    float diff_x = gammac * (row_x_in[sx] - base_x);                             \
    diff_x *= diff_x;                                                            \
    diff += kXMul * diff_x;                                                      \
    diff = masking_sqrt(diff);                                                   \
    if ((y % 4) != 0) row_out[sx - x0] += diff;                                  \
    else row_out[sx - x0] = diff;                                                \
  } while (0)

    if (x0 == 0) {
      SCALAR_PIXEL(x0);
      ++x;
    }
    /* "SIMD" loop (:443-479), 8 lanes */
    for (; x + 1 + LANES < x1; x += LANES) {
      for (size_t l = 0; l < LANES; l++) {
        size_t xx = x + l;
        float in = row_in[xx], in_r = row_in[xx + 1], in_l = row_in[xx - 1];
        float in_t = row_in2[xx], in_b = row_in1[xx];
        float base = 0.25f * ((in_r + in_l) + (in_t + in_b));
        float gammacv = ratio_of_derivatives(in + match_gamma_offset, 0);
        float diff = gammacv * (in - base);
        diff = diff * diff;
        float in_x = row_x_in[xx], in_x_r = row_x_in[xx + 1], in_x_l = row_x_in[xx - 1];
        float in_x_t = row_x_in2[xx], in_x_b = row_x_in1[xx];
        float base_x = 0.25f * ((in_x_r + in_x_l) + (in_x_t + in_x_b));
        float diff_x = gammacv * (in_x - base_x);
        diff_x = diff_x * diff_x;
        diff = fma32(kXMul, diff_x, diff);
        diff = masking_sqrt(diff);
        if ((y & 3) != 0) diff = diff + row_out[xx - x0];
        row_out[xx - x0] = diff;
      }
    }
    for (; x < x1; ++x) SCALAR_PIXEL(x);
#undef SCALAR_PIXEL
    if (y % 4 == 3) {
      float* row_dout = pre_erosion + ((y - y_start) / 4) * PRE_PITCH;
      for (size_t qx = 0; qx < (x1 - x0) / 4; qx++) {
        row_dout[qx] = (row_out[qx * 4] + row_out[qx * 4 + 1] + row_out[qx * 4 + 2] +
                        row_out[qx * 4 + 3]) * 0.25f;
      }
    }
  }
  fuzzy_erosion(x0 % 8 == 0 ? 0 : 1, y_start % 8 == 0 ? 0 : 1, nbx * 2, nby * 2,
                pre_erosion, pre_xs, pre_ys, aq_map);
  for (size_t y = 0; y < nby; ++y)
    for (size_t x = 0; x < nbx; ++x) /* ComputeMaskForAcStrategyUse (:46-50) */
      mask[y * 8 + x] = 1.0f / (aq_map[y * 8 + x] + 0.001f);
  per_block_modulations(distance, s, scale, bx0, nbx, nby, aq_map);
  for (size_t y = 0; y < nby; ++y)
    for (size_t x = 0; x < nbx; ++x) {
      int v = (int)(aq_map[y * 8 + x] * inv_scale + 0.5f);
      raw_quant[y * 8 + x] = (uint8_t)clampi(v, 1, 255);
    }
}

/* ------------------------------------------------------------------------- */
/* enc_chroma_from_luma.cc                                                   */
/* ------------------------------------------------------------------------- */

static const float kInvColorFactor = 1.0f / 84; /* ref: chroma_from_luma.h:21 */
static inline float y_to_x_ratio(int8_t x) { return x * kInvColorFactor; }
static inline float y_to_b_ratio(int8_t b) { return 1.0f + b * kInvColorFactor; }

typedef struct {
  float ca[8], cb[8];
} cfl_acc;

/* ref: enc_chroma_from_luma.cc:40-62 (FindBestMultiplier), streaming form */
static inline void cfl_accumulate(cfl_acc* acc, const float* m, const float* sv, float base) {
  for (int l = 0; l < 8; l++) {
    float a = kInvColorFactor * m[l];
    float b = base * m[l] - sv[l];
    acc->ca[l] = fma32(a, a, acc->ca[l]);
    acc->cb[l] = fma32(a, b, acc->cb[l]);
  }
}
static int32_t cfl_finish(const cfl_acc* acc, size_t num, float distance_mul) {
  if (num == 0) return 0;
  float x = -sum_of_lanes8(acc->cb) / (sum_of_lanes8(acc->ca) + num * distance_mul * 0.5f);
  return (int32_t)maxf(-128.0f, minf(127.0f, roundf(x)));
}

/* ref: enc_chroma_from_luma.cc:64-131 (ComputeCmapTile) */
static void compute_cmap_tile(const stripe_t* s, size_t bx0, size_t nbx, size_t nby,
                              const dequant_matrices* dq, int8_t* ytox, int8_t* ytob) {
  const float kDistanceMultiplierAC = 1e-3f;
  cfl_acc accx, accb;
  memset(&accx, 0, sizeof accx);
  memset(&accb, 0, sizeof accb);
  size_t num_ac = 0;
  const float* qm_x = dq_inv_matrix(dq, STRAT_DCT, 0);
  const float* qm_b = dq_inv_matrix(dq, STRAT_DCT, 2);
  float block_y[64], block_x[64], block_b[64];
  for (size_t y = 0; y < nby; ++y) {
    for (size_t x = bx0; x < bx0 + nbx; x++) {
      orc_dct8x8(srow(s, 1, y * 8) + x * 8, STRIPE_PITCH, block_y);
      orc_dct8x8(srow(s, 0, y * 8) + x * 8, STRIPE_PITCH, block_x);
      orc_dct8x8(srow(s, 2, y * 8) + x * 8, STRIPE_PITCH, block_b);
      block_y[0] = 0;
      block_x[0] = 0;
      block_b[0] = 0;
      for (int i = 0; i < 64; i += 8) {
        float yx[8], cx[8], yb[8], cb[8];
        for (int l = 0; l < 8; l++) {
          yx[l] = block_y[i + l] * qm_x[i + l];
          cx[l] = block_x[i + l] * qm_x[i + l];
          yb[l] = block_y[i + l] * qm_b[i + l];
          cb[l] = block_b[i + l] * qm_b[i + l];
        }
        cfl_accumulate(&accx, yx, cx, 0.0f);
        cfl_accumulate(&accb, yb, cb, 1.0f);
        num_ac += 8;
      }
    }
  }
  *ytox = (int8_t)cfl_finish(&accx, num_ac, kDistanceMultiplierAC);
  *ytob = (int8_t)cfl_finish(&accb, num_ac, kDistanceMultiplierAC);
}

/* ------------------------------------------------------------------------- */
/* enc_ac_strategy.cc                                                        */
/* ------------------------------------------------------------------------- */

/* ref: enc_ac_strategy.cc:51-146 (EstimateEntropy).  (bx,by) = block in stripe,
 * (cx,cy) = block in tile (indexes qf/maskf, tile-local 8x8, stride 8). */
static float estimate_entropy(int strategy, const stripe_t* s, size_t bx, size_t by,
                              size_t cx, size_t cy, float distance,
                              const dequant_matrices* dq, const float* qf,
                              const float* maskf, int8_t ytox, int8_t ytob) {
  const int cbx = kCoveredX[strategy], cby = kCoveredY[strategy];
  const size_t num_blocks = (size_t)cbx * cby;
  const size_t size = num_blocks * 64;
  float block[3 * 128];
  for (int c = 0; c < 3; c++)
    transform_from_pixels(strategy, srow(s, c, by * 8) + bx * 8, STRIPE_PITCH, block + size * c);
  float quant = 0, masking = 0;
  for (int iy = 0; iy < cby; iy++)
    for (int ix = 0; ix < cbx; ix++) {
      quant = maxf(quant, qf[(cy + iy) * 8 + cx + ix]);
      masking = maxf(masking, maskf[(cy + iy) * 8 + cx + ix]);
    }
  const float kInfoLossMultiplier = 138.0f;
  const float kInfoLossMultiplier2 = (float)50.46839691767866;
  float entropy = 0.0f;
  float info_loss[8] = {0}, info_loss2[8] = {0};
  const float cmap_factors[3] = {y_to_x_ratio(ytox), 0.0f, y_to_b_ratio(ytob)};
  for (int c = 0; c < 3; c++) {
    const float* inv_matrix = dq_inv_matrix(dq, strategy, c);
    const float cmap_factor = cmap_factors[c];
    float entropy_v[8] = {0}, nzeros_v[8] = {0};
    float slope = minf(1.0f, distance * (1.0f / 3));
    float cost_of_1 = 1 + slope * 8.8703248061477744f;
    const float kCost2 = 4.4628149885273363f;
    const float kCostDelta = 5.3359184934516337f;
    for (size_t i = 0; i < size; i += 8) {
      for (int l = 0; l < 8; l++) {
        float in = block[c * size + i + l];
        float in_y = block[size + i + l] * cmap_factor;
        float im = inv_matrix[i + l];
        float val = (in - in_y) * (im * quant);
        float rval = nearbyintf(val);
        float diff = fabsf(val - rval);
        info_loss[l] = info_loss[l] + diff;
        info_loss2[l] = fma32(diff, diff, info_loss2[l]);
        float q = fabsf(rval);
        entropy_v[l] = entropy_v[l] + (q >= 1.5f ? kCost2 : 0.0f);
        entropy_v[l] = fma32(sqrtf(q), kCostDelta, entropy_v[l]);
        nzeros_v[l] = nzeros_v[l] + (q == 0.0f ? 0.0f : 1.0f);
      }
    }
    for (int l = 0; l < 8; l++) entropy_v[l] = fma32(nzeros_v[l], cost_of_1, entropy_v[l]);
    entropy += sum_of_lanes8(entropy_v);
    size_t num_nzeros = (size_t)sum_of_lanes8(nzeros_v);
    size_t nbits = ceil_log2_nonzero(num_nzeros + 1) + 1;
    const float kZerosMul = 7.565053364251793f;
    entropy += kZerosMul * (ceil_log2_nonzero(nbits + 17) + nbits);
  }
  float infoloss = sum_of_lanes8(info_loss);
  float infoloss2 = (float)sqrt((double)(num_blocks * sum_of_lanes8(info_loss2)));
  float info_loss_score = (kInfoLossMultiplier * infoloss + kInfoLossMultiplier2 * infoloss2);
  return entropy + masking * info_loss_score;
}

/* ref: enc_ac_strategy.cc:167-238 (FindBest16x16Transform).  strat points at the
 * image-absolute ac_strategy byte of block (tile bx0+cx, by0+cy); sstride = grid pitch.
 * ent8 (optional) receives the 8 candidate entropies. */
/* ref: enc_ac_strategy.cc:178-185: mul8x8 / mul16x8 are function-local `static const`s of the
 * reference, i.e. computed from the distance of the FIRST call in the process and reused for every
 * later frame.  A single-image process (cjxl_tiny) never notices.  g_strategy_distance > 0
 * reproduces a later call of such a process (the two multipliers come from that distance);
 * 0 (default) uses the frame's own distance. */
static float g_strategy_distance = 0.0f;
void orc_set_strategy_distance(float first_call_distance) { g_strategy_distance = first_call_distance; }

static void find_best_16x16(const stripe_t* s, size_t bx, size_t by, size_t cx, size_t cy,
                            float distance, const dequant_matrices* dq, const float* qf,
                            const float* maskf, int8_t ytox, int8_t ytob, uint8_t* strat,
                            size_t sstride, float* ent8) {
  const float k8x8mul1 = (float)(-0.55 * 0.75f);
  const float k8x8mul2 = 1.0735757687292623f * 0.75f;
  const float k8x8base = (float)1.4;
  const float sdist = g_strategy_distance > 0.0f ? g_strategy_distance : distance;
  const float mul8x8 = k8x8mul2 + k8x8mul1 / (sdist + k8x8base);
  const float k8X16mul1 = (float)-0.55;
  const float k8X16mul2 = (float)0.9019587899705066;
  const float k8X16base = (float)1.6;
  const float mul16x8 = k8X16mul2 + k8X16mul1 / (sdist + k8X16base);
  float entropy[2][2];
  for (size_t dy = 0; dy < 2; ++dy)
    for (size_t dx = 0; dx < 2; ++dx) {
      float e = 3.0f * mul8x8;
      e += mul8x8 * estimate_entropy(STRAT_DCT, s, bx + cx + dx, by + cy + dy, cx + dx,
                                     cy + dy, distance, dq, qf, maskf, ytox, ytob);
      entropy[dy][dx] = e;
    }
  float e16x8_left = mul16x8 * estimate_entropy(STRAT_DCT16X8, s, bx + cx, by + cy, cx, cy,
                                                distance, dq, qf, maskf, ytox, ytob);
  float e16x8_right = mul16x8 * estimate_entropy(STRAT_DCT16X8, s, bx + cx + 1, by + cy,
                                                 cx + 1, cy, distance, dq, qf, maskf, ytox, ytob);
  float e8x16_top = mul16x8 * estimate_entropy(STRAT_DCT8X16, s, bx + cx, by + cy, cx, cy,
                                               distance, dq, qf, maskf, ytox, ytob);
  float e8x16_bottom = mul16x8 * estimate_entropy(STRAT_DCT8X16, s, bx + cx, by + cy + 1, cx,
                                                  cy + 1, distance, dq, qf, maskf, ytox, ytob);
  if (ent8) {
    ent8[0] = entropy[0][0]; ent8[1] = entropy[0][1];
    ent8[2] = entropy[1][0]; ent8[3] = entropy[1][1];
    ent8[4] = e16x8_left; ent8[5] = e16x8_right;
    ent8[6] = e8x16_top; ent8[7] = e8x16_bottom;
  }
  float cost16x8 = minf(e16x8_left, entropy[0][0] + entropy[1][0]) +
                   minf(e16x8_right, entropy[0][1] + entropy[1][1]);
  float cost8x16 = minf(e8x16_top, entropy[0][0] + entropy[0][1]) +
                   minf(e8x16_bottom, entropy[1][0] + entropy[1][1]);
#define SET_STRAT(X, Y, T)                                                     \
  do {                                                                         \
    for (int iy = 0; iy < kCoveredY[T]; iy++)                                  \
      for (int ix = 0; ix < kCoveredX[T]; ix++)                                \
        strat[((Y) + iy) * sstride + (X) + ix] =                               \
            (uint8_t)(((T) << 1) | ((iy | ix) == 0 ? 1 : 0));                  \
  } while (0)
  if (cost16x8 < cost8x16) {
    if (e16x8_left < entropy[0][0] + entropy[1][0]) SET_STRAT(0, 0, STRAT_DCT16X8);
    if (e16x8_right < entropy[0][1] + entropy[1][1]) SET_STRAT(1, 0, STRAT_DCT16X8);
  } else {
    if (e8x16_top < entropy[0][0] + entropy[0][1]) SET_STRAT(0, 0, STRAT_DCT8X16);
    if (e8x16_bottom < entropy[1][0] + entropy[1][1]) SET_STRAT(0, 1, STRAT_DCT8X16);
  }
#undef SET_STRAT
}

/* ref: enc_ac_strategy.cc:240-266 (AdjustQuantField) on an image-absolute grid */
static void adjust_quant_field(const uint8_t* strat, uint8_t* quant, size_t stride,
                               size_t nbx, size_t nby) {
  for (size_t y = 0; y < nby; ++y)
    for (size_t x = 0; x < nbx; ++x) {
      uint8_t a = strat[y * stride + x];
      if (!(a & 1)) continue;
      int t = a >> 1;
      uint8_t m = quant[y * stride + x];
      for (int iy = 0; iy < kCoveredY[t]; iy++)
        for (int ix = 0; ix < kCoveredX[t]; ix++) {
          uint8_t q = quant[(y + iy) * stride + x + ix];
          if (q > m) m = q;
        }
      for (int iy = 0; iy < kCoveredY[t]; iy++)
        for (int ix = 0; ix < kCoveredX[t]; ix++) quant[(y + iy) * stride + x + ix] = m;
    }
}

/* ------------------------------------------------------------------------- */
/* enc_group.cc                                                              */
/* ------------------------------------------------------------------------- */

/* ref: enc_group.cc:186-218 (AdjustQuantBias) */
static inline float adjust_quant_bias(int c, int32_t quant_i, const float* biases) {
  float quant = (float)quant_i;
  uint32_t sign = f2u(quant) & 0x80000000u;
  float abs_quant = u2f(f2u(quant) & 0x7FFFFFFFu);
  int is_01 = abs_quant < 1.125f;
  int not_0 = abs_quant > 0.0f;
  float one_bias = not_0 ? u2f(f2u(biases[c]) ^ sign) : 0.0f;
  float bias = nfma32(biases[3], 1.0f / quant, quant); /* ApproximateReciprocal := 1/x */
  return is_01 ? one_bias : bias;
}

/* ref: enc_group.cc:221-278 (QuantizeBlockAC); xsize>=ysize canonical dims */
static void quantize_block_ac(const float* block_in, int c, const float* qm, int32_t quant,
                              float scale, float qm_multiplier, size_t xsize, size_t ysize,
                              int32_t* block_out) {
  const float qac = scale * quant;
  float thres[4] = {0.58f, 0.635f, 0.66f, 0.7f};
  if (c == 0)
    for (int i = 1; i < 4; ++i) thres[i] += 0.08f;
  if (c == 2)
    for (int i = 1; i < 4; ++i) thres[i] = 0.75f;
  if (xsize > 1 || ysize > 1)
    for (int i = 0; i < 4; ++i)
      thres[i] -= clampf(0.003f * xsize * ysize, 0.f, (c > 0 ? 0.08f : 0.12f));
  const float quantv = qac * qm_multiplier;
  for (size_t y = 0; y < ysize * 8; y++) {
    size_t yfix = (size_t)(y >= ysize * 8 / 2) * 2;
    const size_t off = y * 8 * xsize;
    for (size_t x = 0; x < xsize * 8; x++) {
      float thr;
      if (xsize == 1) thr = (x % 8) >= 4 ? thres[yfix + 1] : thres[yfix];
      else thr = thres[yfix + (size_t)((x / 8) * 8 >= xsize * 8 / 2)];
      float q = qm[off + x] * quantv;
      float in = block_in[off + x];
      float val = q * in;
      int nz = fabsf(val) >= thr;
      block_out[off + x] = nz ? (int32_t)nearbyintf(val) : 0;
    }
  }
}

/* ref: enc_group.cc:281-302 (QuantizeRoundtripYBlockAC) */
static void quantize_roundtrip_y(const float* qm, const float* dqm, float scale, int32_t quant,
                                 size_t xsize, size_t ysize, float* inout, int32_t* quantized) {
  quantize_block_ac(inout, 1, qm, quant, scale, 1.0f, xsize, ysize, quantized);
  const float inv_qac = (float)(1.0 / (scale * quant));
  const float kDefaultQuantBias[4] = {1.0f - 0.05465007330715401f, 1.0f - 0.07005449891748593f,
                                      1.0f - 0.049935103337343655f, 0.145f};
  for (size_t k = 0; k < 64 * xsize * ysize; k++) {
    float adj = adjust_quant_bias(1, quantized[k], kDefaultQuantBias);
    inout[k] = (adj * dqm[k]) * inv_qac;
  }
}

/* ref: ac_context.h:64-114 */
static inline size_t block_context(size_t c, uint8_t code) { return ORC_kBlockContextMap[c * 27 + code]; }
static inline size_t zero_density_context(size_t nonzeros_left, size_t k, size_t covered,
                                          size_t log2_covered, size_t prev) {
  nonzeros_left = (nonzeros_left + covered - 1) >> log2_covered;
  k >>= log2_covered;
  return (ORC_kCoeffNumNonzeroContext[nonzeros_left] + ORC_kCoeffFreqContext[k]) * 2 + prev;
}
static inline size_t zero_density_contexts_offset(size_t block_ctx) { return 4 * 37 + 458 * block_ctx; }
static inline size_t non_zero_context(size_t non_zeros, size_t block_ctx) {
  size_t b = non_zeros < 8 ? non_zeros : non_zeros >= 64 ? 36 : 4 + non_zeros / 2;
  return b * 4 + block_ctx;
}

typedef struct {
  uint8_t* data;
  size_t size, cap;
} byte_buf;

static void emit_token(byte_buf* b, size_t ctx, uint32_t value) {
  if (b->size + 3 > b->cap) {
    b->cap = b->cap ? b->cap * 2 : 4096;
    b->data = (uint8_t*)realloc(b->data, b->cap);
  }
  b->data[b->size++] = ORC_kACContextMap[ctx];
  b->data[b->size++] = (uint8_t)(value & 0xFF);
  b->data[b->size++] = (uint8_t)((value >> 8) & 0xFF);
}

/* ref: enc_group.cc:150-160 */
static inline int32_t predict_from_top_and_left(const uint8_t* row_top, const uint8_t* row,
                                                size_t x, int32_t default_val) {
  if (x == 0) return row_top == NULL ? default_val : row_top[x];
  if (row_top == NULL) return row[x - 1];
  return (row_top[x] + row[x - 1] + 1) / 2;
}

typedef struct {
  /* image-absolute grids */
  size_t bstride;       /* xsize_blocks */
  size_t tstride;       /* xsize_tiles */
  int16_t* quant_dc[3];
  uint8_t* raw_quant;
  uint8_t* strategy;
  int8_t* ytox;
  int8_t* ytob;
} frame_grids;

/* ref: enc_group.cc:304-497 (WriteACGroup) for one stripe.
 * (bx_img0, by_img0): image-absolute block origin of the stripe; nbx x nby blocks.
 * num_nzeros: [3][32][32] per-group grid; nzeros_by0 = row offset of this stripe in it. */
static void write_ac_stripe(const stripe_t* s, size_t bx_img0, size_t by_img0, size_t nbx,
                            size_t nby, const dequant_matrices* dq, float scale, float scale_dc,
                            uint32_t x_qm_scale, frame_grids* g, uint8_t (*num_nzeros)[32][32],
                            size_t nzeros_by0, byte_buf* out) {
  static const float kInvDCQuant[3] = {4096.0f, 512.0f, 256.0f};
  const float kDCQuant1 = 1.0f / kInvDCQuant[1];
  float inv_factor[3];
  float cfl_factor[3] = {0.0f, 0.0f, kInvDCQuant[2] * kDCQuant1};
  for (int c = 0; c < 3; ++c) inv_factor[c] = kInvDCQuant[c] * scale_dc;
  const float x_qm_mul = powf(1.25f, x_qm_scale - 2.0f);
  float coeffs_in[3 * 128];
  int32_t quantized[3 * 128];
  float tmp_dc[4];
  const size_t tmp_dc_stride = 2;

  for (size_t by = 0; by < nby; ++by) {
    const size_t iby = by_img0 + by;
    const size_t nzeros_by = nzeros_by0 + by;
    for (size_t bx = 0; bx < nbx; ++bx) {
      const size_t ibx = bx_img0 + bx;
      const size_t tx = ibx / 8, ty = iby / 8;
      const float x_factor = y_to_x_ratio(g->ytox[ty * g->tstride + tx]);
      const float b_factor = y_to_b_ratio(g->ytob[ty * g->tstride + tx]);
      const uint8_t acs = g->strategy[iby * g->bstride + ibx];
      if (!(acs & 1)) continue;
      const int strategy = acs >> 1;
      size_t cx = kCoveredX[strategy], cy = kCoveredY[strategy];
      if (cy > cx) { size_t t = cx; cx = cy; cy = t; }
      const size_t covered_blocks = cx * cy;
      const size_t size = 64 * covered_blocks;
      const int32_t quant_ac = g->raw_quant[iby * g->bstride + ibx];

      transform_from_pixels(strategy, srow(s, 1, by * 8) + bx * 8, STRIPE_PITCH, coeffs_in + size);
      dc_from_lowest_frequencies(strategy, coeffs_in + size, tmp_dc, tmp_dc_stride);
      for (int iy = 0; iy < kCoveredY[strategy]; ++iy)
        for (int ix = 0; ix < kCoveredX[strategy]; ++ix)
          g->quant_dc[1][(iby + iy) * g->bstride + ibx + ix] =
              (int16_t)roundf(inv_factor[1] * tmp_dc[iy * tmp_dc_stride + ix]);
      quantize_roundtrip_y(dq_inv_matrix(dq, strategy, 1), dq_matrix(dq, strategy, 1), scale,
                           quant_ac, cx, cy, coeffs_in + size, quantized + size);

      transform_from_pixels(strategy, srow(s, 0, by * 8) + bx * 8, STRIPE_PITCH, coeffs_in);
      transform_from_pixels(strategy, srow(s, 2, by * 8) + bx * 8, STRIPE_PITCH, coeffs_in + 2 * size);
      for (size_t k = 0; k < size; k++) {
        float in_y = coeffs_in[size + k];
        coeffs_in[k] = nfma32(x_factor, in_y, coeffs_in[k]);
        coeffs_in[2 * size + k] = nfma32(b_factor, in_y, coeffs_in[2 * size + k]);
      }
      for (int c = 0; c <= 2; c += 2) {
        quantize_block_ac(coeffs_in + c * size, c, dq_inv_matrix(dq, strategy, c), quant_ac, scale,
                          c == 0 ? x_qm_mul : (float)1.0, cx, cy, quantized + c * size);
        dc_from_lowest_frequencies(strategy, coeffs_in + c * size, tmp_dc, tmp_dc_stride);
        for (int iy = 0; iy < kCoveredY[strategy]; ++iy)
          for (int ix = 0; ix < kCoveredX[strategy]; ++ix) {
            size_t pos = (iby + iy) * g->bstride + ibx + ix;
            g->quant_dc[c][pos] = (int16_t)roundf(tmp_dc[iy * tmp_dc_stride + ix] * inv_factor[c] -
                                                  g->quant_dc[1][pos] * cfl_factor[c]);
          }
      }

      /* Tokenize (:444-494) */
      const size_t log2_covered_blocks = covered_blocks == 1 ? 0 : 1;
      static const int kChan[3] = {1, 0, 2};
      for (int ci = 0; ci < 3; ci++) {
        const int c = kChan[ci];
        const int32_t* block = quantized + c * size;
        /* NumNonZero8x8ExceptDC (:109-148) / NumNonZeroExceptLLF (:51-105) */
        int32_t nzeros = 0;
        for (size_t k = 0; k < size; k++) {
          if (k < cx) continue; /* LLF: first cx entries of row 0 (cy == 1) */
          if (block[k] != 0) nzeros++;
        }
        if (covered_blocks == 1) {
          num_nzeros[c][nzeros_by][bx] = (uint8_t)nzeros;
        } else {
          uint8_t shifted = (uint8_t)((nzeros + covered_blocks - 1) >> log2_covered_blocks);
          for (int iy = 0; iy < kCoveredY[strategy]; iy++)
            for (int ix = 0; ix < kCoveredX[strategy]; ix++)
              num_nzeros[c][nzeros_by + iy][bx + ix] = shifted;
        }
        const uint8_t* order = &ORC_kCoeffOrder[strategy == STRAT_DCT ? 0 : 64];
        const uint8_t* row_top = nzeros_by == 0 ? NULL : num_nzeros[c][nzeros_by - 1];
        int32_t predicted = predict_from_top_and_left(row_top, num_nzeros[c][nzeros_by], bx, 32);
        const size_t block_ctx = block_context(c, kStrategyCode[strategy]);
        const size_t nzero_ctx = non_zero_context(predicted, block_ctx);
        const size_t histo_offset = zero_density_contexts_offset(block_ctx);
        emit_token(out, nzero_ctx, (uint32_t)nzeros);
        size_t prev = (nzeros > (int32_t)(size / 16)) ? 0 : 1;
        for (size_t k = covered_blocks; k < size && nzeros != 0; ++k) {
          int32_t coeff = block[order[k]];
          size_t ctx = histo_offset +
                       zero_density_context(nzeros, k, covered_blocks, log2_covered_blocks, prev);
          emit_token(out, ctx, pack_signed(coeff));
          prev = coeff != 0;
          nzeros -= prev;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* enc_frame.cc: ProcessTile / ProcessDCGroup loops                          */
/* ------------------------------------------------------------------------- */

static void* xcalloc(size_t n, size_t sz) {
  void* p = calloc(n ? n : 1, sz);
  if (!p) abort();
  return p;
}

/* One 256x256 group (the reference's unit of independent work, enc_frame.cc:716-757): everything the hot path
 * computes for it.  `s` and `num_nzeros` are the caller's scratch (one set per thread). */
typedef struct {
  const float* const* planes;
  size_t stride, xsize, ysize;
  int force_dct8;
  const orc_distance_params* distp;
  const dequant_matrices* dq;
  orc_frame* f;
  frame_grids g;
  size_t cells_x;
} group_job;

static void encode_group(const group_job* J, size_t dc_gx, size_t dc_gy, size_t gix, size_t dc_xgroups, stripe_t* s,
                         uint8_t (*num_nzeros)[32][32]) {
  const float* const* planes = J->planes;
  const size_t stride = J->stride, xsize = J->xsize, ysize = J->ysize;
  const int force_dct8 = J->force_dct8;
  const orc_distance_params distp = *J->distp;
  const dequant_matrices dq = *J->dq;
  orc_frame* f = J->f;
  frame_grids g = J->g;
  const size_t cells_x = J->cells_x;
  {
    const size_t gx = gix % dc_xgroups, gy = gix / dc_xgroups;
    const size_t image_gx = dc_gx * 8 + gx, image_gy = dc_gy * 8 + gy;
    const size_t group_index = image_gy * f->xsize_groups + image_gx;
    const size_t gw = xsize - image_gx * 256 < 256 ? xsize - image_gx * 256 : 256;
    const size_t gh = ysize - image_gy * 256 < 256 ? ysize - image_gy * 256 : 256;
    const size_t g_xtiles = div_ceil(gw, 64), g_ytiles = div_ceil(gh, 64);
    byte_buf tok = {0, 0, 0};
    for (size_t ty = 0; ty < g_ytiles; ++ty) {
      const size_t image_ty = image_gy * 4 + ty;
      const size_t sx0 = image_gx * 256, sy0 = image_ty * 64;
      const size_t sw = gw;
      const size_t sh = ysize - sy0 < 64 ? ysize - sy0 : 64;
      const size_t sxb = div_ceil(sw, 8), syb = div_ceil(sh, 8);
      const size_t bx_img0 = image_gx * 32, by_img0 = image_ty * 8;
      copy_and_pad(planes, stride, sx0, sy0, sw, sh, s);
      for (size_t y = 0; y < s->ysize; y++)
        orc_to_xyb(s->px[0][y], s->px[1][y], s->px[2][y], s->xsize);
      for (int c = 0; c < 3 && f->xyb[c]; c++)
        for (size_t y = 0; y < s->ysize; y++)
          memcpy(f->xyb[c] + (by_img0 * 8 + y) * (f->xsize_blocks * 8) + bx_img0 * 8,
                 s->px[c][y], s->xsize * sizeof(float));
      for (size_t tx = 0; tx < g_xtiles; ++tx) {
        /* ref: ProcessTile enc_frame.cc:648-683 */
        const size_t tbx0 = tx * 8;
        const size_t tnbx = sxb - tbx0 < 8 ? sxb - tbx0 : 8;
        const size_t tnby = syb < 8 ? syb : 8;
        float aq_map[64], mask[64];
        uint8_t rq[64];
        memset(aq_map, 0, sizeof aq_map);
        memset(mask, 0, sizeof mask);
        compute_aq_tile(s, tbx0, tnbx, tnby, distp.distance, distp.inv_scale, aq_map, mask, rq);
        for (size_t y = 0; y < tnby; y++)
          for (size_t x = 0; x < tnbx; x++) {
            size_t pos = (by_img0 + y) * g.bstride + bx_img0 + tbx0 + x;
            g.raw_quant[pos] = rq[y * 8 + x];
            if (f->quant_field) {
              f->quant_field[pos] = aq_map[y * 8 + x];
              f->masking[pos] = mask[y * 8 + x];
            }
          }
        int8_t ytox = 0, ytob = 0;
        compute_cmap_tile(s, tbx0, tnbx, tnby, &dq, &ytox, &ytob);
        const size_t itx = image_gx * 4 + tx;
        g.ytox[image_ty * g.tstride + itx] = ytox;
        g.ytob[image_ty * g.tstride + itx] = ytob;
        if (!force_dct8) {
          for (size_t cy = 0; cy + 1 < tnby; cy += 2)
            for (size_t cx = 0; cx + 1 < tnbx; cx += 2) {
              size_t abx = bx_img0 + tbx0 + cx, aby = by_img0 + cy;
              find_best_16x16(s, tbx0, 0, cx, cy, distp.distance, &dq, aq_map, mask, ytox,
                              ytob, g.strategy + aby * g.bstride + abx, g.bstride,
                              f->entropy8 ? f->entropy8 + ((aby / 2) * cells_x + abx / 2) * 8 : NULL);
            }
          adjust_quant_field(g.strategy + by_img0 * g.bstride + bx_img0 + tbx0,
                             g.raw_quant + by_img0 * g.bstride + bx_img0 + tbx0, g.bstride,
                             tnbx, tnby);
        }
      }
      write_ac_stripe(s, bx_img0, by_img0, sxb, syb, &dq, distp.scale, distp.scale_dc,
                      distp.x_qm_scale, &g, num_nzeros, ty * 8, &tok);
    }
    f->group_tokens[group_index] = tok.data;
    f->group_token_bytes[group_index] = tok.size;
  }
}

/* The groups of the frame in the reference's order (DC groups in raster order, groups in raster order inside a DC
 * group, enc_frame.cc:839-844 / :716), as a flat list: entry i = (dc_gx, dc_gy, gix, dc_xgroups). */
typedef struct {
  const group_job* job;
  size_t (*list)[4];
  size_t count;
  size_t next;       /* shared cursor (atomic) */
} group_queue;

static void* group_worker(void* arg) {
  group_queue* q = (group_queue*)arg;
  stripe_t* s = (stripe_t*)xcalloc(1, sizeof(stripe_t));
  uint8_t (*num_nzeros)[32][32] = (uint8_t(*)[32][32])xcalloc(3, 32 * 32);
  for (;;) {
    const size_t i = __atomic_fetch_add(&q->next, 1, __ATOMIC_RELAXED);
    if (i >= q->count) break;
    encode_group(q->job, q->list[i][0], q->list[i][1], q->list[i][2], q->list[i][3], s, num_nzeros);
  }
  free(num_nzeros);
  free(s);
  return NULL;
}

int orc_encode_hot_path_threads(const float* const planes[3], size_t stride, size_t xsize, size_t ysize,
                                float distance, int force_dct8, int nthreads, int keep_intermediates, orc_frame* f) {
  memset(f, 0, sizeof *f);
  if (xsize == 0 || ysize == 0 || !(distance > 0)) return 1;
  /* ref quirk F12: images that fit one 8x8 block trap in the reference. */
  if (xsize <= 8 && ysize <= 8) return 2;
  orc_distance_params distp;
  orc_compute_distance_params(distance, &distp);
  dequant_matrices dq;
  dequant_matrices_init(&dq);

  f->xsize = xsize;
  f->ysize = ysize;
  f->xsize_blocks = div_ceil(xsize, 8);
  f->ysize_blocks = div_ceil(ysize, 8);
  f->xsize_tiles = div_ceil(xsize, 64);
  f->ysize_tiles = div_ceil(ysize, 64);
  f->xsize_groups = div_ceil(xsize, 256);
  f->ysize_groups = div_ceil(ysize, 256);
  const size_t nblocks = f->xsize_blocks * f->ysize_blocks;
  const size_t ntiles = f->xsize_tiles * f->ysize_tiles;
  const size_t ngroups = f->xsize_groups * f->ysize_groups;
  for (int c = 0; c < 3; c++) {
    f->quant_dc[c] = (int16_t*)xcalloc(nblocks, sizeof(int16_t));
    if (keep_intermediates) f->xyb[c] = (float*)xcalloc(nblocks * 64, sizeof(float));
  }
  f->raw_quant_field = (uint8_t*)xcalloc(nblocks, 1);
  f->ac_strategy = (uint8_t*)xcalloc(nblocks, 1);
  memset(f->ac_strategy, (STRAT_DCT << 1) | 1, nblocks); /* FillDCT8, dc_group_data.h:28 */
  f->ytox_map = (int8_t*)xcalloc(ntiles, 1);
  f->ytob_map = (int8_t*)xcalloc(ntiles, 1);
  f->group_tokens = (uint8_t**)xcalloc(ngroups, sizeof(uint8_t*));
  f->group_token_bytes = (size_t*)xcalloc(ngroups, sizeof(size_t));
  const size_t cells_x = f->xsize_blocks / 2 + 1, cells_y = f->ysize_blocks / 2 + 1;
  if (keep_intermediates) {
    /* (debug outputs for the parity tests: 13 bytes per pixel that the timed CPU baseline does not fill) */
    f->quant_field = (float*)xcalloc(nblocks, sizeof(float));
    f->masking = (float*)xcalloc(nblocks, sizeof(float));
    f->entropy8 = (float*)xcalloc(cells_x * cells_y * 8, sizeof(float));
    for (size_t i = 0; i < cells_x * cells_y * 8; i++) f->entropy8[i] = NAN;
  }

  frame_grids g;
  g.bstride = f->xsize_blocks;
  g.tstride = f->xsize_tiles;
  for (int c = 0; c < 3; c++) g.quant_dc[c] = f->quant_dc[c];
  g.raw_quant = f->raw_quant_field;
  g.strategy = f->ac_strategy;
  g.ytox = f->ytox_map;
  g.ytob = f->ytob_map;

  group_job job;
  job.planes = planes;
  job.stride = stride;
  job.xsize = xsize;
  job.ysize = ysize;
  job.force_dct8 = force_dct8;
  job.distp = &distp;
  job.dq = &dq;
  job.f = f;
  job.g = g;
  job.cells_x = cells_x;
  size_t (*list)[4] = (size_t(*)[4])xcalloc(ngroups, sizeof(size_t[4]));
  size_t count = 0;
  const size_t xsize_dc_groups = div_ceil(xsize, 2048), ysize_dc_groups = div_ceil(ysize, 2048);
  for (size_t dc_gy = 0; dc_gy < ysize_dc_groups; dc_gy++)
    for (size_t dc_gx = 0; dc_gx < xsize_dc_groups; dc_gx++) {
      /* ref: ProcessDCGroup enc_frame.cc:685-763 */
      const size_t dcw = xsize - dc_gx * 2048 < 2048 ? xsize - dc_gx * 2048 : 2048;
      const size_t dch = ysize - dc_gy * 2048 < 2048 ? ysize - dc_gy * 2048 : 2048;
      const size_t dc_xgroups = div_ceil(dcw, 256), dc_ygroups = div_ceil(dch, 256);
      for (size_t gix = 0; gix < dc_xgroups * dc_ygroups; ++gix) {
        list[count][0] = dc_gx;
        list[count][1] = dc_gy;
        list[count][2] = gix;
        list[count][3] = dc_xgroups;
        count++;
      }
    }
  group_queue q;
  q.job = &job;
  q.list = list;
  q.count = count;
  q.next = 0;
  /* The reference runs the groups one after the other on the calling thread (its ThreadPool argument is unused,
   * SURVEY.md F3); nthreads > 1 spreads the same independent units over threads -- same results, group by group
   * (every group writes its own part of the grids and its own token buffer). */
  if (nthreads > (int)count) nthreads = (int)count;
  if (nthreads <= 1) {
    group_worker(&q);
  } else {
    pthread_t* th = (pthread_t*)xcalloc((size_t)nthreads, sizeof(pthread_t));
    int started = 0;
    for (int t = 0; t < nthreads - 1; t++) {
      if (pthread_create(&th[t], NULL, group_worker, &q) != 0) break;
      started++;
    }
    group_worker(&q);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
  }
  free(list);
  return 0;
}

int orc_encode_hot_path(const float* const planes[3], size_t stride, size_t xsize, size_t ysize,
                        float distance, int force_dct8, orc_frame* f) {
  return orc_encode_hot_path_threads(planes, stride, xsize, ysize, distance, force_dct8, 1, 1, f);
}

void orc_frame_free(orc_frame* f) {
  if (!f) return;
  for (int c = 0; c < 3; c++) {
    free(f->quant_dc[c]);
    free(f->xyb[c]);
  }
  free(f->raw_quant_field);
  free(f->ac_strategy);
  free(f->ytox_map);
  free(f->ytob_map);
  if (f->group_tokens)
    for (size_t i = 0; i < f->xsize_groups * f->ysize_groups; i++) free(f->group_tokens[i]);
  free(f->group_tokens);
  free(f->group_token_bytes);
  free(f->quant_field);
  free(f->masking);
  free(f->entropy8);
  memset(f, 0, sizeof *f);
}
