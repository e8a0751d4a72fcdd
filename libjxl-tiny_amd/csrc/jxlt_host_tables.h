// Host-side construction of the constant tables the kernels read
// (jxlt_dev::DeviceTables) and of the frame geometry.
#ifndef JXLT_HOST_TABLES_H_
#define JXLT_HOST_TABLES_H_

#include <math.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>

#include "jxlt_device_common.h"
#include "jxlt_tables.h"

namespace jxlt_dev {

// ceil(log2(x)) for x >= 1 (common.h: CeilLog2Nonzero)
inline int HostCeilLog2Nonzero(uint32_t x) {
  int fl = 0;
  while ((x >> (fl + 1)) != 0) fl++;
  return (x & (x - 1)) == 0 ? fl : fl + 1;
}

// `scale` is DistanceParams.scale (enc_frame.cc:129); the per-quant reciprocal
// float(1.0 / (scale * q)) is evaluated in double exactly as
// QuantizeRoundtripYBlockAC does (enc_group.cc:289).
inline void BuildDeviceTables(float scale, DeviceTables* t) {
  memset(t, 0, sizeof(*t));
  for (int i = 0; i < 576; i++) {
    memcpy(&t->weights[i], &JXLT_kQuantWeightBits[i], 4);
    t->inv_weights[i] = static_cast<float>(1.0 / t->weights[i]);  // quant_weights.cc:144-146
  }
  for (int n = 0; n < 9; n++) {
    t->table_offset[n] = JXLT_kQuantTableOffset[n];
    if ((int)JXLT_kQuantTableOffset[n] != jxlt_dev::quant_table_offset(n)) {
      fprintf(stderr, "jxlt: quant table layout changed; update quant_table_offset()\n");
      abort();
    }
    for (int b = 0; b < JXLT_kQuantTableLLF[n]; b++) t->inv_weights[JXLT_kQuantTableOffset[n] + b] = 0.0f;
  }
  // the kernels' transforms are unnormalised (jxlt_device_common.h: kDct8Norm / kDct16Norm): exact power-of-two scaling
  for (int i = 0; i < 576; i++) t->inv_weights[i] *= i < 192 ? 1.0f / kDct8Norm : 1.0f / kDct16Norm;
  t->inv_qac[0] = 0.0f;
  for (int q = 1; q < 256; q++) t->inv_qac[q] = static_cast<float>(1.0 / (scale * q));
  memcpy(t->coeff_order, JXLT_kCoeffOrder, sizeof(t->coeff_order));
  memcpy(t->freq_context, JXLT_kCoeffFreqContext, sizeof(t->freq_context));
  memcpy(t->nnz_context, JXLT_kCoeffNumNonzeroContext, sizeof(t->nnz_context));
  memcpy(t->block_context_map, JXLT_kBlockContextMap, sizeof(t->block_context_map));
  memcpy(t->ac_context_map, JXLT_kACContextMap, sizeof(t->ac_context_map));
  memcpy(t->gradient_lut, JXLT_kGradientContextLut, sizeof(t->gradient_lut));
  for (int i = 0; i < 1024; i++) t->sqrt_lut[i] = sqrtf((float)i);  // IEEE: correctly rounded
  for (int n = 0; n < 132; n++) {
    const int nbits = HostCeilLog2Nonzero((uint32_t)n + 1) + 1;
    t->zeros_cost[n] = 7.565053364251793f * (float)(HostCeilLog2Nonzero((uint32_t)nbits + 17) + nbits);
  }
  // quantisation in scan order: the constants of scan position p, per position class
  for (int cls = 0; cls < 3; cls++) {
    const bool two_block = cls != 0;
    for (int p = 0; p < 64; p++) {
      const int n = JXLT_kCoeffOrder[cls * 64 + p];  // natural coefficient index (0..63 / 0..127)
      const int r = n >> 3;
      const int quad = two_block ? ((r >= 8 ? 2 : 0) | (r & 1)) : ((r >= 4 ? 2 : 0) | ((n & 7) >= 4 ? 1 : 0));
      for (int c = 0; c < 3; c++) {
        t->scan_consts[cls][c][p] = t->inv_weights[quant_table_offset((two_block ? 3 : 0) + c) + n];
        t->scan_consts[cls][4 + c][p] = quant_zeroing_threshold(c, two_block, quad);
      }
      t->scan_consts[cls][3][p] = t->weights[quant_table_offset(two_block ? 4 : 1) + n] * (two_block ? kDct16Norm : kDct8Norm);
      // staging slot of coefficient (row r, column l) within its block: l * 8 + (r & 7), second block: bit 6
      t->scan_slot[cls][p] = static_cast<uint8_t>((n & 64) | ((n & 7) << 3) | ((n >> 3) & 7));
    }
  }
}

// TileArgs::mul8x8 / bias8x8 / mul16x8 / cost_of_1 from A->distance and A->strategy_distance: the very float
// expressions of enc_ac_strategy.cc (:93-96, :178-185, :203), one IEEE operation each (this header is compiled with
// -ffp-contract=off wherever it is used).
inline void SetStrategyScalars(TileArgs* A) {
  const float k8x8mul1 = (float)(-0.55 * 0.75f);
  const float k8x8mul2 = 1.0735757687292623f * 0.75f;
  const float k8x8base = (float)1.4;
  volatile float den8 = A->strategy_distance + k8x8base;
  volatile float quot8 = k8x8mul1 / den8;
  A->mul8x8 = k8x8mul2 + quot8;
  A->bias8x8 = 3.0f * A->mul8x8;
  const float k8X16mul1 = (float)-0.55, k8X16mul2 = (float)0.9019587899705066, k8X16base = (float)1.6;
  volatile float den16 = A->strategy_distance + k8X16base;
  volatile float quot16 = k8X16mul1 / den16;
  A->mul16x8 = k8X16mul2 + quot16;
  volatile float third = A->distance * (1.0f / 3);
  const float slope = third < 1.0f ? third : 1.0f;
  volatile float scaled = slope * 8.8703248061477744f;
  A->cost_of_1 = 1 + scaled;
}

inline FrameGeom MakeGeom(size_t xsize, size_t ysize) {
  FrameGeom g;
  g.xsize = static_cast<int>(xsize);
  g.ysize = static_cast<int>(ysize);
  g.xsize_blocks = static_cast<int>((xsize + 7) / 8);
  g.ysize_blocks = static_cast<int>((ysize + 7) / 8);
  g.xsize_tiles = static_cast<int>((xsize + 63) / 64);
  g.ysize_tiles = static_cast<int>((ysize + 63) / 64);
  g.xsize_groups = static_cast<int>((xsize + 255) / 256);
  g.ysize_groups = static_cast<int>((ysize + 255) / 256);
  return g;
}

inline float XQmMultiplier(uint32_t x_qm_scale) {  // pow(1.25f, x_qm_scale - 2.0f), exact
  float m = 1.0f;
  for (uint32_t i = 2; i < x_qm_scale; i++) m *= 1.25f;
  return m;
}

}  // namespace jxlt_dev

// tile_kernel arguments for rows [y0, y0 + rows) of a frame as a frame of its own (y0 a multiple of 256 and the
// rows inside ONE row of DC groups -- or y0 a multiple of 2048 and any number of rows: then every block / tile /
// group / DC-group index of the slab is the frame's index minus a constant): the kernel never looks across a group boundary, so only the base pointers move.  `pitch` = A.pitch.
namespace jxlt_dev {
// slab: its number within the frame (every slab counts and lists its root-table overflows for itself)
inline TileArgs SlabTileArgs(const TileArgs& A, size_t y0, size_t rows, ptrdiff_t pitch, size_t slab) {
  TileArgs S = A;
  S.lut_overflow = A.lut_overflow + slab;
  S.overflow_tiles = A.overflow_tiles + (y0 / 64) * (size_t)A.g.xsize_tiles;
  S.g = MakeGeom((size_t)A.g.xsize, rows);
  const size_t xb = (size_t)A.g.xsize_blocks;
  const size_t b0 = (y0 / 8) * xb;  // first block of the slab
  for (int c = 0; c < 3; c++) {
    S.planes[c] = A.planes[c] + (ptrdiff_t)y0 * pitch;
    S.quant_dc[c] = A.quant_dc[c] + b0;
    S.nzgrid[c] = A.nzgrid[c] + b0;
    if (A.dbg_xyb[c]) S.dbg_xyb[c] = A.dbg_xyb[c] + y0 * xb * 8;
  }
  S.raw_quant = A.raw_quant + b0;
  S.strategy = A.strategy + b0;
  S.blk_nz = A.blk_nz + 3 * b0;
  S.blk_nscan = A.blk_nscan + 3 * b0;
  S.blk_nzmask = A.blk_nzmask + 6 * b0;
  S.coef_scan = A.coef_scan + 3 * 64 * b0;
  S.ytox = A.ytox + (y0 / 64) * (size_t)A.g.xsize_tiles;
  S.ytob = A.ytob + (y0 / 64) * (size_t)A.g.xsize_tiles;
  S.group_ntok = A.group_ntok + (y0 / 256) * (size_t)A.g.xsize_groups;
  S.dc_nac = A.dc_nac + (y0 / 2048) * (((size_t)A.g.xsize + 2047) / 2048);
  if (A.dbg_qf) S.dbg_qf = A.dbg_qf + b0;
  if (A.dbg_mask) S.dbg_mask = A.dbg_mask + b0;
  if (A.dbg_ent8) S.dbg_ent8 = A.dbg_ent8 + (y0 / 16) * (xb / 2 + 1) * 8;
  return S;
}
}  // namespace jxlt_dev

#endif  // JXLT_HOST_TABLES_H_
