/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * CPU restatement ("oracle") of the libjxl-tiny per-group encode hot path:
 *   CopyAndPadImage -> ToXYB -> ComputeAdaptiveQuantFieldTile -> ComputeCmapTile
 *   -> FindBest16x16Transform/AdjustQuantField -> WriteACGroup (DCT, quantise,
 *   DC, nzeros, raw 3-byte tokens).
 * Reference: /root/reference/encoder/enc_frame.cc:685-763 and the files it calls
 * (each function in jxl_tiny_oracle.c cites the file:line it follows).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product (libjxl-tiny_amd/) never links it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be built in this image (its Highway
 * dependency is an empty submodule and there is no network), so
 * this restatement is pinned only by (a) line-by-line citation and (b) the
 * size-only known answers recorded in SURVEY.md Appendix C from a one-off probe
 * build (see tests/test_oracle_known_answers.py).
 *
 * Canonical arithmetic model: Highway vectors of 8 float
 * lanes (AVX2-like), MulAdd/NegMulAdd fused (fmaf), every other operation a
 * single IEEE-754 binary32 operation with no contraction, SumOfLanes = halving
 * tree (+4, +2, +1), ApproximateReciprocal(x) = 1/x, Round = ties-to-even,
 * non-scalar branch of enc_adaptive_quantization.cc:230, strategy multipliers
 * computed from the call's distance.
 */
#ifndef JXL_TINY_ORACLE_H_
#define JXL_TINY_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enc_frame.cc:104-156 */
typedef struct {
  float distance;
  int32_t global_scale;
  int32_t quant_dc;
  float scale;
  float inv_scale;
  float scale_dc;
  uint32_t x_qm_scale;
  uint32_t epf_iters;
} orc_distance_params;

void orc_compute_distance_params(float distance, orc_distance_params* p);

/* Whole-image result of the hot path.  Block grids are xsize_blocks x
 * ysize_blocks (row-major, image-absolute block coordinates); tile grids are
 * xsize_tiles x ysize_tiles (64x64 tiles).  Token buffers are one per 256x256
 * group, in raster group order, holding the reference's raw 3-byte records
 * [pre-clustered context u8][value u16 LE] (enc_group.cc:468-470). */
typedef struct {
  size_t xsize, ysize;
  size_t xsize_blocks, ysize_blocks;
  size_t xsize_tiles, ysize_tiles;
  size_t xsize_groups, ysize_groups;
  int16_t* quant_dc[3];     /* dc_group_data.h:33 */
  uint8_t* raw_quant_field; /* dc_group_data.h:34 */
  uint8_t* ac_strategy;     /* (type<<1)|is_first, ac_strategy.h:151-165 */
  int8_t* ytox_map;         /* dc_group_data.h:36 */
  int8_t* ytob_map;
  uint8_t** group_tokens;   /* [num_groups] */
  size_t* group_token_bytes;
  /* optional intermediates (always filled): */
  float* xyb[3];       /* (xsize_blocks*8) x (ysize_blocks*8), XYB incl. padding */
  float* quant_field;  /* per block, float AQ field (tile-local aq_map) */
  float* masking;      /* per block */
  float* entropy8;     /* per 2x2 cell: 8 floats (e00,e01,e10,e11,16x8 L,R,8x16 T,B),
                          grid (xsize_blocks/2+1) x (ysize_blocks/2+1), NaN if not evaluated */
} orc_frame;

/* planes[c] + y*stride_floats addresses row y of channel c (linear sRGB).
 * force_dct8 != 0 mimics OPTIMIZE_BLOCK_SIZES 0 (config.h:12).
 * Returns 0 on success, nonzero for invalid arguments. */
int orc_encode_hot_path(const float* const planes[3], size_t stride_floats,
                        size_t xsize, size_t ysize, float distance,
                        int force_dct8, orc_frame* out);
/* The same with the frame's 256x256 groups -- the reference's independent units, enc_frame.cc:716-757 -- spread over
 * nthreads POSIX threads (the caller is one of them).  Same results as orc_encode_hot_path, which is this with one
 * thread (the reference's own behaviour: its ThreadPool is never used).  bench.py's cpu_baseline
 * legs use it.  keep_intermediates 0: the optional intermediates of orc_frame (xyb, quant_field, masking,
 * entropy8) stay NULL -- the reference does not keep them either. */
int orc_encode_hot_path_threads(const float* const planes[3], size_t stride_floats, size_t xsize, size_t ysize,
                                float distance, int force_dct8, int nthreads, int keep_intermediates, orc_frame* out);
void orc_frame_free(orc_frame* f);
/* Emulates a later EncodeFile call of a process whose first call used `first_call_distance`
 * (the reference's function-local static constants, enc_ac_strategy.cc:178-185); 0 = off. */
void orc_set_strategy_distance(float first_call_distance);

/* Stage-level entry points for unit parity tests. */
void orc_to_xyb(float* r, float* g, float* b, size_t n); /* in place */
void orc_dct8x8(const float* px, size_t stride, float* out64);
void orc_dct16x8(const float* px, size_t stride, float* out128);
void orc_dct8x16(const float* px, size_t stride, float* out128);
float orc_fast_log2f(float x);
float orc_fast_pow2f(float x);

/* ---- bitstream stage (oracle/jxl_tiny_bitstream_oracle.c) ------------------------------
 * Everything EncodeFile / EncodeFrame do after the pixel pipeline (enc_file.cc:55-105,
 * enc_frame.cc:287-595, 766-858 with enc_cluster.cc, enc_huffman_tree.cc,
 * enc_entropy_code.cc): DC-group tokenisation, code optimisation, section writing,
 * headers, TOC.  Independent of the product's host back-end. */
typedef struct {
  size_t xsize, ysize;
  const int16_t* quant_dc[3];      /* image-absolute block grids (pitch ceil(xsize / 8)) */
  const uint8_t* raw_quant_field;
  const uint8_t* ac_strategy;      /* (type << 1) | is_first */
  const int8_t* ytox_map;          /* image-absolute 64x64-tile grids (pitch ceil(xsize / 64)) */
  const int8_t* ytob_map;
  const uint8_t* const* group_tokens; /* raw 3-byte records per 256x256 group, raster order */
  const size_t* group_token_bytes;
} orc_bs_input;
/* The complete .jxl file.  reference_single_symbol: 1 = one bit per token of a single-symbol
 * prefix code as the reference writes it (undecodable there), 0 = zero bits.  *out is
 * malloc'ed (orc_bs_free).  Returns 0 on success. */
int orc_bs_encode_file(const orc_bs_input* in, float distance, int reference_single_symbol, uint8_t** out,
                       size_t* out_size);
/* Raw records of one DC-group section (WriteDCGroup with OPTIMIZE_CODE, enc_frame.cc:536-570). */
int orc_bs_dc_group_records(const orc_bs_input* in, size_t dc_group, uint8_t** out, size_t* out_size);
/* Code tables (depth << 16 | bits per [context][symbol]) from the [64][64] AC histograms
 * (per pre-clustered context) and DC histograms (45 contexts): what OptimizeSections
 * (enc_frame.cc:766-783) arrives at. */
void orc_bs_build_code_tables(const uint32_t* ac_hist, const uint32_t* dc_hist, int reference_single_symbol,
                              uint32_t* ac_table, uint32_t* dc_table);
void orc_bs_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* JXL_TINY_ORACLE_H_ */
