// jxlt_tile_kernel.h -- tile12_kernel: one 768-thread workgroup per 64x64 tile, pixels to scan-ordered
// quantised coefficients and side-band grids (enc_frame.cc:597-683 + enc_group.cc:304-443).
// Part of jxlt_device.h (include that one).
#ifndef JXLT_TILE_KERNEL_H_
#define JXLT_TILE_KERNEL_H_

#include "jxlt_device_common.h"

namespace jxlt_dev {

// ---------------------------------------------------------------------------
// Tile kernel
// ---------------------------------------------------------------------------

// Twelve waves per tile (768 threads).  Octets 0-31 hold the DCT8 coefficients of TWO blocks each, octets 32-95 one
// two-block candidate each (the 8-wave kernel of rounds 1-2, removed in round 4, kept a block AND a candidate per
// octet): half the coefficient registers per thread, 12 waves per workgroup, two workgroups per CU =
// 6 waves per SIMD instead of 4 when the kernel fits 80 vector registers.
constexpr int kTile12Threads = 768;
// 12 waves: LDS copy of the per-scan-position tables of the quantisation phase (DeviceTables::scan_consts, scan_slot,
// inv_qac), in the term area behind the staging of the selected transforms' coefficients
constexpr int kP8ScanWords = 3 * 7 * 64 + 3 * 64 / 4;  // scan_consts + scan_slot
constexpr int kP8TabOffset = 13568;                    // floats from the start of the term area (54 272 B)
// ... and in front of it the entropy estimate's table DeviceTables::zeros_cost (behind the 16 x 768 floats in which
// the estimates park the B coefficients)
constexpr int kZerosCostOffset = 16 * kTile12Threads;
static_assert(kZerosCostOffset + kZerosCostEntries <= kP8TabOffset, "zeros_cost lies between the parks and the P8 tables");
constexpr int kHalo = 5;                 // AQ: +-4 px window, +-1 px Laplacian tap
constexpr int kXYPitch = 64 + 2 * kHalo + 1;  // 75 floats (odd: conflict-free columns)
constexpr int kBPitch = 65;
constexpr int kPrePitch = 19;
constexpr int kCflTermFloats = 64 * 64 * 4;  // LDS floats overlaid by the CfL terms
// float stride between the blocks of the coefficient staging area (3 x 64 values each): 200 = 8 (mod 64), so
// the eight 32-byte runs the octets of a wave store at a time land in different banks
constexpr int kStageStrideF = 200;

struct alignas(256) TileShared {
  // The tables and the per-block state come FIRST: an LDS address below 64 KB folds into the 16-bit offset field
  // of the instruction that uses it.  Behind the 64 KB of planes / terms the entropy estimate paid two address
  // instructions per coefficient (an add for the root table's base, an or for the weight's row).
  float inv_w[576];
  float sqrt_lut[1024];  // sqrtf of the quantised magnitudes below kSqrtLutSize (<= 1024; test builds use fewer)
  float aq[64];            // quant field (tile-local 8x8)
  float mask[64];
  float cfl_sum[4];        // ca_x, cb_x, ca_b, cb_b
  int cmap[2];             // ytox, ytob
  uint8_t raw_quant[64];
  uint8_t strat[64];
  uint32_t ntok;
  uint32_t nfirst;
  uint32_t overflow;  // a quantised magnitude of this tile did not fit the root table
  // (the planes, and with them the transpose areas behind them, start on a multiple of 256 bytes: an area's chunk
  // numbers are address bits, octet_transpose)
  uint32_t pad_to_256[3 + 20];
  float x[64 * kXYPitch];
  float y[64 * kXYPitch];
  float b[64 * kBPitch];
  float rowsum[16 * 72];   // AQ: per 4-row band, per column
  float pre_erosion[16 * kPrePitch];
  float erosion[16 * 16];
  float cfl_pad[kCflTermFloats - (2 * 64 * kXYPitch + 64 * kBPitch + 16 * 72 + 16 * kPrePitch + 16 * 16)];
  // ^ x..cfl_pad (64 KB) are overlaid by the chroma-from-luma terms once every pixel
  //   read is done: 64 blocks x 64 coefficients x (a_x, b_x, a_b, b_b).
  // rowsum..transpose_pad: during the transforms (the AQ buffers are dead by then) the candidate octets'
  // transpose areas, 64 x kTransposePitch floats, behind them the fourth pair wave's half-size rows
  // (kHalfTransposeWaveFloats; the other three pair waves' are in sqrt_lut) and at the very end p4_sums.
  // (+ 192: 17 x 128 floats -- with the 8 x 128 of sqrt_lut the 25 x 128 in which the chain waves of the 12-wave
  // kernel park coefficients during the chains)
  // (+ 64: its last 256 floats -- behind the 64 transpose areas -- hold the four sums per block that P4 leaves for the
  // wave that finishes the quant field, "p4_sums"; until round 4 those sat in sqrt_lut, which is the pair octets'
  // transpose scratch now)
  float transpose_pad[64 * 72 - (kCflTermFloats - (2 * 64 * kXYPitch + 64 * kBPitch)) + 192 + 64];
  static constexpr int kScratchFloats = 16 * 72 + 16 * kPrePitch + 16 * 16 +
                                        (kCflTermFloats - (2 * 64 * kXYPitch + 64 * kBPitch + 16 * 72 + 16 * kPrePitch + 16 * 16)) +
                                        (64 * 72 - (kCflTermFloats - (2 * 64 * kXYPitch + 64 * kBPitch)) + 192 + 64);
  // (its first 128 floats hold the candidate entropies of the 2x2 cells during the strategy search, "ent8")
};
static_assert(offsetof(TileShared, x) % 256 == 0 && offsetof(TileShared, rowsum) % 256 == 0 &&
                  offsetof(TileShared, sqrt_lut) % 256 == 0,
              "the term area is accessed in 16-byte chunks, the transpose areas are 256-byte aligned");
static_assert(TileShared::kScratchFloats >= 64 * kTransposePitch + kHalfTransposeWaveFloats + 256,
              "candidate areas + one pair wave's rows + p4_sums");
static_assert(sizeof(TileShared) <= 81920, "two workgroups per CU");
// After the last pixel read the XYB planes are dead and are reused: chroma-from-luma terms, the parked
// DCT8 coefficients of the entropy estimate, then the staging area of the selected transforms' coefficients
// (64 blocks x 3 channels x 64 floats).


// Per-octet entropy estimate of one transform (enc_ac_strategy.cc:51-146).
// cy/cx/cb: the lane's rows of the Y/X/B coefficients; NR rows (8 or 16).
// kLut: the roots come from the LDS table S.sqrt_lut (a multiply, a convert, a mask and an LDS
// read instead of v_sqrt + the exact-rounding fix-up, ~40 cycles); *qmax then receives the
// largest magnitude seen, and the caller redoes the estimate with kLut = false if it is beyond
// the table (quantised coefficients >= 1024: practically never, but results must not depend on it).
// kParkedB: the B coefficients are not in registers but in LDS, row r at cb[r * kParkStride] (the 12-wave
// kernel parks them there for the duration of the estimates: 32 instead of 48 coefficient registers).
template <int NR, bool kLut, bool kParkedB = false, int kParkStride = 1>
JXLT_DI float estimate_entropy(const float* cx, const float* cy, const float* cb, const float* inv_x,
                               const float* inv_y, const float* inv_b, int l, float quant,
                               float masking, float cmap_x, float cmap_b, float cost_of_1,
                               const float* sqrt_lut, const float* zeros_cost, float* qmax) {
  const float num_blocks = (float)(NR / 8);
  const float kInfoLossMultiplier = 138.0f;
  const float kInfoLossMultiplier2 = (float)50.46839691767866;
  const float kCost2 = 4.4628149885273363f;
  const float kCostDelta = 5.3359184934516337f;
  // (cost_of_1 = 1 + min(1, distance / 3) * 8.87...: wave-uniform, from the host -- TileArgs::cost_of_1)
  float entropy = 0.0f;
  float info_loss = 0.0f, info_loss2 = 0.0f;
  uint32_t qbits = 0;  // OR of the offset words before masking (a v_or is cheaper than a v_max)
  float lut_step = __uint_as_float(4u);  // 2^-147: q * lut_step has the bit pattern 4 * q
  JXLT_LAUNDER_VGPR(lut_step);           // (in a vector register, not an SGPR or a literal: see below)
  (void)lut_step;
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
      const float in = (kParkedB && c == 2) ? cin[r * kParkStride] : cin[r];
      const float im = inv[r * 8 + l];
      // (skipping the subtraction of cy * 0 for the Y channel saves two instructions per coefficient on paper; the
      // register allocator then spills: 90 VGPRs in rounds 1 and 3, 388-408 B of scratch in round 5 -- also with
      // scheduling fences every 2 / 4 / 8 rows of the Y pass, with the rows of that pass chained through empty asm
      // statements, and with a copy of cy[r] in place of the two instructions)
      const float val = (in - cy[r] * cmap_factor) * (im * quant);
      const float rval = rintf(val);
      const float diff = fabsf(val - rval);
      info_loss = info_loss + diff;
      info_loss2 = fma32(diff, diff, info_loss2);
      const float q = fabsf(rval);
      entropy_v = fma32(clamp01(q - 1.0f), kCost2, entropy_v);  // + (q >= 1.5 ? kCost2 : 0)
      float root;
      if (kLut) {
        // The byte offset 4 * q as the bit pattern of the DENORMAL q * 2^-147 (= 4 q units of 2^-149; the kernels run
        // with denormals on, hipcc's default): ONE multiplication, no float -> int conversion (a 4-cycle instruction)
        // and no mask.  q >= kSqrtLutSize reads beyond the table -- other LDS words of the tile or, beyond the
        // workgroup's allocation, the zero the hardware returns for such a read; q >= 2^21, infinities and NaNs give
        // patterns with exponent bits set -- and every such tile is redone by the caller (the OR below keeps the bits
        // for its test).  The multiplier sits in a vector register (a VOP3 instruction takes no literal on gfx950, and
        // one with an SGPR source costs four cycles instead of two).  (Rounds 2-4: the pattern of 4 q + 2^23, a
        // multiply-add and a mask.)
        const uint32_t off_raw = __float_as_uint(q * lut_step);
        root = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sqrt_lut) + JXLT_LUT_WRAP(off_raw));
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
    // The two sums of the channel in ONE butterfly (octet_sum_pair): the lower half of the octet gets the entropy's,
    // the upper half the non-zero count's.  The upper half turns its count into
    // kZerosMul * (CeilLog2Nonzero(nbits + 17) + nbits) with nbits = CeilLog2Nonzero(num_nzeros + 1) + 1 (:133-139)
    // by a look-up in DeviceTables::zeros_cost (its LDS copy): the count -- an integer below 129 held in a float -- times
    // 2^-147 has the bit pattern 4 * count, the table's byte offset (as for the roots above); the lower half, whose
    // "count" is an entropy, reads some aligned word there that nobody uses; the cost then moves down to the lower
    // half, where `entropy` is kept (its value is used from lane 0 alone).
    const float sums = octet_sum_pair(entropy_v, nzeros_v, l);
    const uint32_t zoff = __float_as_uint(sums * lut_step) & 0x3FCu;
    const float zcost = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(zeros_cost) + zoff);
    entropy += sums;
    entropy += octet_upper_to_lower(zcost);
  }
  // (the two information-loss sums likewise: the upper half takes the root, the lower half the weighted sum)
  const float losses = octet_sum_pair(info_loss, info_loss2, l);
  const float infoloss = losses;
  const float infoloss2 = octet_upper_to_lower(sqrtf(num_blocks * losses));
  const float info_loss_score = (kInfoLossMultiplier * infoloss + kInfoLossMultiplier2 * infoloss2);
  // every offset stayed inside the table <=> no bit above the offset field is set
  if (kLut) *qmax = (qbits & ~(uint32_t)(kSqrtLutSize * 4 - 1)) != 0u ? (float)kSqrtLutSize : 0.0f;
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
JXLT_DI void tile_kernel_body(const TileArgs& A, const int tile_id) {
  constexpr int kWaves = 12;
  constexpr int kThreads = kTile12Threads;
  __shared__ TileShared S;
  // (not const: the 12-wave variant re-derives the per-thread values from a "laundered" thread index at phase
  // boundaries, so that the values of one phase are not kept in registers across another phase's peak)
  int tid = (int)threadIdx.x;
  int l = tid & 7;    // lane within octet
  int oct = tid >> 3;  // octet index; for octets 0..63 == block index within tile
  const DeviceTables* T = A.tab;
  // (-DJXLT_TIMING_MARKS, tools/profile_phases.py: the per-phase clocks of thread 0 in the PRODUCTION variant too -- the
  // debug variant's phases are not the production kernel's: it spills and stores the intermediate planes)
#ifdef JXLT_TIMING_MARKS
  constexpr bool kMarks = true;
#else
  constexpr bool kMarks = kDebug;
#endif
  long long t_prev = (kMarks && A.dbg_phase) ? clock64() : 0;
  // Profiling builds (-DJXLT_PHASE_STOPS, tools/phase_pmc.py) can truncate the kernel after
  // phase i; the early exits perturb code generation, so production builds leave them out.
#ifdef JXLT_PHASE_STOPS
#define JXLT_STOP(i) if (((A.flags >> 8) & 15u) == (unsigned)(i) + 1u) return;
#else
#define JXLT_STOP(i)
#endif
  // -DJXLT_ASM_MARKERS (tools/asm_budget.py): an assembly comment where phase i ends, so that the kernel's
  // instruction stream can be split into phases (no instruction; production builds leave it out).
#ifdef JXLT_ASM_MARKERS
#define JXLT_ASM_PHASE_END(i) asm volatile("; JXLT_PHASE after_" #i)
#else
#define JXLT_ASM_PHASE_END(i)
#endif
#define JXLT_MARK(i)                                                        \
  if (kMarks && A.dbg_phase && tid == 0) {                                  \
    const long long t_now = clock64();                                      \
    atomicAdd(&A.dbg_phase[i + 16 * (blockIdx.x % kPhaseClockCopies)], (unsigned long long)(t_now - t_prev)); \
    t_prev = t_now;                                                         \
  }                                                                         \
  JXLT_ASM_PHASE_END(i);                                                    \
  JXLT_STOP(i)

  // ---- geometry (enc_frame.cc:716-751) ------------------------------------
  const int tx_img = tile_id % A.g.xsize_tiles, ty_img = tile_id / A.g.xsize_tiles;
  const int gx = tx_img >> 2;
  const int sx0 = gx * 256, sy0 = ty_img * 64;            // stripe origin (pixels)
  const int sw = imin(256, A.g.xsize - sx0), sh = imin(64, A.g.ysize - sy0);
  const int swp = (sw + 7) & ~7, shp = (sh + 7) & ~7;       // padded stripe size
  const int tbx0 = (tx_img & 3) * 8;                        // tile origin in stripe blocks
  const int nbx = imin(8, swp / 8 - tbx0), nby = shp / 8;   // tile size in blocks
  const int px0 = tbx0 * 8;                                 // tile origin in stripe pixels
  const int bx_img0 = gx * 32 + tbx0, by_img0 = ty_img * 8; // image-absolute block origin
  int obx = oct & 7, oby = oct >> 3;                        // octet's block in the tile
  bool blk_valid = obx < nbx && oby < nby;
  const uint32_t bstride = (uint32_t)A.g.xsize_blocks;

  // ---- P0: tables -> LDS; load + XYB (enc_frame.cc:597-617, enc_xyb.cc) -----
  // The table values are REQUESTED here (every lane, clamped indices: no branches) and stored to LDS behind the
  // pixel requests below, so that all of the tile's global loads are in flight together.  (Loops of "load, wait,
  // store to LDS" in front of the pixel loads cost six serial round trips to L2 per tile.)
  static_assert(kSqrtLutSize <= 512 || kSqrtLutSize == 1024, "table staging below");
  const float tab_inv0 = T->inv_weights[imin(tid, 575)];
  if (tid == 0) {
    S.ntok = 0;
    S.nfirst = 0;
    S.overflow = 0;
  }
  {
    // 12 waves: the 64 interior columns along the lanes, wave w takes the rows w, w + 12, ... (six of them for
    // waves 0-3, five for the others); the ten halo columns (X and Y only) as 6 rows x 10 columns per wave, one
    // pixel per thread.  All 21 loads of a thread are requested before the first use.
    constexpr int kWin = 64 + 2 * kHalo;
    const int c = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: the rows' offsets are scalar)
    const int xi = px0 + c;
    const bool xok_i = xi < swp && c < nbx * 8 + kHalo;
    const int xs_i = (xok_i ? imin(xi, sw - 1) : 0) * A.pix_stride;
    const int hr = (c * 205) >> 11, hc = c - 10 * hr;  // lane -> (row 0..6, halo column 0..9): c / 10, c % 10
    const int hy = 6 * w + hr;
    const int hcx = hc < kHalo ? hc : 64 + hc;          // LDS column of the halo pixel
    const int hx = px0 - kHalo + hcx;
    const bool hok = hr < 6 && hy < shp && hx >= 0 && hx < swp && hx < px0 + nbx * 8 + kHalo;
    const int xs_h = (hok ? imin(hx, sw - 1) : 0) * A.pix_stride;
    float pr[7], pg[7], pb[7];
    // The six interior rows of a wave are wave-uniform: their base addresses are computed on the scalar unit and the
    // lane adds a 32-bit byte offset (global_load with a scalar base -- no vector instruction per load; as one 64-bit
    // expression per load the compiler spent a 64-bit multiply-add, a shift and three 64-bit adds per row on it).
    const ptrdiff_t col_off = (ptrdiff_t)sx0 * A.pix_stride;
    const uint32_t byte_i = (uint32_t)xs_i * 4u;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const ptrdiff_t row_off = (ptrdiff_t)(sy0 + imin(w + 12 * k, sh - 1)) * A.pitch + col_off;  // (scalar)
      // (the row's base pointers are opaque to the compiler: it would otherwise add the lane's offset to the plane's
      // base first and the row's offset per load, a 64-bit vector add each)
      // (... and typed as pointers to GLOBAL memory: behind the opaque point the compiler no longer knows where they
      // came from and would use flat loads)
      JxltGlobalBytes r0 = (JxltGlobalBytes)(A.planes[0] + row_off);
      JxltGlobalBytes r1 = (JxltGlobalBytes)(A.planes[1] + row_off);
      JxltGlobalBytes r2 = (JxltGlobalBytes)(A.planes[2] + row_off);
      JXLT_LAUNDER_SGPR(r0);
      JXLT_LAUNDER_SGPR(r1);
      JXLT_LAUNDER_SGPR(r2);
      pr[k] = *(JxltGlobalFloats)(r0 + byte_i);
      pg[k] = *(JxltGlobalFloats)(r1 + byte_i);
      pb[k] = *(JxltGlobalFloats)(r2 + byte_i);
    }
    {  // (the halo pixel: its row depends on the lane)
      const ptrdiff_t off = (ptrdiff_t)(sy0 + imin(hy, sh - 1)) * A.pitch + col_off + xs_h;
      pr[6] = A.planes[0][off];
      pg[6] = A.planes[1][off];
      pb[6] = A.planes[2][off];
    }
    if (tid < 576) S.inv_w[tid] = tab_inv0;
    // (the root table is staged behind the chains of P5b: until then its place in LDS serves the chain waves)
    if (A.byteswap) {  // big-endian PFM payload (BSwapFloat, read_pfm.cc:206)
#pragma unroll
      for (int k = 0; k < 7; k++) {
        pr[k] = __uint_as_float(__builtin_bswap32(__float_as_uint(pr[k])));
        pg[k] = __uint_as_float(__builtin_bswap32(__float_as_uint(pg[k])));
        pb[k] = __uint_as_float(__builtin_bswap32(__float_as_uint(pb[k])));
      }
    }
    // (one exec-mask region around the six rows -- the lane's column is in the window or not, whatever the row --
    // instead of one per row)
    if (xok_i) {
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int y = w + 12 * k;
      if (y >= 64 || y >= shp) break;  // (wave-uniform)
      float px_, py_, pb_;
      linear_to_xyb<true>(pr[k], pg[k], pb[k], &px_, &py_, &pb_);
      S.x[y * kXYPitch + kHalo + c] = px_;
      S.y[y * kXYPitch + kHalo + c] = py_;
      S.b[y * kBPitch + c] = pb_;
      if (kDebug && A.dbg_xyb[0] && c < nbx * 8) {
        const size_t d = (size_t)(by_img0 * 8 + y) * ((size_t)bstride * 8) + (size_t)(bx_img0 * 8 + c);
        A.dbg_xyb[0][d] = px_;
        A.dbg_xyb[1][d] = py_;
        A.dbg_xyb[2][d] = pb_;
      }
    }
    }
    if (hok) {
      float px_, py_, pb_ = 0.0f;
      linear_to_xyb<false>(pr[6], pg[6], pb[6], &px_, &py_, &pb_);
      S.x[hy * kXYPitch + hcx] = px_;
      S.y[hy * kXYPitch + hcx] = py_;
    }
    (void)kWin;
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
    // One pixel's term from its own value, the sum of its vertical neighbours and its horizontal neighbours.  `vec`:
    // the position is one of the reference's vector loop (its association order, its fused multiply-add); the other
    // positions take the scalar loop's.  Both are computed and selected: a branch per pixel would wait for the LDS
    // before and after each arm.
    auto pixel_term = [&](bool vec, float in, float du, float in_l, float in_r, float ix, float dux, float ix_l,
                          float ix_r) {
      const float gammac = ratio_of_derivatives(in + match_gamma_offset, false);
      // the vector loop adds (right + left) + vertical, the scalar loop (vertical + left) + right: ONE sum of the
      // form (a + left) + c whose outer operands are selected (two selects instead of four additions and a select)
      const float base = 0.25f * (((vec ? in_r : du) + in_l) + (vec ? du : in_r));
      const float base_x = 0.25f * (((vec ? ix_r : dux) + ix_l) + (vec ? dux : ix_r));
      float diff = gammac * (in - base);
      diff = diff * diff;
      float diff_x = gammac * (ix - base_x);
      diff_x = diff_x * diff_x;
      const float fused = fma32(kXMul, diff_x, diff), unfused = diff + kXMul * diff_x;
      return masking_sqrt(vec ? fused : unfused, sqrt_mul);
    };
    // One band column (band q = rows 4q .. 4q + 3, stripe column x): its four pixel terms summed, then P2's
    // 4-column average inside the quad of lanes that holds the four columns of one average.
    auto band_column = [&](int q, int x, bool first_of_quad) {
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
      if (first_of_quad) S.pre_erosion[q * kPrePitch + ((x - aq_x0) >> 2)] = s4 * 0.25f;
    };
    // (Spreading the last, partly filled pass over all threads row by row changes nothing: the other resident
    // workgroup takes the issue slots the idle waves leave.  The 4-column groups that lie entirely in the vector
    // loop's range on a path of their own -- no second association order, no selects, 15-16 of a tile's 17-18 groups --
    // was built in round 4 and lost: 3860 against 3886 VALU per wave, 9.41 against 9.33 M cycles)
    for (int i = tid; i < nbands * aq_w; i += kThreads) {
      const int q = i / aq_w, x = aq_x0 + i % aq_w;
      band_column(q, x, (i & 3) == 0);
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
  // The per-block part behind the four sums (:52-75, :146-285, :518-534): from the erosion value and the sums to the
  // quant field.  The octets leave
  // their sums in LDS and ONE wave does it for the tile's 64 blocks, a lane each, while the terms of
  // chroma-from-luma are published -- an eighth of the instructions, and off the path of the waves that do P4.
  auto block_quant_field = [&](float erosion, float hf, float red, float blue, float gam) {
    const float kRedRampLength = (float)0.019421555948474039;
    const float kBlueRampLength = (float)0.086890611400405895;
    float out_val = compute_mask(erosion);
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
    // PerBlockModulations tail (:249-285)
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
    return fast_pow2f(out_val * 1.442695041f) * mul + add;
  };
  // ... and what is stored per block (raw quant :518-534, the initial strategy)
  auto store_block_quant = [&](int b, float qf) {
    S.aq[b] = qf;
    int v = (int)(qf * A.inv_scale + 0.5f);
    v = v < 1 ? 1 : v > 255 ? 255 : v;
    S.raw_quant[b] = (uint8_t)v;
    S.strat[b] = 1;  // DCT8, first block (FillDCT8)
    if (kDebug && A.dbg_qf) {
      const uint32_t pos = (uint32_t)(by_img0 + (b >> 3)) * bstride + (uint32_t)(bx_img0 + (b & 7));
      A.dbg_qf[pos] = qf;
      A.dbg_mask[pos] = S.mask[b];
    }
  };
  // [block][hf, red, blue, gamma]: behind the transpose areas (read before the chain waves park there)
  float* const p4_sums = &S.transpose_pad[sizeof(S.transpose_pad) / sizeof(float) - 256];
  static_assert(offsetof(TileShared, transpose_pad) + sizeof(S.transpose_pad) - 1024 >=
                    offsetof(TileShared, rowsum) + (64 * kTransposePitch + kHalfTransposeWaveFloats) * sizeof(float),
                "p4_sums lies behind the transpose areas");
  if (tid < 512) {  // (octets 0..63 = waves 0-7)
    const int bxp = px0 + obx * 8, byp = oby * 8;  // block origin (stripe pixels)
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
      {
        if (l == 0) {
          float4 sums;
          sums.x = hf; sums.y = red; sums.z = blue; sums.w = gam;
          *reinterpret_cast<float4*>(p4_sums + oct * 4) = sums;
        }
      }
    }
  }
  // (no barrier here: what P4 writes -- the final quant field, raw_quant, the initial strategies -- is first read
  // behind later barriers, the transforms below read the pixel planes only, and their LDS scratch lies over the
  // adaptive-quantisation buffers, which nobody reads behind the barrier in front of P4.  The waves that have no
  // block in P4 -- 8 to 11 of the 12-wave kernel -- start their transforms at once.)
  JXLT_MARK(3);

  // ---- P6a: candidate two-block transforms (enc_ac_strategy.cc:62-66) -------
  // Waves 0-3 take the 32 DCT16X8 candidates, waves 4-7 the 32 DCT8X16 candidates.
  // Done before chroma-from-luma so that afterwards no pixel is needed any more; the
  // coefficients stay in registers for the entropy estimate and for P8.
  // 12 waves: octets 0..31 (waves 0-3) are PAIR octets -- the DCT8 of the blocks (pbx, pby0) and (pbx, pby0 + 1) --
  // and octets 32..95 the 64 candidate octets (candidate octet co = oct - 32).
  // (the roles are wave-uniform, and the compiler is told so: scalar branches, one set of coefficient registers)
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool pair_role = wave_u < 4;
  const bool cand_role = wave_u >= 4;
  const bool search = (A.flags & 1u) == 0;
  const bool is_tall = wave_u < 8;  // DCT16X8 (16 rows x 8 cols)
  int co, pbx, pby0, cand, cell, ccx, ccy, cbx, cby;
  bool pair_valid0, pair_valid1, cell_valid;
  float* tsc;
  auto derive_roles = [&]() {
    co = oct >= 32 ? oct - 32 : 0;  // candidate octet index (0..63)
    pbx = oct & 7, pby0 = (oct >> 3) * 2;          // pair octet's blocks (12 waves)
    pair_valid0 = pair_role && pbx < nbx && pby0 < nby, pair_valid1 = pair_role && pbx < nbx && pby0 + 1 < nby;
    tsc = &S.rowsum[0] + co * kTransposePitch;  // candidate octet's transpose area (AQ buffers are dead)
    cand = co & 31;                       // candidate index within its type
    cell = cand >> 1;                     // 2x2 cell index (4x4 cells per tile)
    ccx = (cell & 3) * 2, ccy = (cell >> 2) * 2;  // cell origin (tile blocks)
    cbx = is_tall ? ccx + (cand & 1) : ccx;       // candidate's first block
    cby = is_tall ? ccy : ccy + (cand & 1);
    cell_valid = cand_role && search && (ccx + 1 < nbx) && (ccy + 1 < nby);
  };
  derive_roles();
  // Re-derives every per-thread value from the thread index behind a compiler barrier (12 waves only).
  auto new_phase = [&]() {
    {
      JXLT_LAUNDER_VGPR(tid);
      l = tid & 7;
      oct = tid >> 3;
      obx = oct & 7, oby = oct >> 3;
      blk_valid = obx < nbx && oby < nby;
      derive_roles();
    }
  };
  new_phase();
  float c16x[16], c16y[16], c16b[16];
  // The transforms below run under conditions (block in the frame, search on) and so does every later use of their
  // results.  On the other path the registers are "defined" without an instruction (JXLT_DEFINE_VGPR): left undefined,
  // the compiler zero-initialised all 48 in front of the branch, 32 v_mov per thread and tile.
  auto leave_undefined = [&](int r0, int r1) {
#pragma unroll
    for (int r = r0; r < r1; r++) {
      JXLT_DEFINE_VGPR(c16x[r]);
      JXLT_DEFINE_VGPR(c16y[r]);
      JXLT_DEFINE_VGPR(c16b[r]);
    }
  };
  // 12 waves: a pair octet's two DCT8 live in the registers a candidate octet uses for its transform (rows 0-7:
  // first block, rows 8-15: second block), so that a thread needs ONE set of 48 coefficient registers.
  float* const c8x = c16x;
  float* const c8y = c16y;
  float* const c8b = c16b;
  float* const d8x = c16x + 8;
  float* const d8y = c16y + 8;
  float* const d8b = c16b + 8;
  if (pair_role) {
    // The pair octets have no transpose area of their own (the 64 there are belong to the candidate octets, which
    // transpose at the same time): they go through half-size rows per wave, octet_transpose_half -- three waves' in
    // the place of the root table (free until the chains), the fourth's behind the candidate areas.  (Rounds 3-4:
    // register butterflies.)
    constexpr bool kPairMode = true;
    static_assert(sizeof(S.sqrt_lut) >= 3 * kHalfTransposeWaveFloats * sizeof(float), "three pair waves' rows");
    // (waves 0-2: in the root table's place; wave 3: behind the candidate octets' areas)
    float* const psc = wave_u < 3 ? &S.sqrt_lut[0] + wave_u * kHalfTransposeWaveFloats : &S.rowsum[0] + 64 * kTransposePitch;
    const int po = oct & 7;
    const float* pxp = &S.x[(pby0 * 8) * kXYPitch + pbx * 8 + kHalo];
    const float* pyp = &S.y[(pby0 * 8) * kXYPitch + pbx * 8 + kHalo];
    const float* pbp = &S.b[(pby0 * 8) * kBPitch + pbx * 8];
    if (pair_valid0) {
      block_dct8x8<kPairMode>(pxp, kXYPitch, l, psc, c8x, po);
      JXLT_SCHED_FENCE();
      block_dct8x8<kPairMode>(pyp, kXYPitch, l, psc, c8y, po);
      JXLT_SCHED_FENCE();
      block_dct8x8<kPairMode>(pbp, kBPitch, l, psc, c8b, po);
      JXLT_SCHED_FENCE();
    } else {
      leave_undefined(0, 8);
    }
    // (blocks outside the frame: the registers stay undefined, every later use is under the same condition)
    if (pair_valid1) {
      block_dct8x8<kPairMode>(pxp + 8 * kXYPitch, kXYPitch, l, psc, d8x, po);
      JXLT_SCHED_FENCE();
      block_dct8x8<kPairMode>(pyp + 8 * kXYPitch, kXYPitch, l, psc, d8y, po);
      JXLT_SCHED_FENCE();
      block_dct8x8<kPairMode>(pbp + 8 * kBPitch, kBPitch, l, psc, d8b, po);
      JXLT_SCHED_FENCE();
    } else {
      leave_undefined(8, 16);
    }
  } else if (cell_valid) {
    const float* pxp = &S.x[(cby * 8) * kXYPitch + cbx * 8 + kHalo];
    const float* pyp = &S.y[(cby * 8) * kXYPitch + cbx * 8 + kHalo];
    const float* pbp = &S.b[(cby * 8) * kBPitch + cbx * 8];
    // (scheduling fences: interleaving the three independent transforms would triple the
    // live registers and spill)
    if (is_tall) {
      block_dct16x8<true>(pxp, kXYPitch, l, tsc, c16x, oct & 7);
      JXLT_SCHED_FENCE();
      block_dct16x8<true>(pyp, kXYPitch, l, tsc, c16y, oct & 7);
      JXLT_SCHED_FENCE();
      block_dct16x8<true>(pbp, kBPitch, l, tsc, c16b, oct & 7);
    } else {
      block_dct8x16<true>(pxp, kXYPitch, l, tsc, c16x, oct & 7);
      JXLT_SCHED_FENCE();
      block_dct8x16<true>(pyp, kXYPitch, l, tsc, c16y, oct & 7);
      JXLT_SCHED_FENCE();
      block_dct8x16<true>(pbp, kBPitch, l, tsc, c16b, oct & 7);
    }
    JXLT_SCHED_FENCE();
  } else {
    leave_undefined(0, 16);
  }
  JXLT_MARK(4);
  // ---- P5: chroma-from-luma (the DCT8 of every block is in the pair octets' registers) -------
  // (enc_chroma_from_luma.cc:40-131)
  __syncthreads();  // all pixel reads done: the planes are dead from here on
  new_phase();
  {
    // the blocks' quant fields from the sums P4 left (see block_quant_field): wave 11, a lane per block, beside the
    // publishing of the terms; first read behind the chains
    if (wave_u == 11) {
      const int b = tid & 63;
      if ((b & 7) < nbx && (b >> 3) < nby) {
        const float4 sums = *reinterpret_cast<const float4*>(p4_sums + b * 4);
        store_block_quant(b, block_quant_field(S.aq[b], sums.x, sums.y, sums.z, sums.w));
      }
    }
  }
  JXLT_MARK(5);
  // ---- P5b: chroma-from-luma (enc_chroma_from_luma.cc:40-131) ----------------
  {
    // Every octet publishes the terms of its block, a = m/84 and b = base*m - s with
    // m = Y*qm, s = C*qm (:49-53,117-120); then four sequential per-lane fma chains
    // (ca = sum a*a, cb = sum a*b, for X and for B) run over the blocks in raster order.
    // Term layout (round 6): [block][q][lane l][channel X, B][4 floats], q = 0: a of rows 0-3, 1: a of rows 4-7,
    // 2: b of rows 0-3, 3: b of rows 4-7 -- a chain lane reads FIRST factors (always a) from q = 0, 1 and SECOND
    // factors from q = 0, 1 (the sums of a * a: wave 0) or q = 2, 3 (the sums of a * b: wave 1): the two chain waves
    // run the same instructions on two base addresses, no selects.  The sixteen chain lanes of a 16-lane row
    // (channel, l) read l * 8 + channel * 4 + {0..3} + a multiple of 64: all 64 banks once, and q is an immediate
    // offset -- no swizzle, one address per factor.  (Rounds 2-5: chunks of (a, b) pairs of two rows, XOR-swizzled;
    // the second factor was selected in place, eight v_cndmask per round of four blocks on the chain waves.)
    float* terms = &S.x[0];
    const float* qm_x = S.inv_w + 0;    // InvMatrix(DCT, 0)
    const float* qm_b = S.inv_w + 128;  // InvMatrix(DCT, 2)
    const int nblk = nbx * nby;
    // The chains run in ROUNDS of four blocks (below): a tile at the frame's edge whose block count is not a multiple
    // of four gets up to three blocks of zero terms -- fma(0, 0, acc) leaves an accumulator as it is (it is never -0:
    // it starts at +0 and x + y is -0 only for two -0).
    const int nblk_pad = (nblk + 3) & ~3;
    const float kInvColorFactor = 1.0f / 84;
    // (the terms of raster block `rb` of the tile from the lane's rows of its DCT8 coefficients)
    auto publish = [&](int rb, const float* vx, const float* vy, const float* vb) {
      float* dst = &terms[rb * 256 + l * 8];
#pragma unroll
      for (int r = 0; r < 8; r += 4) {
        float4 ax4, bx4, ab4, bb4;
#pragma unroll
        for (int h = 0; h < 4; h++) {
          const int rr = r + h;
          const bool dc = (rr == 0 && l == 0);  // block_*[0] = 0 (:109-111)
          const float by_ = dc ? 0.0f : vy[rr], bx_ = dc ? 0.0f : vx[rr], bb_ = dc ? 0.0f : vb[rr];
          const float qx = qm_x[rr * 8 + l], qb = qm_b[rr * 8 + l];
          const float m_x = by_ * qx, s_x = bx_ * qx, m_b = by_ * qb, s_b = bb_ * qb;
          const float ax = kInvColorFactor * m_x, bx2 = 0.0f * m_x - s_x;
          const float ab = kInvColorFactor * m_b, bb2 = 1.0f * m_b - s_b;
          (&ax4.x)[h] = ax;
          (&bx4.x)[h] = bx2;
          (&ab4.x)[h] = ab;
          (&bb4.x)[h] = bb2;
        }
        const int half = r >> 2;
        *(float4*)&dst[(0 + half) * 64] = ax4;
        *(float4*)&dst[(2 + half) * 64] = bx4;
        *(float4*)&dst[(0 + half) * 64 + 4] = ab4;
        *(float4*)&dst[(2 + half) * 64 + 4] = bb4;
      }
    };
    {
      if (pair_valid0) publish(pby0 * nbx + pbx, c8x, c8y, c8b);
      if (pair_valid1) publish((pby0 + 1) * nbx + pbx, d8x, d8y, d8b);
      if (nblk_pad != nblk) {  // (wave-uniform; tiles at the frame's edge only)
        if (tid < (nblk_pad - nblk) * 256) terms[nblk * 256 + tid] = 0.0f;
      }
    }
    __syncthreads();
    // Chain lanes: wave 0 lanes 0-15 run ca (X: 0-7, B: 8-15), wave 1 lanes 0-15 run cb.
    // The 512 fused multiply-adds of a chain are strictly sequential, so these two waves are
    // the critical path of the workgroup: they run at raised issue priority, and the reads
    // of block blk + 1 are issued before the arithmetic of block blk.
    float acc = 0.0f;
    // (12 waves: the wave index as a scalar)
    const int cw = __builtin_amdgcn_readfirstlane(tid >> 6), cl = tid & 63;
    // 12 waves: the waves that wait for the chains fetch the root table meanwhile (two entries per thread); it goes
    // to LDS behind the chains, where the chain waves' parked coefficients were.
    // (round 6: NOT initialised -- the chain waves would carry six zeros through their loop, the kernel's register peak;
    // they "define" them behind it, JXLT_DEFINE_VGPR, and never store them)
    float late_root0, late_root1;
    // ... and what the scan-order quantisation (P8b) needs per lane: its constants per scan position, the staging slots
    // and the inverse quantiser steps -- 1648 words that every wave would otherwise fetch from global memory, 25 loads
    // per thread, at the start of that phase, with nothing to do meanwhile
    uint32_t late_p8[3];
    float late_zeros_cost;
    {
      if (cw >= 2) {
        late_zeros_cost = T->zeros_cost[imin(tid - 128, kZerosCostEntries - 1)];
        late_root0 = T->sqrt_lut[(tid - 128) & (kSqrtLutSize - 1)];
        late_root1 = T->sqrt_lut[(tid - 128 + 640) & (kSqrtLutSize - 1)];
        const uint32_t* const scan_words = reinterpret_cast<const uint32_t*>(&T->scan_consts[0][0][0]);
        const uint32_t* const qac_words = reinterpret_cast<const uint32_t*>(&T->inv_qac[0]);
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int w = tid - 128 + 640 * j;
          late_p8[j] = w < kP8ScanWords ? scan_words[w] : qac_words[imin(w - kP8ScanWords, 255)];
        }
      }
    }
    // The chains as a RELAY over the four 16-lane rows of the wave.  A row = the 16 chain lanes (X: 8, B: 8); row
    // k handles every fourth block, and the terms of the next four blocks are requested a whole round of four
    // blocks ahead of their use -- in registers the other rows' lanes have anyway.  The
    // sixteen accumulators travel from row to row (0 -> 1 -> 3 -> 2 -> 0) with one v_permlane16_swap /
    // v_permlane32_swap per block (tools/permlane_probe.hip).  With ping-pong buffers in sixteen lanes the terms
    // of block blk + 1 were requested only eight dependent multiply-adds before their use: less than an LDS round
    // trip, and the chains are the workgroup's critical path.
    // A lone wave issues an instruction every ~5 cycles whatever it is (tools/valu_issue_probe.hip), so
    // what the chain waves issue BESIDE the 512 dependent multiply-adds is what the phase costs (round 5: ~16
    // instructions per block).  Round 6: no select of the second factor (term layout above), no test per block (whole
    // rounds, zero terms behind the tile's last block), and the accumulators hop between TWO registers -- a swap
    // exchanges rows of two registers, so the row that receives them is the other register's: no copy per hop.
    const int relay_row = cl >> 4;
    const int relay_pos = relay_row == 0 ? 0 : relay_row == 1 ? 1 : relay_row == 3 ? 2 : 3;  // place in the relay
    if (cw < 2) {
      __builtin_amdgcn_s_setprio(3);
      // 12 waves: the chain waves hold 48 coefficients like every other wave, and the two term sets below are
      // 32 registers more: fifteen of the B coefficients wait in the one piece of LDS that is free right now
      // (the transposes' pad behind the term area)
      // rows 0..16 in the transposes' pad, rows 17..24 in the root table's place
      auto chain_park = [&](int r) -> float& {
        return r < 17 ? S.transpose_pad[r * 128 + tid] : S.sqrt_lut[(r - 17) * 128 + tid];
      };
      {
        static_assert(sizeof(S.transpose_pad) / 4 >= 17 * 128 && sizeof(S.sqrt_lut) / 4 >= 8 * 128, "chain park");
#pragma unroll
        for (int r = 0; r < 16; r++) chain_park(r) = c16b[r];
#pragma unroll
        for (int r = 8; r < 16; r++) chain_park(8 + r) = c16x[r];
        chain_park(24) = c16y[15];
      }
      const int ch = (cl >> 3) & 1;  // 0: X, 1: B
      const float* const first_factors = terms + l * 8 + ch * 4;
      const float* const second_factors = first_factors + cw * 128;  // (a again, or b)
      // Two register sets, used in turn by ROUNDS of four blocks (one per row): at the start of a round every row
      // requests the block it will handle in the NEXT round -- one wave-wide set of four 16-byte loads, a whole
      // round (32 dependent multiply-adds and four hops) ahead of its use.
      JxltFloat4 ta[4], tb[4];  // [0], [1]: first factors of rows 0-3, 4-7; [2], [3]: second factors
      // (issued HERE, a round ahead of their use: as plain loads the compiler sinks them to their first use behind the
      // loop's exit test and every round waits for the LDS -- JXLT_LDS_LOAD4_NOW, jxlt_device_common.h)
      // (addresses in integers that move on by a round -- 4 KB -- per request: two additions.  No clamp at the end of
      // the tile: what the last requests fetch is never used -- with an even number of rounds one round beyond the
      // terms, with an odd number two, i.e. never more than 4 KB beyond the term area's 64 blocks --, and those 4 KB lie
      // inside the workgroup's LDS: the transposes' pad, static_assert below.)
      auto a1 = JXLT_LDS_ADDRESS(first_factors) + (decltype(JXLT_LDS_ADDRESS(first_factors)))(relay_pos << 10);
      auto a2 = JXLT_LDS_ADDRESS(second_factors) + (decltype(a1))(relay_pos << 10);
      static_assert(offsetof(TileShared, x) + (size_t)kCflTermFloats * 4 + 4 * 1024 <= sizeof(TileShared),
                    "the chains request up to a round beyond the term area");
      auto request = [&](JxltFloat4* t) {
        JXLT_LDS_LOAD4_NOW_AT(t[0], a1, 0);
        JXLT_LDS_LOAD4_NOW_AT(t[2], a2, 0);
        JXLT_LDS_LOAD4_NOW_AT(t[1], a1, 256);
        JXLT_LDS_LOAD4_NOW_AT(t[3], a2, 256);
        a1 += 4096;
        a2 += 4096;
      };
      float acc2 = 0.0f;  // (the accumulators' other register)
      // every lane runs the eight steps; only the row that holds the accumulators has meaningful ones
      auto block_steps = [&](float& a, const JxltFloat4* t) {
        a = fma32(t[0].x, t[2].x, a);
        a = fma32(t[0].y, t[2].y, a);
        a = fma32(t[0].z, t[2].z, a);
        a = fma32(t[0].w, t[2].w, a);
        a = fma32(t[1].x, t[3].x, a);
        a = fma32(t[1].y, t[3].y, a);
        a = fma32(t[1].z, t[3].z, a);
        a = fma32(t[1].w, t[3].w, a);
      };
      // (a swap trades the odd rows / the upper half of its first operand for the even rows / the lower half of its
      // second one; both results are kept, so the instruction works in place)
      auto hop16 = [&](float& vdst, float& src0) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(vdst), __float_as_uint(src0), false, false);
        vdst = __uint_as_float(r[0]);
        src0 = __uint_as_float(r[1]);
      };
      auto hop32 = [&](float& vdst, float& src0) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(vdst), __float_as_uint(src0), false, false);
        vdst = __uint_as_float(r[0]);
        src0 = __uint_as_float(r[1]);
      };
      // one round: the set `t` has arrived when all but the four loads of the OTHER set, requested behind it, have
      auto round4 = [&](JxltFloat4* t) {
        JXLT_LDS_WAIT4(4, t[0], t[1], t[2], t[3]);
        block_steps(acc, t);   // row 0
        hop16(acc2, acc);      // acc row 0 -> acc2 row 1
        block_steps(acc2, t);  // row 1
        hop32(acc, acc2);      // acc2 row 1 -> acc row 3
        block_steps(acc, t);   // row 3
        hop16(acc, acc2);      // acc row 3 -> acc2 row 2
        block_steps(acc2, t);  // row 2
        hop32(acc2, acc);      // acc2 row 2 -> acc row 0
      };
      request(ta);
#pragma clang loop unroll(disable)
      for (int blk = 0; blk < nblk_pad; blk += 8) {
        request(tb);
        round4(ta);
        request(ta);
        if (blk + 4 < nblk_pad) round4(tb);  // (wave-uniform: an odd number of rounds ends here)
      }
      // (whatever was requested last is not used; it has arrived before its registers serve anything else)
      JXLT_LDS_DRAIN4(ta[0], ta[1], ta[2], ta[3]);
      JXLT_LDS_DRAIN4(tb[0], tb[1], tb[2], tb[3]);
      JXLT_DEFINE_VGPR(late_root0);
      JXLT_DEFINE_VGPR(late_root1);
      JXLT_DEFINE_VGPR(late_zeros_cost);
#pragma unroll
      for (int j = 0; j < 3; j++) JXLT_DEFINE_VGPR(late_p8[j]);
      {
        JXLT_COMPILER_FENCE();
#pragma unroll
        for (int r = 0; r < 16; r++) c16b[r] = chain_park(r);
#pragma unroll
        for (int r = 8; r < 16; r++) c16x[r] = chain_park(8 + r);
        c16y[15] = chain_park(24);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    const int final_pos = 0;  // (whole rounds: the accumulators end where they started, in row 0)
    const bool chain_lane = cw < 2 && relay_pos == final_pos;
    const int chain_ch = (cl >> 3) & 1;
    const float total = octet_sum(acc);
    // cfl_sum: ca_x, cb_x, ca_b, cb_b
    if (chain_lane && l == 0) S.cfl_sum[chain_ch * 2 + cw] = total;
    __syncthreads();
    {
      if (cw >= 2) {  // (visible to P6b behind the barrier below)
        if (tid - 128 < kSqrtLutSize) S.sqrt_lut[tid - 128] = late_root0;
        if (tid - 128 + 640 < kSqrtLutSize) S.sqrt_lut[tid - 128 + 640] = late_root1;
        if (tid - 128 < kZerosCostEntries) (&S.x[0] + kZerosCostOffset)[tid - 128] = late_zeros_cost;
        uint32_t* const p8_tab = reinterpret_cast<uint32_t*>(&S.x[0]) + kP8TabOffset;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const int w = tid - 128 + 640 * j;
          if (w < kP8ScanWords + 256) p8_tab[w] = late_p8[j];
        }
      }
    }
    if (tid < 2) {  // FindBestMultiplier tail (:56-61)
      const float kDistanceMultiplierAC = 1e-3f;
      const float num = (float)(nblk * 64);
      float xq = -S.cfl_sum[tid * 2 + 1] / (S.cfl_sum[tid * 2] + num * kDistanceMultiplierAC * 0.5f);
      xq = fmaxf(-128.0f, fminf(127.0f, roundf(xq)));
      S.cmap[tid] = (int)xq;
    }
  }
  __syncthreads();
  new_phase();  // (what the phases below derive from the thread index does not stay in registers across the chains)
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
  // (the LDS copy of DeviceTables::zeros_cost: in the dead term area, between the parked coefficients and the
  // quantisation phase's tables; staged behind the chains with the root table)
  const float* const zeros_cost = &S.x[0] + kZerosCostOffset;
  if (search) {
    // DCT8 estimate of block (bx, by) of the tile from the lane's rows of its coefficients
    auto estimate8 = [&](int bx, int by, const float* vx, const float* vy, const float* vb) {
      const int bi = by * 8 + bx;
      float qmax8 = 0.0f;
      const float e = estimate_entropy<8, kLutRoots, true, kThreads>(
          vx, vy, vb, S.inv_w + 0, S.inv_w + 64, S.inv_w + 128, l, fmaxf(0.0f, S.aq[bi]), fmaxf(0.0f, S.mask[bi]),
          cmap_x, cmap_b, A.cost_of_1, S.sqrt_lut, zeros_cost, &qmax8);
      qmax = fmaxf(qmax, qmax8);
      // mul8x8 = k8x8mul2 + k8x8mul1 / (strategy_distance + k8x8base), 3 * mul8x8 (:178-185, :203): from the host
      float e8 = A.bias8x8;
      e8 += A.mul8x8 * e;
      if (l == 0) S.transpose_pad[((by >> 1) * 4 + (bx >> 1)) * 8 + (by & 1) * 2 + (bx & 1)] = e8;
    };
    // 12 waves: the B coefficients (rows 0..15: a candidate's, or rows 0..7 / 8..15: the two blocks of a pair
    // octet) wait in the dead term area during the estimates, which read them from there; back in registers
    // for P8a afterwards.
    float* park = &S.x[0] + tid;
    {
#pragma unroll
      for (int r = 0; r < 16; r++) park[r * kThreads] = c16b[r];
      JXLT_SCHED_FENCE();
      if (pair_valid0) estimate8(pbx, pby0, c8x, c8y, park);
      JXLT_SCHED_FENCE();
      if (pair_valid1) estimate8(pbx, pby0 + 1, d8x, d8y, park + 8 * kThreads);
    }
    JXLT_SCHED_FENCE();
    if (cell_valid) {
      const int o2 = is_tall ? 8 : 1;  // second covered block in the 8x8 tile grid
      const int bi = cby * 8 + cbx;
      const float quant = fmaxf(fmaxf(0.0f, S.aq[bi]), S.aq[bi + o2]);
      const float masking = fmaxf(fmaxf(0.0f, S.mask[bi]), S.mask[bi + o2]);
      const int toff = is_tall ? 3 : 6;
      float qmax16 = 0.0f;
      const float e = estimate_entropy<16, kLutRoots, true, kThreads>(
          c16x, c16y, park, S.inv_w + quant_table_offset(toff), S.inv_w + quant_table_offset(toff + 1),
          S.inv_w + quant_table_offset(toff + 2), l, quant, masking, cmap_x, cmap_b, A.cost_of_1, S.sqrt_lut,
          zeros_cost, &qmax16);
      qmax = fmaxf(qmax, qmax16);
      if (l == 0) S.transpose_pad[cell * 8 + (is_tall ? 4 : 6) + (cand & 1)] = A.mul16x8 * e;
    }
    JXLT_SCHED_FENCE();
    {
      JXLT_COMPILER_FENCE();
#pragma unroll
      for (int r = 0; r < 16; r++) c16b[r] = park[r * kThreads];
    }
  }
  // A magnitude beyond the root table invalidates this tile's estimates: at its end the tile files its index
  // instead of adding its counts, and the launch behind this one (tile*_kernel_redo) does the listed tiles again
  // with computed roots, overwriting everything this pass writes.  (The tile runs to its end all the same: an
  // early exit here changes the control flow of the whole kernel, and with it the register allocation -- the
  // 12-wave kernel was 12 % slower with it -- for the sake of tiles that practically never occur.)
  if (kLutRoots && (qmax >= (float)kSqrtLutSize || (A.flags & 0x1000u) != 0)) S.overflow = 1u;  // (0x1000: test hook)
  __syncthreads();
  JXLT_MARK(6);
  // ---- P7: decision (:213-237) + AdjustQuantField (:240-266) ------------------
  // Wave 0 alone, with the other eleven waiting: what counts is the length of this part (as 16 threads walking the
  // four blocks of their cell through LDS, with a barrier of its own, it was ~900 cycles per tile; round 5).  The
  // decision by a lane per 16x16 cell, then AdjustQuantField by a lane per BLOCK: a block of a two-block transform
  // takes the larger of its own and its partner's quantiser -- all lanes read, then all write.
  uint32_t p7_strat = 0u, p7_quant = 0u;
  if (tid < 64) {
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
      }
    }
    JXLT_WAVE_SYNC();  // (the wave's LDS operations execute in order)
    const bool in_frame = (tid & 7) < nbx && (tid >> 3) < nby;
    p7_strat = in_frame ? S.strat[tid] : 0u;
    p7_quant = S.raw_quant[tid];
    const uint32_t code = p7_strat >> 1;  // 0: one block; 1: partner below / above; 2: beside
    if (code != 0) {
      const int o2 = code == 1 ? 8 : 1;
      const uint32_t other = S.raw_quant[(p7_strat & 1u) ? tid + o2 : tid - o2];
      p7_quant = other > p7_quant ? other : p7_quant;
    }
    JXLT_WAVE_SYNC();  // (every lane has read)
    if (code != 0) S.raw_quant[tid] = (uint8_t)p7_quant;
  }
  __syncthreads();
  if (tid < 64) {  // (wave 0: a lane per block)
    const bool in_frame = (tid & 7) < nbx && (tid >> 3) < nby;
    const uint32_t st = p7_strat;
    if (in_frame) {
      const uint32_t pos = (uint32_t)(by_img0 + (tid >> 3)) * bstride + (uint32_t)(bx_img0 + (tid & 7));
      A.strategy[pos] = (uint8_t)st;
      A.raw_quant[pos] = (uint8_t)p7_quant;
    }
    // the tile's first blocks: one ballot instead of an LDS atomic per first block
    const unsigned long long firsts = __ballot((st & 1u) != 0);
    if (tid == 0) S.nfirst = (uint32_t)__popcll(firsts);
  }
  // All pixel reads were done before P5b (the transforms live in registers): from here on the
  // XYB planes are reused as the quantised-coefficient staging area.  No barrier is needed
  // between the stores above (they read S.strat / S.raw_quant, final since the barrier before
  // them) and P8; S.nfirst is read after later barriers.
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
    // (two 16-byte stores per channel and block half.  The eight values have to be copied into aligned register
    // quadruples first -- 76 v_mov in the kernel --; four ds_write2_b32 per call, which take their values from any two
    // registers, were tried in round 5: no copies, but dword pairs at a lane stride of eight floats meet 16 to a bank --
    // 9.35 against 8.87 M cycles per 16384^2 launch.)
    auto put8 = [&](float* d, const float* v) {
      float4 lo, hi;
      lo.x = v[0]; lo.y = v[1]; lo.z = v[2]; lo.w = v[3];
      hi.x = v[4]; hi.y = v[5]; hi.z = v[6]; hi.w = v[7];
      *reinterpret_cast<float4*>(d) = lo;
      *reinterpret_cast<float4*>(d + 4) = hi;
    };
    {
      const int b0 = pby0 * 8 + pbx;
      if (pair_valid0 && S.strat[b0] == 1) {  // the pair octet's blocks that stayed DCT8
        float* d = stagef + b0 * kStageStrideF + l * 8;
        put8(d, c8x);
        put8(d + 64, c8y);
        put8(d + 128, c8b);
      }
      if (pair_valid1 && S.strat[b0 + 8] == 1) {
        float* d = stagef + (b0 + 8) * kStageStrideF + l * 8;
        put8(d, d8x);
        put8(d + 64, d8y);
        put8(d + 128, d8b);
      }
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
    // (12 waves: from the copy in LDS that the waves waiting for the chains made, see late_p8)
    const float* const p8_consts = &S.x[0] + kP8TabOffset;
    const uint8_t* const p8_slots =
        reinterpret_cast<const uint8_t*>(&S.x[0] + kP8TabOffset + 3 * 7 * 64);
    const float* const p8_inv_qac = &S.x[0] + kP8TabOffset + kP8ScanWords;
    auto consts_of = [&](int cls) {
      LaneConsts k;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        k.inv[c] = p8_consts[(cls * 7 + c) * 64 + lane];
        k.thr[c] = p8_consts[(cls * 7 + 4 + c) * 64 + lane];
      }
      k.ydq = p8_consts[(cls * 7 + 3) * 64 + lane];
      return k;
    };
    const LaneConsts k8 = consts_of(0), k16a = consts_of(1), k16b = consts_of(2);
    const int slot8 = p8_slots[lane], slot16a = p8_slots[64 + lane], slot16b = p8_slots[128 + lane];
    // lane b knows block b of the tile; the first blocks of the tile's transforms as a mask
    const bool lane_blk_valid = (lane & 7) < nbx && (lane >> 3) < nby;
    const int strat_of_lane = lane_blk_valid ? (int)S.strat[lane] : 0;
    const int quant_of_lane = (int)S.raw_quant[lane];
    const float inv_qac_of_lane = p8_inv_qac[quant_of_lane];  // (one vector load: no scalar load per transform)
    // (and its place in the frame: the transform's comes by v_readlane instead of six scalar instructions)
    const uint32_t pos_of_lane = (uint32_t)(by_img0 + (lane >> 3)) * bstride + (uint32_t)(bx_img0 + (lane & 7));
    // (transform number t, in raster order of the first blocks, goes to wave t mod 8: lane b finds its block's
    // number as the count of first blocks below it, and the wave's own blocks come out of one more ballot)
    const unsigned long long firsts = __ballot(strat_of_lane & 1);
    const int rank_of_lane =
        (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(firsts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)firsts, 0u));
    unsigned long long todo = __ballot((strat_of_lane & 1) != 0 && (rank_of_lane % kWaves) == wave);
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
    // Guard against coefficients the token format cannot carry (PackSigned(q) must fit 16 bits: -32768 <= q <= 32767;
    // the reference only asserts it in debug builds, enc_bit_writer.cc:120, and the coefficient store below keeps 16
    // bits).  The table-root pass only keeps the largest magnitude per lane -- three v_max3_f32 per transform; a
    // quantised value is never a NaN (the zeroing threshold's comparison turns one into 0), an infinity stays one --:
    // below 32768 nothing is wrong; anything else makes the tile "suspect", and a suspect tile is filed like one that
    // overflowed the root table and done again by tile*_kernel_redo, which tests every value exactly (-32768 is
    // legal) and counts the tile in A.unsupported if one fails.
    float q_largest = 0.0f;
    bool q_bad = false;
    // Two sets of staged coefficients used in turn (the transform being worked on / the next one, requested before
    // the work starts): as ONE set that is copied at the top of the loop the copies were six v_mov per transform.
    float set_a[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}}, set_b[2][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    int next_b = todo != 0 ? (int)__builtin_ctzll(todo) : -1, next_st = 0;
    if (next_b >= 0) {
      next_st = scalar_lane(strat_of_lane, next_b) >> 1;
      fetch(next_b, next_st, set_a);
    }
    // one transform: `in` holds its staged coefficients, the next transform's are requested into `next_v`
    auto one_transform = [&](float (*in)[3], float (*next_v)[3]) {
      const int b = __builtin_amdgcn_readfirstlane(next_b), st = __builtin_amdgcn_readfirstlane(next_st);
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
      const uint32_t pos0 = (uint32_t)scalar_lane((int)pos_of_lane, b);
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
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          if (!kLutRoots) q_bad = q_bad || !(fabsf(quant[h][c] + 0.5f) <= 32767.5f);  // (exact: q is an integer)
        }
      if (kLutRoots) {  // (three v_max3_f32 with |.| modifiers for the six values)
        q_largest = fmaxf(fmaxf(q_largest, fabsf(quant[0][0])), fabsf(quant[0][1]));
        q_largest = fmaxf(fmaxf(q_largest, fabsf(quant[0][2])), fabsf(quant[1][0]));
        q_largest = fmaxf(fmaxf(q_largest, fabsf(quant[1][1])), fabsf(quant[1][2]));
      }
      const int t = wave + kWaves * ntrans;  // the transform's number in the tile
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
        // (scan position covered - 1 of the mask: what the entry's first coefficient token takes in place of "previous
        // coefficient nonzero", enc_group.cc:476-480 -- token_kernel's emit reads it like any other position)
        // (1 << (covered - 1) is `covered` itself)
        m0[c] |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane(nzeros <= 4 * covered ? covered : 0);
      }
      // (the tokeniser takes "nonzeros still to come" and "previous coefficient nonzero" from these masks.  They are
      // wave-uniform -- ballots, in scalar registers -- and leave by SCALAR stores: six s_store_dwordx2 to 48 contiguous
      // bytes.  As vector stores by lanes 0 and 1 they cost eight vector instructions per channel: the scalar values
      // moved to vector registers, selected per lane, a 64-bit address.)
      // (scalar address arithmetic is not free here: in this phase every wave of the CU runs on the ONE scalar unit the
      // four SIMDs share -- 462 scalar against 368 vector instructions per wave, round 5's per-phase counters --, so:
      // one base per block with the channels at constant offsets, 32-bit products where the frame limit allows them
      // (2^25 blocks * 48 bytes of masks), and nothing at all for the second block of a one-block transform.)
      {
        unsigned long long* masks = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(A.blk_nzmask) + pos0 * 48u);
        JXLT_LAUNDER_SGPR(masks);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          JXLT_SCALAR_STORE64(masks, 2 * c, m0[c]);
          JXLT_SCALAR_STORE64(masks, 2 * c + 1, m1[c]);
        }
      }
      {
        // only scan positions below nscan (= up to the last nonzero) are ever read again
        // (the 192 slots of a block: a wave-uniform base the compiler cannot look through + the channel's constant
        // offset + the lane's 16-bit slot -- a store with a scalar base, no 64-bit vector add per store)
        JxltGlobalBytes out0 = (JxltGlobalBytes)(A.coef_scan + (size_t)pos0 * 192);
        JXLT_LAUNDER_SGPR(out0);
        const uint32_t slot = (uint32_t)lane * 2u;
#pragma unroll
        for (int c = 0; c < 3; c++)
          if (lane < nscan[c]) *(JxltGlobalShorts)(out0 + (c * 128 + slot)) = (int16_t)(int)quant[0][c];
        if (two) {
          JxltGlobalBytes out1 = (JxltGlobalBytes)(A.coef_scan + (size_t)pos1 * 192);
          JXLT_LAUNDER_SGPR(out1);
#pragma unroll
          for (int c = 0; c < 3; c++)
            if (64 + lane < nscan[c]) *(JxltGlobalShorts)(out1 + (c * 128 + slot)) = (int16_t)(int)quant[1][c];
        }
      }
      file_int(1, t, nz_packed);
      file_int(2, t, nscan_packed);
      ntrans++;
    };
    while (next_b >= 0) {
      one_transform(set_a, set_b);
      if (next_b < 0) break;
      one_transform(set_b, set_a);
    }
    JXLT_SCALAR_STORES_DONE();  // (the masks above: out of the scalar data cache)
    if (lane == 0 && wave_tokens) atomicAdd(&S.ntok, wave_tokens);
    if (kLutRoots) {
      if (__ballot(!(q_largest < 32768.0f)) != 0 && lane == 0) S.overflow = 1u;  // (suspect: see q_largest)
    } else {
      if (__ballot(q_bad) != 0 && lane == 0) atomicAdd(A.unsupported, 1u);
    }
  }
  __syncthreads();
  // the tile's transforms side by side, one per lane: DC of the covered blocks (:392-443) and the per-block outputs.
  // Waves 0, 1, 2 take the channels x, y, b (the chroma waves work y's quantised DC out for themselves, :415-425):
  // this part runs with the other waves idle and the workgroup's LDS held, so its LENGTH counts, not its instruction
  // count -- as one wave doing the three channels in turn (until round 5) the kernel took 1 % longer; with the token
  // count of the tile added to it, 2 % more.
  if (tid < 192) {
    const int lane = tid & 63;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's channel
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
      // (the staged coefficients are unnormalised: kDct8Norm / kDct16Norm times the reference's)
      const float unnorm = two ? 1.0f / kDct16Norm : 1.0f / kDct8Norm;
      auto dc_pair = [&](int ch, float* d_a, float* d_b) {
        const float c0 = dc_stage[(lane * 3 + ch) * 2] * unnorm, c1 = dc_stage[(lane * 3 + ch) * 2 + 1] * unnorm;
        const float b0 = c0 * 1.0f * 1.0f, b1 = c1 * 1.0f * kScale1;
        *d_a = two ? b0 + b1 : c0;
        *d_b = two ? b0 - b1 : 0.0f;
      };
      // y: every wave (the chroma DC is coded relative to it)
      float y_a, y_b;
      dc_pair(1, &y_a, &y_b);
      const float inv_factor_y = kInvDCQuant[1] * A.scale_dc;
      float fdc_a = roundf(inv_factor_y * y_a), fdc_b = roundf(inv_factor_y * y_b);
      if (c != 1) {
        const int16_t dcy_a = (int16_t)fdc_a, dcy_b = (int16_t)fdc_b;
        float d_a, d_b;
        dc_pair(c, &d_a, &d_b);
        const float inv_factor = (c == 0 ? kInvDCQuant[0] : kInvDCQuant[2]) * A.scale_dc;
        const float cfl_factor = c == 0 ? 0.0f : kInvDCQuant[2] * (1.0f / kInvDCQuant[1]);
        fdc_a = roundf(d_a * inv_factor - dcy_a * cfl_factor);
        fdc_b = roundf(d_b * inv_factor - dcy_b * cfl_factor);
      }
      const int16_t qdc_a = (int16_t)fdc_a, qdc_b = (int16_t)fdc_b;
      // (a quantised DC value beyond int16 -- DCGroupData's type -- or not a number: the frame is refused)
      const bool dc_bad = !(fdc_a >= -32768.0f && fdc_a <= 32767.0f && fdc_b >= -32768.0f && fdc_b <= 32767.0f);
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
      if (dc_bad) atomicAdd(A.unsupported, 1u);
    }
  }
  JXLT_MARK(9);
  if (tid == 0) {
    if (kLutRoots && S.overflow != 0) {
      A.overflow_tiles[atomicAdd(&A.lut_overflow[0], 1u)] = (uint32_t)tile_id;
    } else {
      const int group = (ty_img >> 2) * A.g.xsize_groups + gx;
      atomicAdd(&A.group_ntok[group], S.ntok);
      const int dcg = (ty_img >> 5) * ((A.g.xsize + 2047) / 2048) + (tx_img >> 5);
      atomicAdd(&A.dc_nac[dcg], S.nfirst);
    }
  }
#undef JXLT_MARK
#undef JXLT_ASM_PHASE_END
#undef JXLT_STOP
#undef SX
#undef SY
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed placement, used for speed only), and each XCD
// has its own L2.  Every XCD gets one contiguous raster range of tiles so that horizontally adjacent tiles --
// which share the +-5 px halo columns and the partially covered 128-byte lines -- are served by the same L2.
JXLT_DI int xcd_ordered_tile(const TileArgs& A) {
  const int n = A.g.xsize_tiles * A.g.ysize_tiles;
  const int b = (int)blockIdx.x, xcd = b & 7, idx = b >> 3;
  const int q = n >> 3, r = n & 7;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}
// The tiles the launch in front filed in A.overflow_tiles, with every root computed (a fixed, small grid:
// usually the list is empty and every workgroup leaves at once).
JXLT_DI void tile_redo_body(const TileArgs& A) {
  const uint32_t n = A.lut_overflow[0];
  for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
    tile_kernel_body<false, true>(A, (int)A.overflow_tiles[e]);
    __syncthreads();  // (the next tile starts by writing the shared state this one has just read)
  }
}
constexpr int kRedoGrid = 512;

// tile12_kernel: roots of the entropy estimate from the LDS table; tile12_kernel_redo: every root computed -- the
// same results, for the tiles in which tile12_kernel met a quantised magnitude beyond the table.  (Rounds 1-2 ran
// this tile with 8 waves -- a block AND a candidate per octet, 125 registers, 4 waves per SIMD; the variant was kept
// selectable through round 3 as the reference of the A/B in DESIGN.md 4.1.0 and went in round 4.)
__global__ void __launch_bounds__(kTile12Threads, 6) tile12_kernel(const TileArgs A) {
  tile_kernel_body<true, false>(A, xcd_ordered_tile(A));
}
__global__ void __launch_bounds__(kTile12Threads, 6) tile12_kernel_debug(const TileArgs A) {
  tile_kernel_body<true, true>(A, xcd_ordered_tile(A));
}
__global__ void __launch_bounds__(kTile12Threads, 6) tile12_kernel_redo(const TileArgs A) { tile_redo_body(A); }

}  // namespace jxlt_dev

#endif  // JXLT_TILE_KERNEL_H_
