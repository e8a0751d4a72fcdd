// jxlt_device_common.h -- kernel argument blocks and __device__ helpers shared by the kernels of the
// JPEG XL tiny hot path: arithmetic primitives of the canonical 8-lane model, the register-held 1-D DCTs,
// the adaptive-quantisation helpers.  Part of jxlt_device.h (include that one).
#ifndef JXLT_DEVICE_COMMON_H_
#define JXLT_DEVICE_COMMON_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

namespace jxlt_dev {

// ---------------------------------------------------------------------------
// Kernel arguments
// ---------------------------------------------------------------------------

// Constant tables, resident in HBM (built on the host: jxlt_host_tables.h, jxlt_capi_encode.hip).
struct DeviceTables {
  float weights[576];      // dequant weights (quant_weights.cc:17-134)
  // float(1.0 / w), LLF zeroed (quant_weights.cc:144-153) -- TIMES the normalisation the kernels' transforms leave
  // out: tile_kernel keeps its coefficients unnormalised (kDct8Norm / kDct16Norm times the reference's, see
  // block_dct8x8), and every use of a coefficient is a product with one of these tables
  float inv_weights[576];
  float inv_qac[256];      // float(1.0 / (double)(scale * q)) (enc_group.cc:289)
  uint16_t table_offset[9];  // host copy of quant_table_offset() below (checked when the tables are built)
  uint8_t coeff_order[192];
  uint16_t freq_context[64];
  uint16_t nnz_context[64];
  uint8_t block_context_map[81];
  alignas(4) uint8_t ac_context_map[1980];  // (token_kernel copies it to LDS as 495 words)
  uint8_t gradient_lut[1024];  // enc_frame.cc:226-281
  float sqrt_lut[1024];        // sqrtf(i), correctly rounded (EstimateEntropy's cost of a coefficient)
  // What the quantisation needs to know of scan position p (tile_kernel quantises in scan order, lane = scan
  // position), per position class -- 0: DCT8, 1 / 2: first / second 64 positions of a two-block transform:
  // [0..2] InvMatrix of x, y, b at the position's coefficient (normalisation included, as in inv_weights), [3]
  // dequantisation weight of y times kDct8Norm / kDct16Norm (the dequantised y meets unnormalised coefficients),
  // [4..6] zeroing threshold of x, y, b (enc_group.cc:227-242); scan_slot: where the staging area keeps that
  // coefficient (bit 6: in the transform's second block).
  float scan_consts[3][7][64];
  uint8_t scan_slot[3][64];
  // EstimateEntropy's cost of coding the number of non-zeros of a channel (enc_ac_strategy.cc:133-139), as a function
  // of that number n = 0 .. 128: kZerosMul * (CeilLog2Nonzero(nbits + 17) + nbits), nbits = CeilLog2Nonzero(n + 1) + 1
  // -- twenty integer instructions per channel and estimate in the kernel until round 4, a table word now.
  float zeros_cost[132];
};
constexpr int kZerosCostEntries = 129;

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

// The kernels' block transforms leave out the reference's normalisation (1/8 per 8-point pass, 1/16 per 16-point
// pass, StoreToBlockAndScale, enc_transforms-inl.h:387-390): their coefficients are kDct8Norm (one-block) or
// kDct16Norm (two-block) times the reference's.  Powers of two commute exactly with every rounded operation, and a
// coefficient is only ever used in products with table values (InvMatrix in the entropy estimate, the chroma-from-luma
// terms and the quantisation; the dequantisation weight on the way back), so the factor lives in the host-built
// tables and costs nothing: 48 multiplications per thread less.
constexpr float kDct8Norm = 64.0f;
constexpr float kDct16Norm = 128.0f;

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
  // Wave-uniform factors of the strategy search, computed once on the host (jxlt_host_tables.h: SetStrategyScalars;
  // IEEE float arithmetic, what the kernels computed per estimate until round 4): mul8x8 / mul16x8 of
  // enc_ac_strategy.cc:178-185 from strategy_distance, 3 * mul8x8 (:203), cost_of_1 (:93-96) from distance
  float mul8x8, bias8x8, mul16x8, cost_of_1;
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
  unsigned long long* blk_nzmask;  // [block*3 + c][2]: which of the scan positions covered .. 127 are nonzero; bit
                                   // covered - 1: nzeros <= 4 * covered (the first coefficient token's "previous")
  int16_t* coef_scan;   // [block*3 + c][64] quantised coefficients in scan order
  uint32_t* group_ntok; // per group token count (atomic)
  uint32_t* dc_nac;     // per DC group: number of first blocks (atomic)
  uint32_t* lut_overflow;    // [1] number of tiles of this launch that met a quantised magnitude beyond the root table
  uint32_t* overflow_tiles;  // their indices (capacity: the launch's tiles): redone by tile*_kernel_redo
  // [1] (one word per frame) incremented for every tile / wave that met a value the format cannot carry: a quantised
  // AC coefficient whose token does not fit 16 bits (the reference asserts it in debug builds only,
  // enc_bit_writer.cc:120, and writes a broken stream otherwise) or a quantised DC value beyond int16 (DCGroupData's
  // type, dc_group_data.h:19-37).  The C ABI turns a non-zero count into JXLT_ERR_UNSUPPORTED.
  uint32_t* unsupported;
  // debug (may be null)
  float* dbg_xyb[3];
  float* dbg_qf;
  float* dbg_mask;
  float* dbg_ent8;
  unsigned long long* dbg_phase;  // [16] accumulated shader cycles per phase (thread 0 of each tile)
};

// The writing pass of the section packing runs as up to this many launches (jxlt_pack_kernels.h; the host's mail
// words are laid out for them).
constexpr int kPackMaxLaunches = 8;
constexpr int kPhaseClockCopies = 512;  // TileArgs::dbg_phase: [copies][16] sums of the per-phase clocks (profiling)

struct alignas(16) PackTileInfo {
  uint64_t rec_first;      // absolute index of the tile's first record
  uint64_t bit_pos;        // bit position of the tile in the blob (section-relative until finalised)
  uint64_t sec_start_bit;  // bit position of the tile's section in the blob (section index until finalised)
  uint32_t n_last;         // records in the tile | last tile of its section << 31
  uint32_t before;         // records of the section in front of the tile
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

// The value of x from here on is "new" to the compiler (no instruction): nothing computed from the old value can
// be reused, so what was derived from it does not have to stay in registers.  The CPU execution model defines
// this as nothing.
#ifndef JXLT_LAUNDER_VGPR
#define JXLT_LAUNDER_VGPR(x) asm volatile("" : "+v"(x))
#endif

// x is "written" here as far as the compiler can tell (no instruction: it holds whatever the register held).  For
// values that are computed under a condition and only ever used under the same condition: without a definition on
// the other path the compiler zero-initialises them in front of the branch -- 48 v_mov per thread for the coefficient
// registers of tile_kernel's transform phase.  The CPU execution model defines this as x = 0.
#ifndef JXLT_DEFINE_VGPR
#define JXLT_DEFINE_VGPR(x) asm volatile("" : "=v"(x))
#endif

// The root table's byte offset as the device uses it.  (The CPU execution model -- where an offset beyond the table
// would leave the process's memory instead of reading a harmless word -- wraps it into the table.)
#ifndef JXLT_LUT_WRAP
#define JXLT_LUT_WRAP(off) (off)
#endif

// Pointers to read-only bytes / floats in global memory, for the places that make a pointer opaque
// (JXLT_LAUNDER_SGPR) and have to say what it points into afterwards.  (The CPU execution model: plain pointers.)
#ifndef JXLT_GLOBAL_POINTER_TYPES
#define JXLT_GLOBAL_POINTER_TYPES
typedef __attribute__((address_space(1))) const char* JxltGlobalBytes;
typedef __attribute__((address_space(1))) const float* JxltGlobalFloats;
typedef __attribute__((address_space(1))) int16_t* JxltGlobalShorts;
typedef __attribute__((address_space(1))) const int16_t* JxltGlobalConstShorts;
typedef __attribute__((address_space(1))) const uint32_t* JxltGlobalConstWords;
#endif

// p[i] = v for a wave-uniform 64-bit value and a wave-uniform, 8-byte aligned address in global memory: ONE scalar
// store (s_store_dwordx2; i: a compile-time index) instead of moves to vector registers and a vector store by one
// lane.  Scalar stores go through the scalar data cache: JXLT_SCALAR_STORES_DONE() (s_dcache_wb) behind the last of
// them writes it back -- the compiler does that at a kernel's end only for scalar stores of its own making.  Nothing
// in the kernels READS these locations through the scalar cache.  (The CPU execution model: plain stores.)
#ifndef JXLT_SCALAR_STORE64
#define JXLT_SCALAR_STORE64(p, i, v) \
  asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"((unsigned long long)(v)), "s"(p), "n"((i) * 8) : "memory")
// (Scalar memory operations complete out of order: the wait IN FRONT of s_dcache_wb is what makes it cover the stores
// above it.  Without it the last stores of a wave could reach the cache behind the write-back and stay there, dirty,
// until some later wave of those CUs wrote the cache back -- on a small frame nobody did before token_kernel read
// the masks: one wrong context byte in two of six runs of the HDR tests, round 5.)
#define JXLT_SCALAR_STORES_DONE() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// The same for a wave-uniform value in scalar registers (a pointer, an index).
#ifndef JXLT_LAUNDER_SGPR
#define JXLT_LAUNDER_SGPR(x) asm volatile("" : "+s"(x))
#endif

// Nothing is kept in registers across this point that the compiler could re-read from memory instead, and no
// load behind it is answered from a store in front of it (no instruction).  Used where values are parked in LDS
// to free their registers: without it the compiler forwards the parked values to the loads that fetch them back
// and keeps them in registers all the same.
#ifndef JXLT_COMPILER_FENCE
#define JXLT_COMPILER_FENCE() asm volatile("" ::: "memory")
#endif

// A 16-byte LDS load that is ISSUED WHERE IT STANDS, and the wait that goes with it.  Software pipelining of a loop
// ("request the next round's operands, work on this round's") does not survive the compiler: loads whose values are
// first used in another basic block are sunk to that use, and every round waits for its own operands (round 6: the
// chroma-from-luma chains ran that way; `volatile` is no way out -- on this target it turns the load into a flat,
// system-coherent one with a full wait behind it).  So the load is an assembly statement, and because the compiler
// does not count what assembly has in flight, so is its wait: JXLT_LDS_WAIT4(n, a, b, c, d) = "all but my n newest
// LDS operations have returned" (they return in order) and ties the four registers to that point -- as inputs, so
// that they stay allocated until the data is there, and as outputs, so that no use is moved in front of it.
// `p`: a pointer into LDS, `byte_off`: a compile-time constant.  The CPU execution model loads plainly.
#ifndef JXLT_LDS_LOAD4_NOW
typedef float JxltFloat4 __attribute__((ext_vector_type(4)));
#define JXLT_LDS_LOAD4_NOW(dst, p, byte_off)                                                                   \
  asm volatile("ds_read_b128 %0, %1 offset:%2"                                                                 \
               : "=v"(dst)                                                                                     \
               : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)(p)), "n"(byte_off))
// (the 32-bit LDS address of a pointer into LDS, for address arithmetic in integers)
#define JXLT_LDS_ADDRESS(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)(p))
#define JXLT_LDS_LOAD4_NOW_AT(dst, addr, byte_off) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(byte_off))
#define JXLT_LDS_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
// (for registers whose contents are NOT used afterwards: they only have to stay allocated until the data is there)
#define JXLT_LDS_DRAIN4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(a), "v"(b), "v"(c), "v"(d))
#endif

// The wave's global stores so far are written (acknowledged by L2) -- for bytes that another wave of the workgroup
// overwrites behind a barrier: __syncthreads() alone orders the workgroup's LDS traffic, it does not wait for
// outstanding vector stores.
#ifndef JXLT_STORES_WRITTEN
#define JXLT_STORES_WRITTEN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

JXLT_DI int imin(int a, int b) { return a < b ? a : b; }
JXLT_DI int imax(int a, int b) { return a > b ? a : b; }
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
// Four DPP-fused adds (16 issue cycles).  Left to the compiler the first step is a zero-initialising move, two
// bank-masked DPP moves and an add, and the last one a DPP move and an add: 22 cycles -- it fuses only the middle
// step (GCNDPPCombine does not merge a DPP move whose bank mask is partial).  Floating-point addition is
// commutative, so "partner + own" equals the reference's "own + partner" bit for bit.
// (s_nop 1: a VALU write of a register is followed by two wait states before a DPP instruction may read it; the
// compiler's hazard recogniser does not look into inline assembly.  The CPU execution model of the tests defines
// JXLT_OCTET_SUM_PORTABLE and runs the three exchange steps below.)
JXLT_DI float octet_sum(float v) {
#ifdef JXLT_OCTET_SUM_PORTABLE
  v = v + octet_xor<4>(v);
  v = v + octet_xor<2>(v);
  v = v + octet_xor<1>(v);
  return v;
#else
  float r;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(r)
      : "v"(v));
  return r;
#endif
}
// Two octet sums for the price of one: lanes 0-3 of the octet receive SumOfLanes(a), lanes 4-7 SumOfLanes(b) -- the
// first exchange step adds the partner's `a` in the lower half and the partner's `b` in the upper half (bank-masked
// DPP adds), the other two steps stay inside the halves.  Same additions in the same order as octet_sum for either.
JXLT_DI float octet_sum_pair(float a, float b, int l) {
#ifdef JXLT_OCTET_SUM_PORTABLE
  const float sa = octet_sum(a), sb = octet_sum(b);
  return l < 4 ? sa : sb;
#else
  (void)l;
  float r;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(r)
      : "v"(a), "v"(b));
  return r;
#endif
}
// The value lanes 4-7 of the octet hold, in lanes 0-3 (lane l reads lane l + 4; lanes 4-7 keep their own).
JXLT_DI float octet_upper_to_lower(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), kDppRowShl4, 0xF, 0x5, false));
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
  const uint32_t hi = (__float_as_uint((float)value) >> 21) - (127u << 2);
  return value < 16 ? value : hi;
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

// Correctly rounded sqrtf for x in [2^-27, 2^63] from the reciprocal square root: v_rsq_f32, two multiplications and five
// fused multiply-adds (the rsq form of the compiler's own IEEE expansion, without its range scaling) -- no compare, no
// select, one instruction fewer than sqrt_exact_midrange and none of the 4-cycle kind.  tools/sqrt_rsq_probe.hip
// compares it with IEEE sqrtf on all 90 x 2^23 floats of that range on the GPU (tests/test_gpu_parity.py runs it);
// the CPU execution model, whose "rsq" is not this GPU's, takes sqrtf.
JXLT_DI float sqrt_exact_by_rsq(float x) {
#ifdef JXLT_SQRT_PORTABLE
  return sqrtf(x);
#else
  const float r = __builtin_amdgcn_rsqf(x);
  float g = x * r;
  float h = 0.5f * r;
  const float e = nfma32(h, g, 0.5f);
  g = fma32(g, e, g);
  h = fma32(h, e, h);
  const float d = nfma32(g, g, x);
  return fma32(d, h, g);
#endif
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
  // Round 5: ONE v_max instead of a compare and a select (all three of the 4-cycle kind).  An input at or below
  // kFloor = 2^-90 is replaced by kFloor, whose result is `add` exactly -- as the reference's for an input clamped to
  // zero: the cube root of 2^-90 is 2^-30, far below half an ulp of add (2^-27 for add = -0.1559...), and r stays
  // finite on the way (r ~ x^(-1/3) = 2^30, r^4 = 2^120).  Inputs in (0, 2^-90] -- biased mixes that cancel to
  // within 1e-27: where the reference's own iteration overflows (r^4 > 2^128 below 2^-96) -- are outside the domain
  // of the guarantee like every other route to a NaN.
  const float k1_3 = 1.0f / 3, k4_3 = 4.0f / 3;
  const float kFloor = 8.0779356694631609e-28f;  // 2^-90
  const float x = fmaxf(mixed, kFloor);
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
  return fma32(r2, x, add);
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

// kFenced: scheduling fences between the stages, so that no more than 24 values are live at a time (the 12-wave
// kernel, which has 80 registers; left to itself the scheduler keeps the inputs, both halves and the first
// half's transform in flight together).
template <bool kFenced = false>
JXLT_DI void dct16(float* m) {
  const float kW[8] = {(float)0.5024192861881557, (float)0.5224986149396889,
                       (float)0.5669440348163577, (float)0.6468217833599901,
                       (float)0.7881546234512502, (float)1.060677685990347,
                       (float)1.7224470982383342, (float)5.101148618689155};
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = m[i] + m[15 - i];
  if (!kFenced) dct8(a);
#pragma unroll
  for (int i = 0; i < 8; i++) b[i] = (m[i] - m[15 - i]) * kW[i];
  if (kFenced) {
    JXLT_SCHED_FENCE();
    dct8(a);
    JXLT_SCHED_FENCE();
  }
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

// No instruction: the wave's LDS operations execute in order (tools/lds_order_probe.hip checks exactly this on the
// GPU).  What has to be stopped is the compiler -- the stores and the loads of a transpose go through different
// types (float / float4), which type-based alias analysis treats as independent -- hence the memory clobber.
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

// 8x8 transpose across the 8 lanes of an octet: lane l holds v[j] = M[j][l] and ends with v[j] = M[l][j].
// Through a private LDS area of the octet: eight dword stores (row j of the matrix, a column per lane), two 16-byte
// loads of the lane's row.  (In registers -- three butterfly stages of DPP moves and selects, ~175 issue cycles -- the
// kernel was slower in round 1; that variant went in round 5.)
//
// LAYOUT (round 5).  What a transpose costs is LDS time, not issue slots -- the transform phase of tile_kernel keeps
// the LDS busier than the vector units -- and the LDS serves an instruction in fixed lane groups
// (MI355X_MICROARCH.md, "LDS"): a dword store in two groups of 32 lanes over 32 banks, a 16-byte load in four groups
// of 16 lanes -- lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- over 64 banks.  So the rows 0-3
// of octets 0 and 3 meet the rows 4-7 of octets 1 and 2 in one load group.  The area is 64 floats = 16 chunks of 16
// bytes, 256-byte aligned, no padding; the chunk of (row r, columns 4h .. 4h + 3) of the wave's octet o is
//     ((r & 3) ^ (o & 3))  |  (h ^ (o >> 1 & 1)) << 2  |  (r >> 2) << 3
// -- per load group the sixteen chunks fall on sixteen different bank quadruples, per store group the eight chunk
// starts on eight (tools/lds_layout_search.py evaluates the guide's rules; rounds 1-4 used a padded layout of 72
// floats made for 32 banks and plain lane halves: its loads met 3 to a bank, 16 extra LDS cycles per transpose, 1150
// per tile -- SQ_LDS_BANK_CONFLICT of the transform phase, profiles/pmc/r05_tile8192_phases.txt).
constexpr int kTransposePitch = 64;
// sc: the octet's area; o: the octet's number within its wave (0..7); l: the lane within the octet.
JXLT_DI void octet_transpose(float* v, float* sc, int o, int l) {
  const int sw = o & 3, hs = (o >> 1) & 1;
  // stores: row j, the lane's column l -> chunk ((j & 3) ^ sw) | ((l >> 2) ^ hs) << 2 | (j >> 2) << 3, float l & 3
  float* const w0 = sc + ((((l >> 2) ^ hs) << 2) << 2) + (l & 3);
#pragma unroll
  for (int j = 0; j < 8; j++) w0[((((j & 3) ^ sw)) << 2) + ((j >> 2) << 5)] = v[j];
  JXLT_OCTET_SYNC();
  // loads: the lane's row l, columns 0-3 and 4-7
  const float* const r0 = sc + (((((l & 3) ^ sw)) | (hs << 2) | ((l >> 2) << 3)) << 2);
  const float4 a = *reinterpret_cast<const float4*>(r0);
  const float4 b = *reinterpret_cast<const float4*>(sc + ((((((l & 3) ^ sw)) | ((hs ^ 1) << 2) | ((l >> 2) << 3))) << 2));
  JXLT_OCTET_SYNC();  // (the next transpose overwrites the area)
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Lane i of the wave stores `val` to the dword i of the LDS row at `row_base` + byte offset `off` (a compile-time
// constant): ds_write_addtid_b32 -- no address register, and half the LDS store path's time of a ds_write_b32
// (2 cycles per wave instruction instead of 4, MI355X_MICROARCH.md).  `row_base` is wave-uniform (it goes to M0; one
// wait state between the scalar move and the store; M0 is a reserved register that the compiler does not hand to
// anything else in these kernels -- tools/asm_budget.py's assembly shows no other use -- and that a clobber list may
// not name).  The CPU execution model stores through the pointer.
#ifndef JXLT_LDS_STORE_ROW
#define JXLT_LDS_STORE_ROW(row_base, off, val)                                                                  \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%2"                              \
               :                                                                                                \
               : "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)(row_base)), "v"(val), "n"(off) \
               : "memory")
#endif

// The same transpose through HALF an area per octet (round 5: the pair octets of the 12-wave kernel, for which there
// is no LDS for 32 more areas; until round 4 they transposed in registers, the longest path through the transform
// phase).  A wave's eight octets share four ROWS of 64 floats at a pitch of 80: rows 0-3 of the matrices go there --
// every lane stores its column's element of row j to dword `lane` of row j: one ds_write_addtid_b32 --, lanes 0-3 of
// every octet read their rows back, then rows 4-7 and lanes 4-7.  The pitch of 80 floats (16 more than a row) turns
// the bank quadruples by four from row to row, which is what keeps the sixteen lanes of a load group apart (the
// lanes of a group read four different rows of two octets).  An octet reads only what its own eight lanes wrote
// (octets of a wave may have diverged: edge tiles).
constexpr int kHalfTransposeRowPitch = 80;
constexpr int kHalfTransposeWaveFloats = 4 * kHalfTransposeRowPitch;
JXLT_DI void octet_transpose_half(float* v, float* wave_area, int o, int l) {
  const float* const rd = wave_area + (l & 3) * kHalfTransposeRowPitch + o * 8;  // the lane's row (of the half)
  float4 a, b;
  JXLT_LDS_STORE_ROW(wave_area, 0 * kHalfTransposeRowPitch * 4, v[0]);
  JXLT_LDS_STORE_ROW(wave_area, 1 * kHalfTransposeRowPitch * 4, v[1]);
  JXLT_LDS_STORE_ROW(wave_area, 2 * kHalfTransposeRowPitch * 4, v[2]);
  JXLT_LDS_STORE_ROW(wave_area, 3 * kHalfTransposeRowPitch * 4, v[3]);
  JXLT_OCTET_SYNC();
  if (l < 4) {
    a = *reinterpret_cast<const float4*>(rd);
    b = *reinterpret_cast<const float4*>(rd + 4);
  }
  JXLT_OCTET_SYNC();
  JXLT_LDS_STORE_ROW(wave_area, 0 * kHalfTransposeRowPitch * 4, v[4]);
  JXLT_LDS_STORE_ROW(wave_area, 1 * kHalfTransposeRowPitch * 4, v[5]);
  JXLT_LDS_STORE_ROW(wave_area, 2 * kHalfTransposeRowPitch * 4, v[6]);
  JXLT_LDS_STORE_ROW(wave_area, 3 * kHalfTransposeRowPitch * 4, v[7]);
  JXLT_OCTET_SYNC();
  if (l >= 4) {
    a = *reinterpret_cast<const float4*>(rd);
    b = *reinterpret_cast<const float4*>(rd + 4);
  }
  JXLT_OCTET_SYNC();  // (the next transpose overwrites the area)
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Block transforms.  `px` points at the block's top-left sample in an LDS plane
// of row pitch `pitch`; l = lane within the octet.  Results are the lane's
// "rows of 8": coefficient index i = r*8 + l (the reference's SIMD layout).

// The reference scales by 1/N after each 1-D pass (StoreToBlockAndScale, :387-390).  Those
// factors are powers of two, and scaling by a power of two commutes exactly with every
// rounded add/mul/fma (no over/underflow at these magnitudes: pixel differences are 0 or
// >= 1 ulp of O(0.1) values), so they are not applied here at all: the results are kDct8Norm /
// kDct16Norm times the reference's coefficients and the tables they are multiplied with carry
// the inverse (DeviceTables::inv_weights, scan_consts).

// ComputeScaledDCT<8,8> (enc_transforms-inl.h:527-546): i = h*8 + v
// kHalf: transpose through the wave's half-size rows `sc` (octet_transpose_half) instead of the octet's own area
// `sc`; `o` = the octet's number within its wave.
template <bool kHalf = false>
JXLT_DI void block_dct8x8(const float* px, int pitch, int l, float* sc, float* c, int o) {
#pragma unroll
  for (int y = 0; y < 8; y++) c[y] = px[y * pitch + l];
  dct8(c);
  // lane v then holds 8*A[v][x], x = 0..7
  if (kHalf) octet_transpose_half(c, sc, o, l);
  else octet_transpose(c, sc, o, l);
  dct8(c);  // c[h] = kDct8Norm * C[h][v=l]
}

// ComputeScaledDCT<16,8>: 16 rows x 8 cols, i = h*16 + v; r = 2h + (v>=8), lane = v&7
template <bool kFenced = false>
JXLT_DI void block_dct16x8(const float* px, int pitch, int l, float* sc, float* c, int o) {
  float col[16];
#pragma unroll
  for (int y = 0; y < 16; y++) col[y] = px[y * pitch + l];
  dct16<kFenced>(col);
  float lo[8], hi[8];
#pragma unroll
  for (int v = 0; v < 8; v++) {
    lo[v] = col[v];
    hi[v] = col[v + 8];
  }
  if (kFenced) JXLT_SCHED_FENCE();
  octet_transpose(lo, sc, o, l);  // lane t: A[t][x]
  octet_transpose(hi, sc, o, l);  // lane t: A[t+8][x]
  if (kFenced) JXLT_SCHED_FENCE();
  dct8(lo);
  if (kFenced) JXLT_SCHED_FENCE();
  dct8(hi);
#pragma unroll
  for (int h = 0; h < 8; h++) {  // (kDct16Norm times the reference's)
    c[2 * h] = lo[h];
    c[2 * h + 1] = hi[h];
  }
}

// ComputeScaledDCT<8,16>: 8 rows x 16 cols, i = v*16 + h; r = 2v + (h>=8), lane = h&7
template <bool kFenced = false>
JXLT_DI void block_dct8x16(const float* px, int pitch, int l, float* sc, float* c, int o) {
  float lo[8], hi[8];
#pragma unroll
  for (int y = 0; y < 8; y++) {
    lo[y] = px[y * pitch + l];
    hi[y] = px[y * pitch + l + 8];
  }
  dct8(lo);
  if (kFenced) JXLT_SCHED_FENCE();
  dct8(hi);
  if (kFenced) JXLT_SCHED_FENCE();
  octet_transpose(lo, sc, o, l);  // lane v: A[v][x], x < 8
  octet_transpose(hi, sc, o, l);  // lane v: A[v][x], x >= 8
  float row[16];
#pragma unroll
  for (int x = 0; x < 8; x++) {
    row[x] = lo[x];
    row[x + 8] = hi[x];
  }
  if (kFenced) JXLT_SCHED_FENCE();
  dct16<kFenced>(row);
#pragma unroll
  for (int h = 0; h < 8; h++) {  // (kDct16Norm times the reference's)
    lo[h] = row[h];
    hi[h] = row[h + 8];
  }
  octet_transpose(lo, sc, o, l);  // lane t: C[v][h=t], v = 0..7
  octet_transpose(hi, sc, o, l);  // lane t: C[v][h=t+8]
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
  return 0.25f * sqrt_exact_by_rsq(fma32(v, sqrt_mul, kLogOffset));  // argument >= kLogOffset = 26.48
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

}  // namespace jxlt_dev

#endif  // JXLT_DEVICE_COMMON_H_
