// jxlt_token_kernel.h -- group_scan_kernel and token_kernel: the AC groups' token records in stream
// order, a lane per coefficient token (enc_group.cc:444-494).  Part of jxlt_device.h (include that one).
#ifndef JXLT_TOKEN_KERNEL_H_
#define JXLT_TOKEN_KERNEL_H_

#include "jxlt_device_common.h"

namespace jxlt_dev {

// Zeroes up to four small arrays of 32-bit words in one launch (the per-frame counters and histograms).
struct ClearArgs {
  uint32_t* p[4];
  uint32_t n[4];
};
__global__ void __launch_bounds__(256) clear_counters_kernel(const ClearArgs A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (i < A.n[k]) A.p[k][i] = 0u;
}


// ---------------------------------------------------------------------------
// Token kernel: one workgroup per 256x256 group (enc_group.cc:444-494)
// ---------------------------------------------------------------------------
constexpr int kTokenThreads = 512;

// A lane per COEFFICIENT TOKEN of the group's stream (window w = tokens 64 w .. 64 w + 63, whatever entries
// they belong to).  A lane needs nothing from its neighbours: "nonzeros still to come" and "previous coefficient
// nonzero" come from the nonzero masks tile_kernel leaves per entry.  (A 64-lane pass per entry, the structure until
// the end of round 2, filled 18 % of its lane slots on ordinary content -- 39 tokens per Y entry, 0.6 per chroma
// entry -- and kept the CU's one scalar unit busy with per-entry bookkeeping.)  The nzeros tokens (one per entry) are
// written by a thread-per-block pass.
//
// Round 5: what a lane needs to know about its block is worked out ONCE per block, by the thread-per-block pass, and
// left in LDS as a compact list of the blocks that have coefficient tokens (12 bytes each); a window's lanes find
// their block as "the window's first block + the number of listed blocks that start at or below my position" -- one
// ballot and v_mbcnt --, and the entry, scan position, coefficient and mask addresses follow from the twelve bytes
// with some thirty instructions.  Until then every lane redid the per-block arithmetic (a division by the group's
// width, the entries' token counts, the block's position in the frame) and searched the blocks' starts with a
// count-leading-zeros ladder: 160 vector instructions per window, 68 % of them in the 4-cycle class, where this
// kernel is bound by vector issue (7.4 k per wave, eight waves per SIMD).
//
// kWide: frames wider than kTokenNarrowWidth blocks -- the lane's coefficient address is formed in 64 bits there (the
// narrow variant multiplies with v_mad_u32_u24 and addresses relative to its group: 32-bit offsets on scalar bases).
constexpr uint32_t kTokenNarrowWidth = ((1u << 24) - 1u) / 192u;  // 87 381 blocks = 699 048 pixels
template <bool kWide>
JXLT_DI void token_kernel_body(const TokenArgs& A) {
  // The compact list, entry j = the j-th block (in stream order) that has coefficient tokens:
  //   cstart[j] = coefficient tokens in front of it << 12 | row in the group << 7 | column << 2 | second block below
  //               (not beside) << 1 | two-block transform;  0xFFFFFFFF behind the list's end (64 entries)
  //   cdesc[j].x = (tokens in front of the block + 1) | coefficient tokens of its y entry << 18 | of its x entry << 25
  //   cdesc[j].y = per entry (y, x, b: a byte each) its nzeros + the bit tile_kernel leaves at scan position
  //                covered - 1 of its mask (see emit)
  // While the scan runs, the same memory holds its 64-bit words.
  constexpr int kListPad = 68;
  __shared__ alignas(16) uint32_t list_mem[1024 + kListPad + 2 * 1024];
  uint32_t* const cstart = list_mem;
  uint2* const cdesc = reinterpret_cast<uint2*>(list_mem + 1024 + kListPad);
  // per block, while the scan runs: coefficient tokens (bits 0-19) | first blocks (20-30) | blocks with coefficient
  // tokens (32-42) -- afterwards the sums over the blocks in front of it
  uint64_t* const scan = reinterpret_cast<uint64_t*>(list_mem);
  static_assert(sizeof(list_mem) >= 1025 * sizeof(uint64_t), "the scan's words fit");
  __shared__ uint16_t first_c[3072 + 8];  // per window: the list entry that holds its first coefficient token
  __shared__ uint64_t wsum[kTokenThreads / 64];
  __shared__ uint32_t hist[64 * 64];
  // The two context tables as bytes indexed by what a lane has at hand -- "nonzeros still to come" and the scan
  // position -- for one-block transforms ([0, 64)) and two-block transforms ([64, 192): the index halved, rounded
  // up for the nonzeros, enc_group.cc:468-475): a lane adds 0 or 64 to its index instead of shifting it.
  __shared__ uint8_t s_nnz_tab[192], s_freq_tab[192];
  __shared__ alignas(4) uint8_t s_ctx_map[1980];
  // nzeros grid of the group (PredictFromTopAndLeft input of the nzeros tokens); once those are written its first
  // 512 bytes are `flags`: per wave, "a listed block starts at this position of the window"
  __shared__ alignas(4) uint8_t s_nzg[3 * 1024];
  __shared__ uint64_t s_group_base;
  __shared__ uint64_t gsum[kTokenThreads / 64];
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DeviceTables* T = A.tab;
  const int group = A.group_first + (int)blockIdx.x;
  const bool do_hist = A.histogram != nullptr;
  // LDS histogram slot of (pre-clustered context, symbol): the symbol is rotated by the context, so that the
  // small symbols nearly all tokens have do not land in the same few banks for every context
  auto hist_slot = [](uint32_t cm, uint32_t sym) { return cm * 64u + ((sym + cm) & 63u); };
  if (do_hist)
    for (int i = tid; i < 64 * 64; i += kTokenThreads) hist[i] = 0;
  static_assert(1980 % 4 == 0 && 1980 / 4 <= kTokenThreads, "one word per thread");
  if (tid < 1980 / 4)
    reinterpret_cast<uint32_t*>(s_ctx_map)[tid] = reinterpret_cast<const uint32_t*>(T->ac_context_map)[tid];
  if (tid < 192) {  // (entry 0 of the reference's tables is a marker no token reaches)
    s_nnz_tab[tid] = (uint8_t)T->nnz_context[tid < 64 ? tid : (tid - 64 + 1) >> 1];
    s_freq_tab[tid] = (uint8_t)T->freq_context[tid < 64 ? tid : (tid - 64) >> 1];
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
  // 32-bit block indices (the C ABI limits a frame to 2^28 blocks, a group's tokens to 196 608 records)
  const uint32_t bstride = (uint32_t)A.g.xsize_blocks;
  const uint32_t nbx_magic = 65536u / (uint32_t)nbx + 1u;  // b / nbx == (b * magic) >> 16 for b < 1024, nbx <= 32
  const uint32_t origin = (uint32_t)by0 * bstride + (uint32_t)bx0;  // the group's first block

  // metadata per entry, coefficient-token and first-block counts per block, predicted-nzeros grid -> LDS
  const int nblk = nbx * nby;
  // (a group has at most 1024 blocks, two per thread: all ten bytes of both are requested before the first is
  // used, whether the block turns out to be a first block or not -- one round trip instead of four; the thread keeps
  // them in registers for the pass behind the scan)
  static_assert(2 * kTokenThreads >= 1024, "two blocks per thread");
  uint32_t blk_a[2], blk_nscan[2][3], blk_nzs[2][3];  // strategy byte; per entry in stream order y, x, b
  {
    uint32_t ld_a[2], ld_nscan[2][3], ld_nz[2][3], ld_grid[2][3];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int b = imin(tid + r * kTokenThreads, nblk - 1);
      const int by = (int)(((uint32_t)b * nbx_magic) >> 16), bx = b - by * nbx;
      const uint32_t pos = origin + (uint32_t)by * bstride + (uint32_t)bx;
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
      blk_a[r] = 0;
#pragma unroll
      for (int ci = 0; ci < 3; ci++) blk_nscan[r][ci] = blk_nzs[r][ci] = 0;
      if (b >= nblk) continue;
      const uint32_t a = ld_a[r];
      blk_a[r] = a;
      const int covered = (a >> 1) == 0 ? 1 : 2;
      uint32_t ncoef = 0;
#pragma unroll
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
        if (a & 1) {
          blk_nscan[r][ci] = ld_nscan[r][c] > (uint32_t)covered ? ld_nscan[r][c] - covered : 0;  // coefficient tokens
          blk_nzs[r][ci] = ld_nz[r][c];
          ncoef += blk_nscan[r][ci];
        }
        s_nzg[c * 1024 + b] = (uint8_t)ld_grid[r][c];
      }
      scan[b + 1] = (uint64_t)(ncoef | ((a & 1) << 20)) | ((uint64_t)(ncoef != 0 ? 1u : 0u) << 32);
    }
  }
  if (tid == 0) scan[0] = 0;
  __syncthreads();
  if (tid == 0) {
    uint64_t base = 0;
    for (int w = 0; w < kTokenThreads / 64; w++) base += gsum[w];
    s_group_base = base;
    A.group_tok_offset[group] = base;
    const int ngroups = A.g.xsize_groups * A.g.ysize_groups;
    if (group + 1 == ngroups) A.group_tok_offset[ngroups] = base + A.group_ntok[group];
  }
  // inclusive scan over scan[1..nblk] (blocked: each thread owns a contiguous run; the three fields at once: the
  // sums stay inside their bit ranges, <= 387 072 coefficient tokens, <= 1024 first blocks, <= 1024 listed blocks)
  {
    const int per = (nblk + kTokenThreads - 1) / kTokenThreads;
    const int beg = 1 + tid * per, end = imin(1 + nblk, beg + per);
    uint64_t sum = 0;
    for (int i = beg; i < end; i++) sum += scan[i];
    uint64_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t wbase = 0;
    for (int w = 0; w < wave; w++) wbase += wsum[w];
    uint64_t run = wbase + incl - sum;
    for (int i = beg; i < end; i++) {
      run += scan[i];
      scan[i] = run;
    }
  }
  __syncthreads();
  // scan[b] is now what lies in front of block b (scan[nblk]: the totals)
  const uint32_t kLow = (1u << 20) - 1u;
  const uint64_t totals = scan[nblk];
  const uint32_t ncoef_total = (uint32_t)totals & kLow;
  const int nlisted = (int)(totals >> 32);
  const int nwin = (int)((ncoef_total + 63u) >> 6);
  uint64_t in_front[2];
#pragma unroll
  for (int r = 0; r < 2; r++) in_front[r] = scan[imin(tid + r * kTokenThreads, nblk)];
  __syncthreads();  // (the list is written over the scan's words)
  uint8_t* out = A.tokens + 3 * s_group_base;  // (written before the barriers of the scan above)
  if (tid < kListPad) cstart[nlisted + tid] = 0xFFFFFFFFu;
  // per block: its list entry, its windows' index entries and the nzeros tokens of its three entries
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int b = tid + r * kTokenThreads;
    if (b >= nblk) continue;
    const uint32_t a = blk_a[r];
    if (!(a & 1)) continue;  // not the first block of a transform: no entries
    const uint32_t s0 = (uint32_t)in_front[r] & kLow, nfirst = ((uint32_t)in_front[r] >> 20) & 0x7FFu;
    const uint32_t j = (uint32_t)(in_front[r] >> 32);
    const int st = (int)((a >> 1) & 0x7F);
    const int covered = st == 0 ? 1 : 2;
    const int cby = (int)(((uint32_t)b * nbx_magic) >> 16), cbx = b - cby * nbx;
    const uint32_t tl0 = s0 + 3u * nfirst;  // tokens in front of the block: coefficient tokens + three per first block
    const uint32_t ncoef = blk_nscan[r][0] + blk_nscan[r][1] + blk_nscan[r][2];
    if (ncoef != 0) {
      for (uint32_t q = (s0 + 63u) >> 6; (q << 6) < s0 + ncoef; q++) first_c[q] = (uint16_t)j;  // (windows that start in b)
      cstart[j] = (s0 << 12) | ((uint32_t)cby << 7) | ((uint32_t)cbx << 2) | (st == 1 ? 2u : 0u) | (uint32_t)(covered - 1);
      uint32_t adj = 0;
#pragma unroll
      for (int ci = 0; ci < 3; ci++)  // (nzeros + the bit at scan position covered - 1: see emit)
        adj |= (blk_nzs[r][ci] + (blk_nzs[r][ci] <= 4u * (uint32_t)covered ? 1u : 0u)) << (8 * ci);
      uint2 dd;
      dd.x = (tl0 + 1u) | (blk_nscan[r][0] << 18) | (blk_nscan[r][1] << 25);
      dd.y = adj;
      cdesc[j] = dd;
    }
    const int bctx_y = st == 0 ? 0 : 1, bctx_c = 2 + bctx_y;  // (ac_context.h:64-114, see below)
    uint32_t tl = tl0;
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
      const int nzl = (int)blk_nzs[r][ci];
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
      // (Every lane's record lies in a cache line of its own here -- a store instruction is 64 requests --, so ONE
      // unaligned 32-bit store where the byte behind the record is written later anyway: by this thread (the block's
      // next nzeros token) or by the coefficient-token loop behind the barrier (the entry's first coefficient token).
      // Only a block's LAST token is followed by another thread's record: the three bytes alone there.  As a byte and
      // a 16-bit store everywhere this pass cost 0.085 of the kernel's 0.83 Mcycles at 16384^2.)
      if (ci < 2 || blk_nscan[r][2] != 0) {
        typedef uint32_t __attribute__((aligned(1))) UnalignedWord;
        *reinterpret_cast<UnalignedWord*>(o) = (uint32_t)cm | ((uint32_t)nzl << 8);
      } else {
        o[0] = cm;
        o[1] = (uint8_t)(nzl & 0xFF);
        o[2] = (uint8_t)(nzl >> 8);
      }
      if (do_hist) {
        atomicAdd(&hist[hist_slot(cm, hybrid_uint_symbol((uint32_t)nzl))], 1u);
      }
      tl += 1 + blk_nscan[r][ci];
    }
  }
  JXLT_STORES_WRITTEN();  // (the byte behind an nzeros record, see above: written before another wave overwrites it)
  __syncthreads();  // the list and the window index are complete; the nzeros grid is read

  // ---- the coefficient tokens: wave w takes windows w, w + 8, ... ---------------------------------------------
  // Three stages per window: "locate" finds the lane's block, entry and scan position (LDS only), "request" asks
  // for its coefficient and its entry's nonzero masks, "emit" (when the values have arrived) derives the context
  // and stores the record.  The lanes behind the stream's end (last window only) repeat
  // the last token, so that every emit issues the same stores -- the wait for the next window's loads
  // can then be a count (loads and stores share one counter) instead of "everything".
  constexpr int kWaves = kTokenThreads / 64;
  uint8_t* const flags = &s_nzg[wave * 64];
  flags[lane] = 0;
  JXLT_WAVE_SYNC();
  // the group's coefficients and masks (a scalar base each; the lanes' offsets are relative to the group's first block)
  JxltGlobalBytes coef_base = (JxltGlobalBytes)((const char*)A.coef_scan + (size_t)origin * 384u);
  JxltGlobalBytes mask_base = (JxltGlobalBytes)((const char*)A.blk_nzmask + (size_t)origin * 48u);
  JXLT_LAUNDER_SGPR(coef_base);
  JXLT_LAUNDER_SGPR(mask_base);
  const uint32_t row192 = bstride * 192u;  // (narrow variant: below 2^24)
  struct Located {
    uint32_t out_index;  // the token's place in the group's stream
    uint32_t p;          // scan position - 1
    int nz_adj;          // nzeros of its entry + the bit at scan position covered - 1 of its mask + table_half
    int freq_at;         // scan position + table_half (table_half: 0 / 64 for one-block / two-block transforms)
    int ctx_half;        // half the first context of its block context (ac_context.h:64-114)
    bool real;           // counts (the lanes behind the stream's end repeat the last token)
    // where its coefficient / its entry's masks are -- narrow: byte offsets from coef_base / mask_base; wide: the
    // index (block - origin) * 3 + channel of the 64-coefficient run that holds scan position k / of its entry
    uint32_t coef_at, mask_at;
    int coef;            // requested
    uint32_t nz[4];      // requested: the entry's nonzero masks, scan positions covered - 1 .. 127
  };
  auto locate = [&](int q, Located& t) {
    const uint32_t w0 = (uint32_t)q << 6;  // the window's first token
    t.real = w0 + (uint32_t)lane < ncoef_total;
    const uint32_t i = (uint32_t)imin((int)(w0 + (uint32_t)lane), (int)(ncoef_total - 1u));
    const int j0 = __builtin_amdgcn_readfirstlane((int)first_c[q]);
    // the listed blocks that start inside the window (at most 63: each has a token), flagged at their first token's
    // position; my block = the window's first + the number of flags at or below my position.  (The lanes behind the
    // stream's end count every flag, as the last real lane does.)
    const uint32_t cand = cstart[j0 + 1 + lane];
    if (cand < ((w0 + 64u) << 12)) flags[(cand >> 12) - w0] = 1;
    JXLT_WAVE_SYNC();  // (the wave's LDS operations execute in order)
    const bool flagged = flags[lane] != 0;
    flags[lane] = 0;  // (for the next window: after the read above, before that window's flags)
    const unsigned long long fm = __ballot(flagged);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, flagged ? 1u : 0u));
    const uint32_t j = (uint32_t)j0 + rank;
    const uint32_t cw = cstart[j];
    const uint2 d = cdesc[j];
    const uint32_t in_block = i - (cw >> 12);  // coefficient token of the block
    const uint32_t log2c = cw & 1u;
    const uint32_t n_y = __builtin_amdgcn_ubfe(d.x, 18, 7), n_yx = n_y + (d.x >> 25);
    const bool ge1 = in_block >= n_y, ge2 = in_block >= n_yx;  // entry in stream order y, x, b: ge1 + ge2
    const uint32_t ci = (ge1 ? 1u : 0u) + (ge2 ? 1u : 0u);
    t.p = in_block - (ge2 ? n_yx : ge1 ? n_y : 0u) + log2c;
    const uint32_t table_half = log2c << 6;
    t.nz_adj = (int)(__builtin_amdgcn_ubfe(d.y, ci * 8u, 8) + table_half);
    t.freq_at = (int)(t.p + 1u + table_half);
    // block context: 0/1 for Y and 2/3 for X, B, the odd value for the two-block strategies
    t.ctx_half = (ge1 ? 2 * 37 + 458 : 2 * 37) + (log2c ? 229 : 0);
    // tokens in front of the block (coefficient tokens + three per first block), the nzeros tokens of this
    // and the earlier entries of the block, the coefficient tokens of the block in front of this one
    t.out_index = (d.x & 0x3FFFFu) + ci + in_block;
    const uint32_t c64 = ge2 ? 128u : ge1 ? 0u : 64u;  // channel (y, x, b -> 1, 0, 2) * 64
    const uint32_t cbx = __builtin_amdgcn_ubfe(cw, 2, 5), cby = __builtin_amdgcn_ubfe(cw, 7, 5);
    const bool second = t.p >= 63u;  // in the transform's second block
    if constexpr (kWide) {
      const uint32_t rel = cby * bstride + cbx;
      const uint32_t c = c64 >> 6;
      t.mask_at = rel * 3u + c;
      t.coef_at = (second ? rel + ((cw & 2u) ? bstride : 1u) : rel) * 3u + c;
    } else {
      const uint32_t e = __umul24(cby, row192) + __umul24(cbx, 192u) + c64;  // coefficient index of the entry, from the group's
      t.mask_at = e >> 2;                                                     // 16 bytes of masks per 64 coefficients
      t.coef_at = (e + ((t.p + 1u) & 63u) + (second ? ((cw & 2u) ? row192 : 192u) : 0u)) << 1;
    }
  };
  auto request = [&](Located& t) {
    if constexpr (kWide) {
      t.coef = (int)*(const int16_t*)((const char*)coef_base + ((((size_t)t.coef_at << 6) + (size_t)((t.p + 1u) & 63u)) << 1));
      const uint32_t* nzw = (const uint32_t*)((const char*)mask_base + ((size_t)t.mask_at << 4));
#pragma unroll
      for (int j = 0; j < 4; j++) t.nz[j] = nzw[j];
    } else {
      t.coef = (int)*(JxltGlobalConstShorts)(coef_base + t.coef_at);
      JxltGlobalConstWords nzw = (JxltGlobalConstWords)(mask_base + t.mask_at);
#pragma unroll
      for (int j = 0; j < 4; j++) t.nz[j] = nzw[j];
    }
  };
  auto emit = [&](const Located& t, bool counts) {
    // Nonzeros at the scan positions in front of k, and whether the previous one is nonzero -- for the entry's first
    // token (k = covered) "previous" is instead "nzeros <= size / 16" (enc_group.cc:476-480): tile_kernel leaves that
    // bit at scan position covered - 1 of the mask (the masks proper hold positions covered .. 127), and nz_adj counts
    // it, so that "still to come" = nz_adj - (set bits at positions < k) holds for every k without a special case.
    const uint32_t p = t.p;
    const bool upper = p >= 64u;
    const uint32_t x_lo = upper ? t.nz[2] : t.nz[0], x_hi = upper ? t.nz[3] : t.nz[1];
    const uint32_t whole = (uint32_t)__popc(t.nz[0]) + (uint32_t)__popc(t.nz[1]);
    const unsigned long long x = (unsigned long long)x_lo | ((unsigned long long)x_hi << 32);
    const unsigned long long above = ~1ull << (p & 63u);  // positions > p
    const int below = __popcll(x & ~above) + (int)(upper ? whole : 0u);
    const uint32_t prev = (uint32_t)(x >> (p & 63u)) & 1u;
    const int left_at = t.nz_adj - below;  // nzeros still to come at this position (+ table_half)
    const int ctx = ((t.ctx_half + (int)s_nnz_tab[left_at] + (int)s_freq_tab[t.freq_at]) << 1) | (int)prev;
    const uint8_t cm = s_ctx_map[ctx];
    const uint32_t val = pack_signed((int32_t)t.coef);
    uint8_t* o = out + 3u * t.out_index;
    o[0] = cm;
    o[1] = (uint8_t)(val & 0xFF);
    o[2] = (uint8_t)((val >> 8) & 0xFF);
    if (do_hist && counts && t.real) {
      const uint32_t slot = hist_slot(cm, hybrid_uint_symbol(val & 0xFFFFu));
      atomicAdd(&hist[slot], 1u);
    }
  };
  // Four windows in flight per wave: window n of the wave is emitted (the first use of its requested values), the
  // requests of window n + 1 are on their way, window n + 2 is requested and window n + 3 located -- a use comes two
  // "locate"s (LDS round trips only) behind its request, and the wait in front of it is a COUNT that leaves the next
  // window's two loads (and the two stores in between) outstanding: loads and stores share one counter, which
  // returns in order.  For the compiler to see that count, every path to an emit issues the same sequence of loads
  // and stores: requests and emits are unconditional -- behind the wave's last window they repeat a window that has
  // been emitted (same bytes to the same places; nothing counted) -- and the loop is entered behind the first emit.
  // (Round 4: three windows, the wait for "everything the wave has issued"; with the per-block work out of the loop
  // -- 160 -> 90 vector instructions per window -- one "locate" no longer covered the latency: the loop without its
  // loads took 0.58 instead of 0.83 Mcycles.)  Three sets of registers used in turn, no copies (a copy would be a use).
  const int cnt = (nwin - wave + kWaves - 1) / kWaves;  // this wave's windows: wave, wave + 8, ...
  if (cnt > 0) {
    Located s0, s1, s2;
    auto window = [&](int n) { return wave + imin(n, cnt - 1) * kWaves; };
    locate(window(0), s0);
    request(s0);
    locate(window(1), s1);
    request(s1);
    locate(window(2), s2);
    emit(s0, true);
    request(s2);
    if (3 < cnt) locate(window(3), s0);
    for (int n = 1; n < cnt; n += 3) {
      emit(s1, true);
      request(s0);
      if (n + 3 < cnt) locate(window(n + 3), s1);
      emit(s2, n + 1 < cnt);
      request(s1);
      if (n + 4 < cnt) locate(window(n + 4), s2);
      emit(s0, n + 2 < cnt);
      request(s2);
      if (n + 5 < cnt) locate(window(n + 5), s0);
    }
  }
  if (do_hist) {
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += kTokenThreads) {
      const uint32_t n = hist[hist_slot((uint32_t)i >> 6, (uint32_t)i & 63u)];
      if (n) atomicAdd(&A.histogram[i], n);
    }
  }
}

__global__ void __launch_bounds__(kTokenThreads) token_kernel(const TokenArgs A) { token_kernel_body<false>(A); }
__global__ void __launch_bounds__(kTokenThreads) token_kernel_wide(const TokenArgs A) { token_kernel_body<true>(A); }

}  // namespace jxlt_dev

#endif  // JXLT_TOKEN_KERNEL_H_
