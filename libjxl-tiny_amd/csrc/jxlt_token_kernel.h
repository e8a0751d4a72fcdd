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
// they belong to): a lane finds its block through the per-window index of the block that holds the window's
// first token plus the block starts inside the window, and needs nothing from its neighbours -- "nonzeros still
// to come" and "previous coefficient nonzero" come from the nonzero masks tile_kernel leaves per entry.  (A
// 64-lane pass per entry, the structure until the end of round 2, filled 18 % of its lane slots on ordinary
// content -- 39 tokens per Y entry, 0.6 per chroma entry -- and kept the CU's one scalar unit busy with
// per-entry bookkeeping.)  The nzeros tokens (one per entry) are written by a thread-per-block pass.
// kWide: the frame has more than kTokenNarrowBlocks blocks -- coefficient indices (192 per block) need 64 bits
// (+190 VALU per wave, 2.7 % of the kernel's time: frames below the bound run the 32-bit variant).
constexpr size_t kTokenNarrowBlocks = (size_t(1) << 32) / 192;  // 22.4 M blocks = 1.43 Gpixel
template <bool kWide>
JXLT_DI void token_kernel_body(const TokenArgs& A) {
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
  auto hist_slot = [](uint32_t cm, uint32_t sym) { return cm * 64u + ((sym + cm) & 63u); };
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
  // 32-bit block indices and per-block word indices (the C ABI limits a frame to 2^28 blocks -- block * 12 stays
  // below 2^32 --, a group's tokens to 196 608 records); only the coefficient index is formed in 64 bits
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
    uint32_t coef_at, mask_at;  // where its coefficient / its entry's nonzero masks are: element index (kWide: index
                                // of the 64-coefficient run, block * 3 + channel, that holds scan position k) / word index
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
    if constexpr (kWide) {
      t.coef_at = (t.k < 64 ? pos : pos1) * 3 + (uint32_t)c;
    } else {
      t.coef_at = t.k < 64 ? (pos * 3 + (uint32_t)c) * 64 + (uint32_t)t.k : (pos1 * 3 + (uint32_t)c) * 64 + (uint32_t)(t.k - 64);
    }
    t.mask_at = (pos * 3 + (uint32_t)c) * 4;
  };
  auto request = [&](Located& t) {
    if constexpr (kWide) {
      t.coef = (int)A.coef_scan[((size_t)t.coef_at << 6) + (size_t)(t.k & 63)];
    } else {
      t.coef = (int)A.coef_scan[t.coef_at];
    }
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

__global__ void __launch_bounds__(kTokenThreads) token_kernel(const TokenArgs A) { token_kernel_body<false>(A); }
__global__ void __launch_bounds__(kTokenThreads) token_kernel_wide(const TokenArgs A) { token_kernel_body<true>(A); }

}  // namespace jxlt_dev

#endif  // JXLT_TOKEN_KERNEL_H_
