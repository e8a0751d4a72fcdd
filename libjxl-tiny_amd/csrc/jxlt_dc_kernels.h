// jxlt_dc_kernels.h -- the DC groups' token records (enc_frame.cc:287-424, 536-570).
// Part of jxlt_device.h (include that one).
#ifndef JXLT_DC_KERNELS_H_
#define JXLT_DC_KERNELS_H_

#include "jxlt_device_common.h"

namespace jxlt_dev {

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
    // All of the thread's DC values are requested before the first record is stored (its column and the one to
    // the left, the part's rows and the one above, three channels: 54 independent loads).  As a loop of "load the
    // neighbours, store the record" every row waited for its own loads: the byte stores of the records may alias
    // anything, so the compiler keeps the order.
    constexpr int kRowsMax = 256 / kDcParts;  // a DC group is at most 256 block rows high
    const int rows_per = (d.nby + kDcParts - 1) / kDcParts;
    const int y0 = part * rows_per, y1 = imin(d.nby, y0 + rows_per);
    const int x = tid;
    if (x < d.nbx && y0 < y1) {
      int cur[3][kRowsMax + 1], lft[3][kRowsMax + 1];  // [channel in stream order][row - (y0 - 1)]
#pragma unroll
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;
        const int16_t* q = A.quant_dc[c] + (size_t)d.by0 * bstride + d.bx0 + x;
#pragma unroll
        for (int r = 0; r <= kRowsMax; r++) {
          const int y = y0 - 1 + r;
          const bool there = y >= 0 && y < y1;
          const int16_t* qr = q + (ptrdiff_t)(there ? y : y0) * (ptrdiff_t)bstride;
          cur[ci][r] = there ? (int)qr[0] : 0;
          lft[ci][r] = (there && x > 0) ? (int)qr[-1] : 0;
        }
      }
      // (likewise the context look-ups: all of them before the first store)
      uint8_t ctx_of[3][kRowsMax + 1];
      int res_of[3][kRowsMax + 1];
#pragma unroll
      for (int r = 1; r <= kRowsMax; r++) {
        const int y = y0 - 1 + r;
#pragma unroll
        for (int ci = 0; ci < 3; ci++) {
          const int left = x ? lft[ci][r] : y ? cur[ci][r - 1] : 0;
          const int top = y ? cur[ci][r - 1] : left;
          const int topleft = (x && y) ? lft[ci][r - 1] : left;
          const int guess = clamped_gradient(top, left, topleft);
          int gp = 512 + top + left - topleft;
          gp = gp < 0 ? 0 : gp > 1023 ? 1023 : gp;
          res_of[ci][r] = cur[ci][r] - guess;
          ctx_of[ci][r] = A.tab->gradient_lut[gp];
        }
      }
#pragma unroll
      for (int r = 1; r <= kRowsMax; r++) {
        const int y = y0 - 1 + r;
        if (y >= y1) break;
#pragma unroll
        for (int ci = 0; ci < 3; ci++)
          put_record(rec, d.pos_dc + (uint32_t)(ci * d.nb + y * d.nbx + x), ctx_of[ci][r], pack_signed(res_of[ci][r]),
                     hist);
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

// The two per-first-block token runs need, for every first block, its rank among the DC group's
// first blocks and the previous first block's (strategy code, quant field).  One workgroup per
// chunk of kDcChainBlocks blocks (64 chunks per full DC group, all of them in parallel):
// dc_chain_summary_kernel records each chunk's first-block count and its last first block's
// values; dc_chain_kernel derives a chunk's carry from the summaries of its predecessors.
// A thread has kDcChainPasses blocks of its chunk, kDcChainThreads apart (round 6; one block per thread until then:
// 4096 workgroups of 1024 threads are eight rounds of workgroups over the chip for a few dozen instructions each --
// 20 + 55 us in front of every large frame's token_kernel; with a quarter of the threads they are two).
constexpr int kDcChainThreads = 256;
constexpr int kDcChainBlocks = 1024;                             // blocks per chunk
constexpr int kDcChainPasses = kDcChainBlocks / kDcChainThreads;  // blocks per thread
constexpr int kDcChainChunks = 65536 / kDcChainBlocks;           // per DC group (256 x 256 blocks)
static_assert(kDcChainChunks == 64, "a chunk's carry is worked out by one wave, a lane per preceding chunk");

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

// (a workgroup per chunk: with several chunks per workgroup -- as dc_chain_kernel below -- this kernel, which has no
// histogram to add, got slower: 0.046 / 0.047 / 0.056 / 0.078 Mcycles with 4096 / 2048 / 1024 / 512 workgroups at 16384^2)
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
  DcChunkBlock b[kDcChainPasses];
#pragma unroll
  for (int k = 0; k < kDcChainPasses; k++) b[k] = dc_chunk_block(A, d, chunk * kDcChainBlocks + k * kDcChainThreads + tid);
#pragma unroll
  for (int k = 0; k < kDcChainPasses; k++) {
    const unsigned long long m = __ballot(b[k].first);
    if ((tid & 63) == 0 && m != 0) {
      atomicAdd(&count, (uint32_t)__popcll(m));
      atomicMax(&last_idx, k * kDcChainThreads + (tid & ~63) + 63 - __clzll((long long)m));
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kDcChainPasses; k++)
    if (b[k].first && k * kDcChainThreads + tid == last_idx) last_val = (uint32_t)((b[k].code << 8) | b[k].qfm1);
  __syncthreads();
  if (tid == 0) A.chain_summary[dcg * kDcChainChunks + chunk] = count | (last_val << 16);
}

// (`items` chunks = DC groups of the launch x kDcChainChunks; a workgroup takes the chunks blockIdx.x, + gridDim.x, ...
// and adds its histogram to the frame's ONCE: with a workgroup per chunk 4096 workgroups added ~30 words each to the same
// ~30 addresses, and those 4096 device-scope atomics per address were most of the kernel's 57 us -- round 6)
constexpr int kDcChainGrid = 1024;
__global__ void __launch_bounds__(kDcChainThreads) dc_chain_kernel(const DcArgs A, int items) {
  __shared__ uint32_t hist[16 * 64];  // the two runs only use contexts 3..10
  __shared__ uint32_t wsum[kDcChainBlocks / 64];  // first blocks per 64 consecutive blocks of the chunk
  __shared__ uint16_t compact[kDcChainBlocks + 1];  // (code << 8) | (qf - 1) of the chunk's first blocks
  __shared__ uint32_t carry_rank;
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 64; i += kDcChainThreads) hist[i] = 0;
  const size_t bstride = (size_t)A.g.xsize_blocks;
  for (int item = (int)blockIdx.x; item < items; item += (int)gridDim.x) {
    const int dcg = A.dcg_first + item / kDcChainChunks, chunk = item % kDcChainChunks;
    const uint32_t nac = A.dc_nac[dcg];
    const DcGeom d = dc_geom(A.g, dcg, nac);
    if (chunk * kDcChainBlocks >= d.nb) continue;  // (partial DC groups have fewer chunks; the same for every thread)
    uint8_t* rec = A.records + 3 * A.dc_rec_offset[dcg];
    // (the thread's blocks: requested in front of the carry's loads)
    DcChunkBlock b[kDcChainPasses];
#pragma unroll
    for (int k = 0; k < kDcChainPasses; k++) b[k] = dc_chunk_block(A, d, chunk * kDcChainBlocks + k * kDcChainThreads + tid);
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
    // exclusive rank of the first blocks inside the chunk: blocks 64 (4 k + wave) ... + 63 are pass k of this wave
    uint32_t in_wave[kDcChainPasses];
#pragma unroll
    for (int k = 0; k < kDcChainPasses; k++) {
      const unsigned long long m = __ballot(b[k].first);
      in_wave[k] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[k * (kDcChainThreads / 64) + wave] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t rank[kDcChainPasses];
    {
      uint32_t running = 0;
      int next = 0;
#pragma unroll
      for (int k = 0; k < kDcChainPasses; k++) {
        const int mine = k * (kDcChainThreads / 64) + wave;
        for (; next < mine; next++) running += wsum[next];
        rank[k] = running + in_wave[k];
        if (b[k].first) compact[1 + rank[k]] = (uint16_t)((b[k].code << 8) | b[k].qfm1);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kDcChainPasses; k++) {
      if (!b[k].first) continue;
      const int code = b[k].code, qfm1 = b[k].qfm1;
      const uint16_t prev = compact[rank[k]];  // previous first block (or the carried one)
      const uint32_t grank = carry_rank + rank[k];
      // strategy token (enc_frame.cc:364-383): left = previous code (0 for the very first)
      const int left_s = (grank == 0) ? 0 : (prev >> 8);
      const uint32_t ctx_s = left_s > 11 ? 7 : left_s > 5 ? 8 : left_s > 3 ? 9 : 10;
      put_record(rec, d.pos_strategy + grank, ctx_s, pack_signed(code), hist);
      // quant-field token (:384-408): left = previous (qf-1), initially code of block (0,0)
      const int left_q = (grank == 0) ? code00 : (prev & 0xFF);
      const uint32_t ctx_q = left_q > 11 ? 3 : left_q > 5 ? 4 : left_q > 3 ? 5 : 6;
      put_record(rec, d.pos_qf + grank, ctx_q, pack_signed(qfm1 - left_q), hist);
    }
    __syncthreads();  // (compact, wsum and carry_rank serve the next chunk)
  }
  __syncthreads();
  for (int i = tid; i < 16 * 64; i += kDcChainThreads)
    if (hist[i]) atomicAdd(&A.histogram[i], hist[i]);
}

}  // namespace jxlt_dev

#endif  // JXLT_DC_KERNELS_H_
