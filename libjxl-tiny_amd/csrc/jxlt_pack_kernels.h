// jxlt_pack_kernels.h -- entropy-coded sections at their final bit positions, tile by tile
// (enc_frame.cc:784-800, enc_entropy_code.h:34-42).  Part of jxlt_device.h (include that one).
#ifndef JXLT_PACK_KERNELS_H_
#define JXLT_PACK_KERNELS_H_

#include "jxlt_device_common.h"

namespace jxlt_dev {

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
// (kPackMaxLaunches: jxlt_device_common.h)

// (PackTileInfo: jxlt_device_common.h)

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
  // The writing pass runs as `launches` launches over the tile ranges [launch_t0[i], launch_t0[i + 1]);
  // pack_tile_finalize_kernel files in launch_sec_end[i] how many sections are complete behind launch i (all their
  // tiles lie below launch_t0[i + 1]): the host issues the copy of exactly those sections behind launch i.
  // launch_sec_end may be null (no hand-over by launches).
  uint32_t launches;
  uint32_t launch_t0[kPackMaxLaunches + 1];
  uint32_t* launch_sec_end;         // [launches]
  // Single pass (pack_tile_stream_kernel, round 4): no measuring pass -- every tile learns where its bits start from
  // the tiles in front of it while it runs (tile_state: a 64-bit word per tile, see PackTileState), the sections'
  // bit counts are summed up by their tiles (sec_bits, zeroed by the plan).
  unsigned long long* tile_state;   // [tiles]
  unsigned long long* block_state;  // [tiles / 64 + 1]
  uint32_t* tile_ticket;            // [kPackMaxLaunches]: how many workgroups of a launch have started (zeroed by the plan)
  uint32_t launch_index;            // which of the writing launches this is (its last tile files launch_sec_end)
  uint32_t* lookback_stats;         // optional (JXLT_TRACE_EVENTS): [0] tiles [1] windows looked at [2] reloads of a window
                                    // [3] most windows one tile looked at
};

// What the tiles of the single pass tell the tiles behind them.  Naturally aligned 64-bit words, each written by ONE
// agent-scope store and read by agent-scope loads -- the data carries its own tag, so no flag, no fence (the
// hand-off the CDNA guide calls "data-tagged granule").  Two levels, because what limits such a pass on this machine
// is the round trip between workgroups on different XCDs (store visible behind the other XCD's L2 + load +
// reduction, ~2.4 us measured): a tile can settle its position only once some tile inside its look-back window
// knows its END, so with one level the ENDs advance one window per round trip -- 64 tiles per window gave 26 tiles
// per us, 256 gave ~60, the writing pass of the two-pass form does 120.
//   tile_state[t]   status (bits 63..62) 0 nothing yet | 1 SIZE: bits 31..0 = the tile's bit count, bit 61 = it is the
//                   first tile of a section (its start is rounded up to a byte)
//   block_state[b]  (block b = tiles 64 b ... 64 b + 63)  0 nothing yet | 1 what the block does to the position in
//                   front of it (PackWindowAhead: bit 61 rounds, bits 60..31 pre, bits 30..0 rest), stored by the
//                   block's last tile as soon as it has seen the sizes of the other 63 | 2 the block's END in the
//                   blob, in bits (bits 61..0), stored by the same tile when it knows where it starts
// A tile needs the SIZEs of the tiles in front of it in its own block (started within a microsecond of it) and, of the
// blocks in front, the nearest END and what the blocks between do: 64 blocks = 4096 tiles per window, five times
// what is in flight, so one round trip behind the slowest size settles every position -- no chain of ends.
#ifndef JXLT_THREADFENCE_SYSTEM
#define JXLT_THREADFENCE_SYSTEM() __threadfence_system()
#endif
constexpr unsigned long long kPackStateSize = 1ull << 62, kPackStateEnd = 2ull << 62, kPackStateFirst = 1ull << 61;
constexpr int kPackBlockTiles = 64;
#if defined(__HIP_DEVICE_COMPILE__)
JXLT_DI unsigned long long pack_state_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
JXLT_DI void pack_state_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
JXLT_DI void pack_state_wait() { __builtin_amdgcn_s_sleep(2); }
#else
JXLT_DI unsigned long long pack_state_load(const unsigned long long* p) { return *(const volatile unsigned long long*)p; }
JXLT_DI void pack_state_store(unsigned long long* p, unsigned long long v) { *(volatile unsigned long long*)p = v; }
JXLT_DI void pack_state_wait() {}
#endif
// What a run of tiles does to the position in front of it: p -> round8(p + pre) + rest (a section starts inside the
// run), or p -> p + pre.  32 bits do for 64 tiles of < 2^17 bits and for 64 blocks of < 2^23.
struct PackWindowAhead {
  uint32_t pre, rest, rounds;
};
JXLT_DI PackWindowAhead pack_window_concat(const PackWindowAhead& a, const PackWindowAhead& b) {  // a (older), then b
  PackWindowAhead r;
  const bool ar = a.rounds != 0, br = b.rounds != 0;
  r.rounds = a.rounds | b.rounds;
  r.pre = ar ? a.pre : a.pre + b.pre;
  const uint32_t joined = a.rest + b.pre;
  r.rest = br ? (ar ? ((joined + 7u) & ~7u) + b.rest : b.rest) : (ar ? joined : 0u);
  return r;
}
JXLT_DI unsigned long long pack_ahead_apply(const PackWindowAhead& a, unsigned long long p) {
  return a.rounds ? ((p + a.pre + 7) & ~7ull) + a.rest : p + a.pre;
}
JXLT_DI PackWindowAhead pack_ahead_of_tile(uint32_t bits, bool first) {
  PackWindowAhead one;
  one.pre = first ? 0u : bits;
  one.rest = first ? bits : 0u;
  one.rounds = first ? 1u : 0u;
  return one;
}
JXLT_DI PackWindowAhead pack_ahead_of_state(unsigned long long st, bool block) {
  if (!block) return pack_ahead_of_tile((uint32_t)st, (st & kPackStateFirst) != 0);
  PackWindowAhead one;
  one.rounds = (uint32_t)(st >> 61) & 1u;
  one.pre = (uint32_t)(st >> 31) & 0x3FFFFFFFu;
  one.rest = (uint32_t)st & 0x7FFFFFFFu;
  return one;
}
JXLT_DI unsigned long long pack_block_state_of(const PackWindowAhead& a) {
  return kPackStateSize | ((unsigned long long)(a.rounds & 1u) << 61) | ((unsigned long long)a.pre << 31) | a.rest;
}
// Lanes 63 ... 0 hold runs, oldest in the highest lane: their concatenation, in every lane (six exchange steps; a
// lane-by-lane loop took ~1.3 us per window).
JXLT_DI PackWindowAhead pack_wave_concat(PackWindowAhead v, int lane) {
  for (int d = 1; d < 64; d <<= 1) {
    PackWindowAhead o;
    o.pre = __shfl(v.pre, (lane + d) & 63);
    o.rest = __shfl(v.rest, (lane + d) & 63);
    o.rounds = __shfl(v.rounds, (lane + d) & 63);
    if (lane + d >= 64) o.pre = o.rest = o.rounds = 0u;
    v = pack_window_concat(o, v);
  }
  PackWindowAhead r;
  r.pre = __shfl(v.pre, 0);
  r.rest = __shfl(v.rest, 0);
  r.rounds = __shfl(v.rounds, 0);
  return r;
}

// The states of the tiles in front of `tile` in its own block: lane l holds tile - 1 - l (lanes beyond the block's
// first tile: an empty size).  Requested right behind the tile's own size, looked at behind the packing.
JXLT_DI unsigned long long pack_mates_load(const unsigned long long* tile_state, uint32_t tile, int lane) {
  return lane < (int)(tile & (kPackBlockTiles - 1)) ? pack_state_load(tile_state + (tile - 1 - (uint32_t)lane)) : kPackStateSize;
}
// ... and of the 64 blocks in front of block `nearest + 1`: lane l holds block nearest - l (in front of block 0:
// an END at position 0).
JXLT_DI unsigned long long pack_blocks_load(const unsigned long long* block_state, long long nearest, int lane) {
  const long long idx = nearest - lane;
  return idx >= 0 ? pack_state_load(block_state + idx) : kPackStateEnd;
}
// What the tiles in front of `tile` in its block do to the block's start.  One wave, the same value in every lane;
// waits for mates that have not said their size yet (they run beside this tile).
JXLT_DI PackWindowAhead pack_block_mates(const unsigned long long* tile_state, uint32_t tile, int lane, unsigned long long loaded,
                                         uint32_t* reloads) {
  unsigned long long st = loaded;
  while (__ballot((st >> 62) == 0) != 0) {
    pack_state_wait();
    st = pack_mates_load(tile_state, tile, lane);
    ++*reloads;
  }
  return pack_wave_concat(pack_ahead_of_state(st, false), lane);
}
// Where block `block` starts: the nearest block in front that knows its END, moved through the blocks between.
JXLT_DI unsigned long long pack_block_start(const unsigned long long* block_state, uint32_t block, int lane,
                                            unsigned long long loaded, uint32_t* windows, uint32_t* reloads) {
  PackWindowAhead ahead = {0u, 0u, 0u};  // the blocks between the one that knows its end and `block`
  unsigned long long ahead_pre = 0, ahead_rest = 0;  // (64 bits across windows)
  bool ahead_rounds = false;
  unsigned long long base = 0;
  unsigned long long st = loaded;
  for (long long nearest = (long long)block - 1; nearest >= 0; nearest -= 64) {
    if (nearest != (long long)block - 1) st = pack_blocks_load(block_state, nearest, lane);
    int end_lane;
    for (;;) {
      const unsigned long long knows_end = __ballot((st >> 62) == 2);
      const unsigned long long silent = __ballot((st >> 62) == 0);
      end_lane = knows_end ? (int)__builtin_ctzll(knows_end) : 64;
      if ((silent & (end_lane >= 64 ? ~0ull : ((1ull << end_lane) - 1))) == 0) break;
      pack_state_wait();
      st = pack_blocks_load(block_state, nearest, lane);
      ++*reloads;
    }
    ++*windows;
    PackWindowAhead v = {0u, 0u, 0u};
    if (lane < end_lane) v = pack_ahead_of_state(st, true);
    const PackWindowAhead w = pack_wave_concat(v, lane);
    // w (older), then what has been gathered so far
    if (!ahead_rounds) {
      if (w.rounds) {
        ahead_rounds = true;
        ahead_rest = w.rest + ahead_pre;
        ahead_pre = w.pre;
      } else {
        ahead_pre += w.pre;
      }
    } else if (w.rounds) {
      ahead_rest = ((w.rest + ahead_pre + 7) & ~7ull) + ahead_rest;
      ahead_pre = w.pre;
    } else {
      ahead_pre += w.pre;
    }
    if (end_lane < 64) {
      const uint32_t elo = __shfl((uint32_t)st, end_lane), ehi = __shfl((uint32_t)(st >> 32), end_lane);
      base = (((unsigned long long)ehi << 32) | elo) & ~(3ull << 62);
      break;
    }
  }
  (void)ahead;
  return ahead_rounds ? ((base + ahead_pre + 7) & ~7ull) + ahead_rest : base + ahead_pre;
}

JXLT_DI uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
JXLT_DI uint32_t pack_section_records(const PackTileArgs& A, int sec) {
  return A.sec_rec_count ? A.sec_rec_count[sec] : (uint32_t)(A.sec_rec_offset[sec + 1] - A.sec_rec_offset[sec]);
}

// ---------------------------------------------------------------------------
// Exclusive scan of up to a few ten thousand 32-bit counts into 64-bit offsets (single workgroup):
// offsets[i] = counts[0] + ... + counts[i - 1], offsets[n] = the total.  Every thread owns a contiguous run
// (all of its loads in flight together), one wave scan + one barrier for the runs' totals.
// ---------------------------------------------------------------------------
constexpr int kScanThreads = 1024;
constexpr int kScanMaxPerThread = 32;  // counts per thread and pass: frames of the usual shapes (<= 32 768 sections of a
                                       // kind) take one pass, narrow and tall ones (64 x 16M: 65 536 AC groups) several
__global__ void __launch_bounds__(kScanThreads) group_scan_kernel(const uint32_t* counts, uint64_t* offsets, int n) {
  __shared__ uint64_t wave_total[kScanThreads / 64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kPass = kScanThreads * kScanMaxPerThread;
  uint64_t carry = 0;  // the total of the passes before this one
  for (int base = 0; base < n || base == 0; base += kPass) {
    const int m = imin(n - base, kPass);
    const int per = (m + kScanThreads - 1) / kScanThreads;
    const int beg = base + tid * per, end = imin(base + m, beg + per);
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
    uint64_t run = carry + incl - mine;
    for (int w = 0; w < wave; w++) run += wave_total[w];
    for (int w = 0; w < kScanThreads / 64; w++) carry += wave_total[w];
#pragma unroll
    for (int k = 0; k < kScanMaxPerThread; k++) {
      if (k < per && beg + k < end) {
        offsets[beg + k] = run;
        run += v[k];
      }
    }
    __syncthreads();  // (the next pass overwrites wave_total)
  }
  if (tid == 0) offsets[n] = carry;
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
  if (A.tile_state) {  // (single pass: the tiles sum the section's bits up themselves)
    A.sec_bits[s] = 0;
    if (s == 0 && A.launch_sec_end)
      for (int i = 0; i < kPackMaxLaunches; i++) {
        A.launch_sec_end[i] = 0xFFFFFFFFu;  // "no tile in this launch"
        A.tile_ticket[i] = 0;
      }
  }
  for (uint32_t t = t0; t < t1; t++) {
    if (A.tile_state) {
      A.tile_state[t] = 0;
      if ((t & (kPackBlockTiles - 1)) == 0) A.block_state[t / kPackBlockTiles] = 0;
    }
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

// The tile plan of up to 1024 sections in ONE launch (round 6): count, exclusive scan and plan by one workgroup, a
// thread per section -- for the frames of a batch the three launches of the plan (x 2 kinds) were a seventh of a
// frame's ~28 launches, and many small launches from several host threads are what a batch of small frames is bound by.
constexpr int kPackPlanSmallSections = 1024;
JXLT_DI void pack_tile_plan_small_body(const PackTileArgs& A, uint64_t* tile_base_out) {
  __shared__ uint32_t wave_total[kPackPlanSmallSections / 64];
  const int s = (int)threadIdx.x, lane = s & 63, wave = s >> 6;
  const uint32_t cnt = s < A.nsec ? pack_section_records(A, s) : 0u;
  const uint32_t tiles = (cnt + kPackTile - 1) / kPackTile;
  uint32_t incl = tiles;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  uint32_t t0 = incl - tiles;
  for (int w = 0; w < wave; w++) t0 += wave_total[w];
  if (s < A.nsec) {
    A.sec_tiles[s] = tiles;
    tile_base_out[s] = t0;
    if (s + 1 == A.nsec) tile_base_out[A.nsec] = t0 + tiles;
    const uint32_t t1 = t0 + tiles;
    const uint64_t rec0 = A.sec_rec_offset[s];
    if (A.tile_state) {
      A.sec_bits[s] = 0;
      if (s == 0 && A.launch_sec_end)
        for (int i = 0; i < kPackMaxLaunches; i++) {
          A.launch_sec_end[i] = 0xFFFFFFFFu;
          A.tile_ticket[i] = 0;
        }
    }
    for (uint32_t t = t0; t < t1; t++) {
      if (A.tile_state) {
        A.tile_state[t] = 0;
        if ((t & (kPackBlockTiles - 1)) == 0) A.block_state[t / kPackBlockTiles] = 0;
      }
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
}
__global__ void __launch_bounds__(kPackPlanSmallSections) pack_tile_plan_small_kernel(const PackTileArgs A, uint64_t* tile_base_out) {
  pack_tile_plan_small_body(A, tile_base_out);
}
// ... and the plans of BOTH kinds of section in one launch (workgroup 0: the DC-group sections, 1: the AC sections)
__global__ void __launch_bounds__(kPackPlanSmallSections) pack_tile_plan_small2_kernel(const PackTileArgs A0, uint64_t* base0,
                                                                                        const PackTileArgs A1, uint64_t* base1) {
  if (blockIdx.x == 0) pack_tile_plan_small_body(A0, base0);
  else pack_tile_plan_small_body(A1, base1);
}

// In front of a single pass (round 6): the code table fetched from the host's page-locked copy AND the blob zeroed, in
// one launch instead of a publish kernel + the runtime's fill kernel(s).  Workgroup 0 fetches the table, all zero
// their share of the blob (16-byte stores; `zero_bytes` is a multiple of 16).
constexpr int kPackPrepareThreads = 256;
__global__ void __launch_bounds__(kPackPrepareThreads) pack_prepare_kernel(const uint32_t* table_src, uint32_t* table_dst,
                                                                           uint8_t* blob, unsigned long long zero_bytes) {
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x == 0) {
    const uint4* src4 = reinterpret_cast<const uint4*>(table_src);
    uint4* dst4 = reinterpret_cast<uint4*>(table_dst);
    for (uint32_t i = tid; i < 64 * 64 / 4; i += kPackPrepareThreads) dst4[i] = src4[i];
  }
  uint4* out = reinterpret_cast<uint4*>(blob);
  const unsigned long long n16 = zero_bytes >> 4;
  uint4 zero;
  zero.x = zero.y = zero.z = zero.w = 0u;
  for (unsigned long long i = (unsigned long long)blockIdx.x * kPackPrepareThreads + tid; i < n16;
       i += (unsigned long long)gridDim.x * kPackPrepareThreads)
    out[i] = zero;
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
// LDS layouts of the code table.  A tile's records use few (context, symbol) pairs, mostly the small symbols of
// many contexts: with rows of 64 dwords those all sit in the banks of symbols 0-3.  The 32-bit table rotates every
// row by its context (symbol 0 of the 64 contexts: 64 banks; 366 -> 129 conflict cycles per wave, same time).  (The
// byte table of the measuring pass stays context-major: symbol-major measured slower, its fill conflicts.)
JXLT_DI int pack_table_slot(uint32_t ctx, uint32_t sym) { return (int)(ctx * 64 + ((sym + ctx) & 63u)); }
// Bit length and bits of one record, without branches (a tile mixes raw-bit records, small values and values with
// extra bits lane by lane: as branches every record cost the wave all three paths plus the exec-mask bookkeeping).
// Symbol and extra-bit count of a value >= 16 come from its float form (hybrid_uint_symbol, jxlt_device_common.h);
// records that are not there (beyond a section's last one) are turned into "zero raw bits" by the callers.
JXLT_DI void pack_split_value(uint32_t value, uint32_t* sym, uint32_t* nbits) {  // token.h:32-48
  const uint32_t fbits = __float_as_uint((float)value);
  const bool big = value >= 16;
  *sym = big ? (fbits >> 21) - (127u << 2) : value;
  *nbits = big ? (fbits >> 23) - 129u : 0u;
}
JXLT_DI void pack_bits_of(uint32_t rec24, const uint32_t* table, uint32_t* nb, uint32_t* data) {
  const uint32_t ctx = rec24 & 0xFFu, value = rec24 >> 8;
  uint32_t sym, nbits;
  pack_split_value(value, &sym, &nbits);
  const uint32_t extra = __builtin_amdgcn_ubfe(value, 0u, nbits);
  const uint32_t e = table[pack_table_slot(ctx & 63u, sym)];
  const uint32_t depth = e >> 16;
  const bool raw = ctx >= 128;
  *nb = raw ? ctx - 128u : depth + nbits;
  *data = raw ? value : ((e & 0xFFFFu) | (extra << depth));
}
// The measuring pass looks a record's length up in ONE byte table over the whole context byte: rows 0-63 hold
// depth + extra bits of (context, symbol), rows 128-255 the raw-bit records' own count (context - 128) whatever
// the "symbol", rows 64-127 are never addressed.
constexpr int kPackLengthRows = 256;
JXLT_DI uint32_t pack_length_of(uint32_t rec24, const uint8_t* length) {
  const uint32_t ctx = rec24 & 0xFFu, value = rec24 >> 8;
  const uint32_t fbits = __float_as_uint((float)value);
  const uint32_t sym = value >= 16 ? (fbits >> 21) - (127u << 2) : value;
  return length[ctx * 64 + sym];
}
constexpr uint32_t kPackNoRecord = 0x80u;  // a raw-bit record of zero bits

// Consecutive tiles per workgroup (amortises the table load).  The writing pass runs as several launches (the
// copies to the host follow launch by launch), each of which ends with a partly empty machine for as long as a
// workgroup lives: two tiles per workgroup there (0.735 -> 0.64 Mcycles per 16384^2 frame; one tile: 0.685).
constexpr int kPackWriteTilesPerGroup = 2;
constexpr int kPackMeasureTilesPerGroup = 4;

// A sum does not care about the order of its terms: every thread takes four consecutive records (12 bytes, three
// unaligned dword loads straight from global memory -- the lanes of a wave cover 768 contiguous bytes) instead of
// its eight of the staged tile, so there is no staging area, and no barrier until the workgroup's tiles are summed.
// All of a workgroup's loads are in flight before the first length is looked up.  (Same speed as the staged version
// it replaced, 0.283 against 0.288 Mcycles per 16384^2 frame: the pass is bound by its instructions per record.)
__global__ void __launch_bounds__(kPackThreads) pack_tile_measure_kernel(const PackTileArgs A) {
  __shared__ alignas(16) uint8_t length[kPackLengthRows * 64];
  __shared__ uint32_t total[kPackMeasureTilesPerGroup];
  constexpr int kChunk = 4;                                            // records per thread and step
  constexpr int kSteps = kPackTile / (kPackThreads * kChunk);          // steps per tile
  static_assert(kSteps * kPackThreads * kChunk == kPackTile, "whole steps");
  const int tid = (int)threadIdx.x;
  const uint32_t ntiles_all = umin32((uint32_t)A.tile_base[A.nsec], A.tile_end);
  const uint32_t first = A.tile_first + blockIdx.x * kPackMeasureTilesPerGroup;
  if (first >= ntiles_all) return;
  uint32_t w[kPackMeasureTilesPerGroup][kSteps][3];
  int nrec[kPackMeasureTilesPerGroup];
#pragma unroll
  for (int k = 0; k < kPackMeasureTilesPerGroup; k++) {
    const uint32_t tile = umin32(first + k, ntiles_all - 1);
    const PackTileInfo info = A.tile_info[tile];
    nrec[k] = first + k < ntiles_all ? (int)(info.n_last & 0x7FFFFFFFu) : 0;
    const uint8_t* src = A.records + 3 * info.rec_first;
#pragma unroll
    for (int it = 0; it < kSteps; it++) {
      const int byte0 = 3 * kChunk * (tid + it * kPackThreads);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        uint32_t v = 0;
        if (byte0 + 4 * q < 3 * nrec[k]) __builtin_memcpy(&v, src + byte0 + 4 * q, 4);
        w[k][it][q] = v;
      }
    }
  }
  {  // the length table, four entries (one dword) at a time
    uint32_t* const length_w = reinterpret_cast<uint32_t*>(length);
    for (int i = tid; i < 64 * 64 / 4; i += kPackThreads) {  // rows 0-63: depth + extra bits of the symbol
      uint32_t packed = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t sym = (uint32_t)(4 * i + q) & 63u;
        const uint32_t nbits = sym >= 16 ? (sym >> 2) - 2u : 0u;
        packed |= ((A.code_table[4 * i + q] >> 16) + nbits) << (8 * q);
      }
      length_w[i] = packed;
    }
    for (int i = 64 * 64 / 4 + tid; i < kPackLengthRows * 64 / 4; i += kPackThreads) {
      const uint32_t ctx = (uint32_t)i >> 4;  // (16 dwords per row)
      length_w[i] = ctx >= 128 ? (ctx - 128u) * 0x01010101u : 0u;
    }
  }
  if (tid < kPackMeasureTilesPerGroup) total[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPackMeasureTilesPerGroup; k++) {
    if (first + k >= ntiles_all) break;
    uint32_t mine = 0;
#pragma unroll
    for (int it = 0; it < kSteps; it++) {
      const int nvalid = nrec[k] - kChunk * (tid + it * kPackThreads);
#pragma unroll
      for (int j = 0; j < kChunk; j++) {
        const int byte = 3 * j;
        const uint32_t hi = byte >> 2 < 2 ? w[k][it][(byte >> 2) + 1] : 0u;
        const uint32_t rec24 = __builtin_amdgcn_alignbyte(hi, w[k][it][byte >> 2], (uint32_t)(byte & 3)) & 0xFFFFFFu;
        mine += pack_length_of(j < nvalid ? rec24 : kPackNoRecord, length);
      }
    }
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((tid & 63) == 0) atomicAdd(&total[k], mine);
  }
  __syncthreads();
  if (tid < kPackMeasureTilesPerGroup && first + tid < ntiles_all) A.tile_bits[first + tid] = total[tid];
}

// A wave per section: the bit offsets of its tiles by wave-wide prefix sums, 64 tiles at a time.  (A thread per
// section walking its tiles -- ~100 dependent load / store pairs for a DC-group section -- took 21 us per frame.)
constexpr int kPackOffsetsSectionsPerGroup = 4;
__global__ void __launch_bounds__(64 * kPackOffsetsSectionsPerGroup) pack_tile_offsets_kernel(const PackTileArgs A) {
  const int s = (int)(blockIdx.x * kPackOffsetsSectionsPerGroup + (threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63);
  if (s >= A.nsec) return;
  const uint32_t t0 = (uint32_t)A.tile_base[s], t1 = (uint32_t)A.tile_base[s + 1];
  uint32_t off = 0;
  for (uint32_t base = t0; base < t1; base += 64) {
    const uint32_t t = base + (uint32_t)lane;
    const uint32_t v = t < t1 ? A.tile_bits[t] : 0u;
    uint32_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    if (t < t1) A.tile_info[t].bit_pos = off + incl - v;
    off += __shfl(incl, 63);
  }
  if (lane == 0) {
    A.sec_bits[s] = off;
    A.sec_bytes[s] = (off + 7) >> 3;
  }
}

__global__ void __launch_bounds__(256) pack_tile_finalize_kernel(const PackTileArgs A) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (uint32_t)A.tile_base[A.nsec]) {
    // (no tile at all -- every section is empty: every launch "completes" all of them)
    if (t == 0 && A.launch_sec_end)
      for (uint32_t i = 0; i < A.launches; i++) A.launch_sec_end[i] = (uint32_t)A.nsec;
    return;
  }
  PackTileInfo info = A.tile_info[t];
  const uint32_t sec = (uint32_t)info.sec_start_bit;
  // The sections that are complete behind writing launch i: everything in front of the section that holds the
  // first tile of launch i + 1 (the last launch completes them all).  Filed by the thread of that tile; boundaries
  // at or beyond the tile count by the thread of the last tile.
  if (A.launch_sec_end) {
    const uint32_t ntiles = (uint32_t)A.tile_base[A.nsec];
    for (uint32_t i = 0; i < A.launches; i++) {
      const uint32_t b = A.launch_t0[i + 1];
      if (i + 1 == A.launches || b >= ntiles) {
        if (t == ntiles - 1) A.launch_sec_end[i] = (uint32_t)A.nsec;
      } else if (b == t) {
        A.launch_sec_end[i] = sec;
      }
    }
  }
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

// Per tile: records -> (length, bits) per thread, workgroup scan of the lengths, every record OR-ed into the LDS
// window at its bit position, window -> blob.  Two barriers per tile: the next tile's records go to the staging
// area right behind the scan's barrier (every thread has its own records in registers by then), and the window is
// cleared by the threads that store it.
// kStream: the single pass -- the tile's bit position is not in its descriptor (there was no measuring pass) but
// comes from the tiles in front of it (pack_tile_start), between the scan of the lengths and the packing; the blob
// was zeroed as a whole before the launch (the dwords in which tiles meet are OR-ed into, as in the two-pass form).
// (ONE tile per workgroup in the single pass: a workgroup's second tile says its size only when the first has its
// position, and the first tile of the next workgroup waits for that size -- with two tiles per workgroup the whole
// launch became one chain, 46 ms instead of 0.3 for the 16384^2 frame)
constexpr int kPackStreamTilesPerGroup = 1;
template <bool kStream, int kTilesPerGroup>
JXLT_DI void pack_tile_write_body(const PackTileArgs& A) {
  __shared__ uint32_t table[64 * 64];
  __shared__ alignas(16) uint32_t stage[kPackTile * 3 / 4 + 4];
  __shared__ alignas(16) uint32_t window[kPackWindowWords];
  __shared__ uint32_t wave_sum[kPackThreads / 64];
  __shared__ unsigned long long stream_start;
  const int tid = (int)threadIdx.x;
  const uint32_t ntiles_all = umin32((uint32_t)A.tile_base[A.nsec], A.tile_end);
  // Which tile: the two-pass form's writing pass takes them by workgroup index.  The single pass hands them out in the
  // order in which workgroups START (a ticket per launch): a tile waits for the tiles in front of it, and with
  // tickets those belong to workgroups that are running or done, whatever else shares the device.  By workgroup
  // index a tile can wait for one that has not been dispatched because every slot of its XCD is held by waiting tiles
  // of ANOTHER context's single pass, whose own predecessors wait for a slot the same way: two contexts packing
  // 16384^2 halves on one GPU took 12 SECONDS per frame (tools/shard_overhead.py, first version).
  __shared__ uint32_t ticket;
  uint32_t first_tile;
  PackTileInfo cur, nxt;
  PackStagedLoads loads;
  if constexpr (kStream) {
    if (tid == 0) ticket = A.tile_first + atomicAdd(&A.tile_ticket[A.launch_index], 1u);
    // (the code table does not depend on the tile: its loads are in flight while the ticket arrives)
    uint32_t tl[64 * 64 / kPackThreads];
#pragma unroll
    for (int q = 0; q < 64 * 64 / kPackThreads; q++) tl[q] = A.code_table[tid + q * kPackThreads];
    __syncthreads();
    first_tile = ticket;
    if (first_tile >= ntiles_all) return;
    cur = A.tile_info[first_tile];
    nxt = cur;
    pack_request_tile(A.records + 3 * cur.rec_first, (int)(cur.n_last & 0x7FFFFFFFu), tid, &loads);
#pragma unroll
    for (int q = 0; q < 64 * 64 / kPackThreads; q++) {
      const uint32_t i = (uint32_t)(tid + q * kPackThreads);
      table[pack_table_slot(i >> 6, i & 63u)] = tl[q];
    }
  } else {
    first_tile = A.tile_first + blockIdx.x * kTilesPerGroup;
    if (first_tile >= ntiles_all) return;
    // The descriptors of the first two tiles and the first tile's records are requested in front of the code table.
    cur = A.tile_info[first_tile];
    nxt = A.tile_info[umin32(first_tile + 1, ntiles_all - 1)];
    pack_request_tile(A.records + 3 * cur.rec_first, (int)(cur.n_last & 0x7FFFFFFFu), tid, &loads);
    // (all eight loads of the code table in flight before the first LDS store)
    uint32_t tl[64 * 64 / kPackThreads];
#pragma unroll
    for (int q = 0; q < 64 * 64 / kPackThreads; q++) tl[q] = A.code_table[tid + q * kPackThreads];
#pragma unroll
    for (int q = 0; q < 64 * 64 / kPackThreads; q++) {
      const uint32_t i = (uint32_t)(tid + q * kPackThreads);
      table[pack_table_slot(i >> 6, i & 63u)] = tl[q];
    }
  }
  const uint32_t last_tile = ntiles_all - 1;
  for (int i = tid; i < kPackWindowWords; i += kPackThreads) window[i] = 0u;
  pack_store_tile(loads, (int)(cur.n_last & 0x7FFFFFFFu), stage, tid);
  if (kTilesPerGroup > 1 && first_tile + 1 < ntiles_all)
    pack_request_tile(A.records + 3 * nxt.rec_first, (int)(nxt.n_last & 0x7FFFFFFFu), tid, &loads);
  uint32_t* outw = reinterpret_cast<uint32_t*>(A.out);
  __syncthreads();  // table, first tile's records, clear window
  for (int kt = 0; kt < kTilesPerGroup; kt++) {
    const uint32_t tile = first_tile + kt;
    if (tile >= ntiles_all) break;
    const bool has_next = kt + 1 < kTilesPerGroup && tile + 1 < ntiles_all;
    const bool has_next2 = kt + 2 < kTilesPerGroup && tile + 2 < ntiles_all;
    // (the descriptor of the tile after the next: needed behind the scan, requested here)
    const PackTileInfo nxt2 = A.tile_info[umin32(tile + 2, last_tile)];
    const int n = (int)(cur.n_last & 0x7FFFFFFFu);
    // pass 1: length and bits of this thread's records
    PackThreadRecords recs;
    pack_load_thread_records(stage, tid, &recs);
    uint32_t nb[kPackPerThread];
    uint32_t data[kPackPerThread];
    uint32_t mine = 0;
    const int nvalid = n - tid * kPackPerThread;
#pragma unroll
    for (int j = 0; j < kPackPerThread; j++) {
      pack_bits_of(j < nvalid ? pack_thread_record(recs, j) : kPackNoRecord, table, &nb[j], &data[j]);
      mine += nb[j];
    }
    // exclusive prefix of `mine` over the workgroup: wave scan + per-wave totals
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(incl, d);
      if ((tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();  // every thread holds its records: the staging area is free
    if (has_next) pack_store_tile(loads, (int)(nxt.n_last & 0x7FFFFFFFu), stage, tid);
    if (has_next2) pack_request_tile(A.records + 3 * nxt2.rec_first, (int)(nxt2.n_last & 0x7FFFFFFFu), tid, &loads);
    uint32_t wave_base = 0, tile_bits = 0;
#pragma unroll
    for (int w = 0; w < kPackThreads / 64; w++) {
      const uint32_t v = wave_sum[w];
      if (w < (tid >> 6)) wave_base += v;
      tile_bits += v;
    }
    uint64_t pos_bit = cur.bit_pos;  // where this tile's bits start
    [[maybe_unused]] unsigned long long mates_loaded = 0, blocks_loaded = 0;
    if constexpr (kStream) {
      static_assert(kTilesPerGroup == 1, "the single pass packs one tile per workgroup");
      // Wave 0 tells the tiles behind this one its size and asks for the states of the tiles in front of it in its
      // block and of the blocks in front; the answers are looked at behind the packing (which does not need the
      // position: the window is packed from bit 0 on and shifted to the position's low five bits when it is stored).
      if (tid < 64) {
        if (tid == 0) {
          if (tile_bits) atomicAdd(&A.sec_bits[(uint32_t)cur.sec_start_bit], tile_bits);
          pack_state_store(A.tile_state + tile, kPackStateSize | (cur.before == 0 ? kPackStateFirst : 0ull) | tile_bits);
        }
        mates_loaded = pack_mates_load(A.tile_state, tile, tid);
        blocks_loaded = pack_blocks_load(A.block_state, (long long)(tile / kPackBlockTiles) - 1, tid);
      }
      pos_bit = 0;
    }
    const uint32_t lead = (uint32_t)(pos_bit & 31u);
    const uint64_t word0 = pos_bit >> 5;
    // pass 2: the thread's records concatenated in a register pair, completed dwords OR-ed into the window
    // (LDS atomics are the expensive part: a record at a time -- 16 per thread -- made the kernel 45 % slower)
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
    if constexpr (kStream) {
      if (tid < 64) {
        const uint32_t sec = (uint32_t)cur.sec_start_bit;  // (the plan's section index: nothing has replaced it)
        const bool first = cur.before == 0;
        const uint32_t block = tile / kPackBlockTiles;
        const bool closes_block = (tile & (kPackBlockTiles - 1)) == kPackBlockTiles - 1;
        uint32_t windows = 0, reloads = 0;
        const PackWindowAhead mates = pack_block_mates(A.tile_state, tile, tid, mates_loaded, &reloads);
        // (the block's last tile says what the block does to a position as soon as it can: the blocks behind do not
        // have to wait until it knows where it starts)
        if (closes_block && tid == 0)
          pack_state_store(A.block_state + block, pack_block_state_of(pack_window_concat(mates, pack_ahead_of_tile(tile_bits, first))));
        const unsigned long long block_start = pack_block_start(A.block_state, block, tid, blocks_loaded, &windows, &reloads);
        unsigned long long start = pack_ahead_apply(mates, block_start);
        if (first) start = (start + 7) & ~7ull;
        if (tid == 0) {
          if (closes_block) pack_state_store(A.block_state + block, kPackStateEnd | (start + tile_bits));
          stream_start = start;
          // (the launch's last tile says which sections the launch has completed)
          if (tile == last_tile && A.launch_sec_end) A.launch_sec_end[A.launch_index] = (cur.n_last >> 31) ? sec + 1 : sec;
          if (A.lookback_stats) {
            atomicAdd(&A.lookback_stats[0], 1u);
            atomicAdd(&A.lookback_stats[1], windows);
            atomicAdd(&A.lookback_stats[2], reloads);
            atomicMax(&A.lookback_stats[3], windows);
          }
        }
      }
      __syncthreads();  // window complete, position known
      const unsigned long long start = stream_start;
      const uint32_t shift = (uint32_t)(start & 31u);
      const uint64_t first_word = start >> 5;
      const uint32_t end_bits = shift + tile_bits;
      const uint32_t nwords = (end_bits + 31) >> 5;
      for (uint32_t i = tid; i < nwords; i += kPackThreads) {
        const uint32_t hi = window[i], lo = i ? window[i - 1] : 0u;
        const uint32_t v = shift ? (hi << shift) | (lo >> (32u - shift)) : hi;
        if (i == 0 || (i + 1 == nwords && (end_bits & 31u) != 0)) {
          if (v) atomicOr(&outw[first_word + i], v);
        } else {
          outw[first_word + i] = v;
        }
      }
      break;  // (one tile per workgroup: nothing to clear, nothing to hand on)
    }
    __syncthreads();  // window complete; next tile's records staged
    // stores: the dwords the tile covers completely with plain stores; its first and its last dword, which it
    // may share with its neighbours (tiles of the same section, or the byte-aligned neighbour sections), are
    // OR-ed into memory that pack_tile_finalize_kernel zeroed.  Whoever stores a dword clears it for the next tile.
    const uint32_t end_bits = lead + tile_bits;
    const uint32_t nwords = (end_bits + 31) >> 5;  // dwords the tile touches
    for (uint32_t i = tid; i < nwords; i += kPackThreads) {
      const uint32_t v = window[i];
      window[i] = 0u;
      if (i == 0 || (i + 1 == nwords && (end_bits & 31u) != 0)) {
        if (v) atomicOr(&outw[word0 + i], v);
      } else {
        outw[word0 + i] = v;
      }
    }
    cur = nxt;
    nxt = nxt2;
  }
}
__global__ void __launch_bounds__(kPackThreads) pack_tile_write_kernel(const PackTileArgs A) { pack_tile_write_body<false, kPackWriteTilesPerGroup>(A); }
__global__ void __launch_bounds__(kPackThreads) pack_tile_stream_kernel(const PackTileArgs A) { pack_tile_write_body<true, kPackStreamTilesPerGroup>(A); }

}  // namespace jxlt_dev

#endif  // JXLT_PACK_KERNELS_H_
