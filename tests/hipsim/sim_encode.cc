// TEST INFRASTRUCTURE: runs the product's device kernels (jxlt_device.h) on the
// CPU execution model in hip/hip_runtime.h and exposes the result through a C
// function with the same output structure as the oracle, so tests can diff them.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../libjxl-tiny_amd/csrc/jxlt_device.h"
#include "../../libjxl-tiny_amd/csrc/jxlt_host_tables.h"

using namespace jxlt_dev;

extern "C" {

struct sim_result {
  size_t xsize_blocks, ysize_blocks, xsize_tiles, ysize_tiles, num_groups;
  int16_t* quant_dc[3];
  uint8_t* raw_quant;
  uint8_t* strategy;
  int8_t* ytox;
  int8_t* ytob;
  uint8_t* tokens;
  uint64_t* group_tok_offset;  // [num_groups+1], in tokens
  float* xyb[3];
  float* qf;
  float* mask;
  float* ent8;
  uint32_t* histogram;  // [2][64*64]: AC, DC
  uint8_t* dc_records;  // DC-group record streams, fixed stride
  uint64_t* dc_rec_offset;  // [ndc + 1]
  uint32_t* dc_count;   // [ndc]
  size_t num_dc_groups;
  uint32_t exact_reruns;  // tiles that were redone with computed roots (tile*_kernel_redo)
  uint32_t unsupported;   // TileArgs::unsupported: non-zero = the C ABI would answer JXLT_ERR_UNSUPPORTED
};

__attribute__((visibility("default"))) void sim_free(sim_result* r);
static float g_sim_strategy_distance = 0.0f;
__attribute__((visibility("default"))) void sim_set_strategy_distance(float d) { g_sim_strategy_distance = d; }

__attribute__((visibility("default"))) int sim_encode(const float* const planes[3], size_t pitch_floats, size_t xsize, size_t ysize,
               float distance, float scale, float inv_scale, float scale_dc, uint32_t x_qm_scale,
               uint32_t flags, sim_result* r) {
  const FrameGeom g = MakeGeom(xsize, ysize);
  DeviceTables* tab = new DeviceTables;
  BuildDeviceTables(scale, tab);
  const size_t nblocks = (size_t)g.xsize_blocks * g.ysize_blocks;
  const size_t ntiles = (size_t)g.xsize_tiles * g.ysize_tiles;
  const size_t ngroups = (size_t)g.xsize_groups * g.ysize_groups;
  memset(r, 0, sizeof(*r));
  r->xsize_blocks = g.xsize_blocks;
  r->ysize_blocks = g.ysize_blocks;
  r->xsize_tiles = g.xsize_tiles;
  r->ysize_tiles = g.ysize_tiles;
  r->num_groups = ngroups;
  TileArgs A;
  memset(&A, 0, sizeof(A));
  for (int c = 0; c < 3; c++) {
    A.planes[c] = planes[c];
    A.quant_dc[c] = r->quant_dc[c] = (int16_t*)calloc(nblocks, 2);
    A.nzgrid[c] = (uint8_t*)calloc(nblocks, 1);
    A.dbg_xyb[c] = r->xyb[c] = (float*)calloc(nblocks * 64, 4);
  }
  A.pitch = (ptrdiff_t)pitch_floats;
  A.pix_stride = 1;
  A.byteswap = 0;
  if (flags & 0x100u) {
    // planes[0] is a raw PFM payload: interleaved RGB, bottom row first (read_pfm.cc:199-209)
    for (int c = 0; c < 3; c++) A.planes[c] = planes[0] + (ysize - 1) * xsize * 3 + c;
    A.pitch = -(ptrdiff_t)(xsize * 3);
    A.pix_stride = 3;
    A.byteswap = (flags & 0x200u) ? 1 : 0;
  }
  A.g = g;
  A.distance = distance;
  A.strategy_distance = g_sim_strategy_distance > 0.0f ? g_sim_strategy_distance : distance;
  A.scale = scale;
  A.inv_scale = inv_scale;
  A.scale_dc = scale_dc;
  A.x_qm_mul = XQmMultiplier(x_qm_scale);
  SetStrategyScalars(&A);
  A.flags = flags & ~0x4000u;  // (0x4000 selects a kernel here, it is not a kernel flag)
  A.tab = tab;
  A.raw_quant = r->raw_quant = (uint8_t*)calloc(nblocks, 1);
  A.strategy = r->strategy = (uint8_t*)calloc(nblocks, 1);
  A.ytox = r->ytox = (int8_t*)calloc(ntiles, 1);
  A.ytob = r->ytob = (int8_t*)calloc(ntiles, 1);
  A.blk_nz = (uint8_t*)calloc(nblocks * 3, 1);
  A.blk_nscan = (uint8_t*)calloc(nblocks * 3, 1);
  A.blk_nzmask = (unsigned long long*)calloc(nblocks * 6, 8);
  A.coef_scan = (int16_t*)malloc(nblocks * 3 * 64 * 2);
  memset(A.coef_scan, 0xBB, nblocks * 3 * 64 * 2);  // unwritten positions must never be consumed
  A.group_ntok = (uint32_t*)calloc(ngroups, 4);
  const size_t ndc = ((xsize + 2047) / 2048) * ((ysize + 2047) / 2048);
  A.dc_nac = (uint32_t*)calloc(ndc, 4);
  A.dbg_qf = r->qf = (float*)calloc(nblocks, 4);
  A.dbg_mask = r->mask = (float*)calloc(nblocks, 4);
  const size_t ncells = ((size_t)g.xsize_blocks / 2 + 1) * ((size_t)g.ysize_blocks / 2 + 1);
  A.dbg_ent8 = r->ent8 = (float*)malloc(ncells * 8 * 4);
  for (size_t i = 0; i < ncells * 8; i++) A.dbg_ent8[i] = __builtin_nanf("");

  // as jxlt_capi_encode.hip: the table-root kernel, then the redo kernel for the tiles it filed (a quantised magnitude
  // beyond the table), per launch; one launch per row of DC groups with the slab arguments of the product
  // (jxlt_host_tables.h: SlabTileArgs)
  const size_t rows_per_slab = 2048, nsl = (ysize + rows_per_slab - 1) / rows_per_slab;
  std::vector<uint32_t> lut_overflow(nsl, 0), overflow_tiles(ntiles, 0xFFFFFFFFu);
  A.lut_overflow = lut_overflow.data();
  A.overflow_tiles = overflow_tiles.data();
  uint32_t unsupported = 0;
  A.unsupported = &unsupported;
  for (size_t sl = 0; sl < nsl; sl++) {
    const size_t y0 = sl * rows_per_slab, rows = std::min(rows_per_slab, ysize - y0);
    const TileArgs S = nsl == 1 ? A : SlabTileArgs(A, y0, rows, A.pitch, sl);
    const dim3 grid((unsigned)((size_t)S.g.xsize_tiles * S.g.ysize_tiles));
    const dim3 redo_grid(std::min<unsigned>(grid.x, 3u));  // (fewer workgroups than the product: the loop over the list runs)
    // 0x800: the production variant (no debug outputs: r->xyb, qf, mask, ent8 stay as initialised)
    if (flags & 0x800u) hipsim::launch(tile12_kernel, grid, dim3(kTile12Threads), S);
    else hipsim::launch(tile12_kernel_debug, grid, dim3(kTile12Threads), S);
    hipsim::launch(tile12_kernel_redo, redo_grid, dim3(kTile12Threads), S);
    r->exact_reruns += lut_overflow[sl];  // tiles redone
  }
  r->unsupported = unsupported;

  r->group_tok_offset = (uint64_t*)calloc(ngroups + 1, 8);
  // (token_kernel finds every group's token offset itself; the total sizes the token buffer here)
  const size_t nslabs = (ysize + 2047) / 2048;
  uint64_t total = 0;
  for (size_t i = 0; i < ngroups; i++) total += A.group_ntok[i];
  r->tokens = (uint8_t*)calloc(total * 3 + 1, 1);
  TokenArgs K;
  memset(&K, 0, sizeof(K));
  K.g = g;
  K.tab = tab;
  K.strategy = A.strategy;
  for (int c = 0; c < 3; c++) K.nzgrid[c] = A.nzgrid[c];
  K.blk_nz = A.blk_nz;
  K.blk_nscan = A.blk_nscan;
  K.blk_nzmask = A.blk_nzmask;
  K.coef_scan = A.coef_scan;
  K.group_ntok = A.group_ntok;
  K.group_tok_offset = r->group_tok_offset;
  K.tokens = r->tokens;
  K.histogram = r->histogram = (uint32_t*)calloc(2 * 64 * 64, 4);
  for (size_t sl = 0; sl < nslabs; sl++) {
    const size_t y0 = sl * 2048, rows = std::min<size_t>(2048, ysize - y0);
    K.group_first = (int)((y0 / 256) * (size_t)g.xsize_groups);
    const dim3 tok_grid((unsigned)(((rows + 255) / 256) * (size_t)g.xsize_groups));
    // (0x4000: the variant with 64-bit coefficient indices, which the product launches for frames above 1.43 Gpixel)
    if (flags & 0x4000u) hipsim::launch(token_kernel_wide, tok_grid, dim3(kTokenThreads), K);
    else hipsim::launch(token_kernel, tok_grid, dim3(kTokenThreads), K);
  }

  {
    const size_t kDcStride = 6 * 65536 + 2 * 1024 + 8;
    r->num_dc_groups = ndc;
    r->dc_rec_offset = (uint64_t*)calloc(ndc + 1, 8);
    for (size_t i = 0; i <= ndc; i++) r->dc_rec_offset[i] = i * kDcStride;
    r->dc_records = (uint8_t*)malloc(ndc * kDcStride * 3);
    memset(r->dc_records, 0xEE, ndc * kDcStride * 3);
    r->dc_count = (uint32_t*)calloc(ndc, 4);
    DcArgs D;
    memset(&D, 0, sizeof(D));
    D.g = g;
    D.tab = tab;
    for (int c = 0; c < 3; c++) D.quant_dc[c] = A.quant_dc[c];
    D.raw_quant = A.raw_quant;
    D.strategy = A.strategy;
    D.ytox = A.ytox;
    D.ytob = A.ytob;
    D.dc_nac = A.dc_nac;
    D.dc_rec_offset = r->dc_rec_offset;
    D.records = r->dc_records;
    D.dc_count = r->dc_count;
    D.histogram = r->histogram + 64 * 64;
    std::vector<uint32_t> chain_summary(ndc * kDcChainChunks, 0xFFFFFFFFu);
    D.chain_summary = chain_summary.data();
    const size_t xdc = (xsize + 2047) / 2048;
    for (size_t sl = 0; sl < nslabs; sl++) {
      D.dcg_first = (int)(sl * xdc);
      hipsim::launch(dc_elementwise_kernel, dim3((unsigned)(xdc * kDcParts)), dim3(256), D);
      hipsim::launch(dc_chain_summary_kernel, dim3((unsigned)(xdc * kDcChainChunks)), dim3(kDcChainThreads), D);
      // (a small grid here, so that a workgroup has several chunks of several DC groups in turn)
      hipsim::launch(dc_chain_kernel, dim3((unsigned)std::min<size_t>(xdc * kDcChainChunks, 24)), dim3(kDcChainThreads), D,
                     (int)(xdc * kDcChainChunks));
    }
  }
  free(A.dc_nac);
  for (int c = 0; c < 3; c++) free(A.nzgrid[c]);
  free(A.blk_nz);
  free(A.blk_nscan);
  free(A.blk_nzmask);
  free(A.coef_scan);
  free(A.group_ntok);
  delete tab;
  return 0;
}

__attribute__((visibility("default"))) void sim_free(sim_result* r) {
  for (int c = 0; c < 3; c++) {
    free(r->quant_dc[c]);
    free(r->xyb[c]);
  }
  free(r->raw_quant);
  free(r->strategy);
  free(r->ytox);
  free(r->ytob);
  free(r->tokens);
  free(r->group_tok_offset);
  free(r->qf);
  free(r->mask);
  free(r->ent8);
  free(r->histogram);
  free(r->dc_records);
  free(r->dc_rec_offset);
  free(r->dc_count);
  memset(r, 0, sizeof(*r));
}

// The copy-free, tile-granular packing: count / scan / measure / offsets / scan, then
// pack_tile_write_kernel in `nlaunch` tile ranges, storing every tile at its final bit position
// behind `out_bytes + 4 * misalign_words` (the blob base must be dword aligned).  out_bytes
// arrives poisoned by the caller, who checks that nothing outside the sections was touched.
__attribute__((visibility("default"))) int sim_pack_direct(const uint8_t* records, const uint64_t* sec_rec_offset,
                                                            int nsec, const uint32_t* code_table, uint8_t* out_bytes,
                                                            int misalign_words, int nlaunch, uint64_t* out_offset,
                                                            uint32_t* out_bits) {
  const uint64_t total = sec_rec_offset[nsec];
  const size_t max_tiles = total / kPackTile + nsec + 1;
  std::vector<uint32_t> sec_bytes(nsec), sec_tiles(nsec), tile_bits(max_tiles, 0xDEADu);
  std::vector<PackTileInfo> tile_info(max_tiles);
  std::vector<uint64_t> tile_base(nsec + 1);
  PackTileArgs P = {};
  P.records = records;
  P.sec_rec_offset = sec_rec_offset;
  P.nsec = nsec;
  P.code_table = code_table;
  P.sec_tiles = sec_tiles.data();
  P.tile_base = tile_base.data();
  P.tile_bits = tile_bits.data();
  P.tile_info = tile_info.data();
  P.sec_bits = out_bits;
  P.sec_bytes = sec_bytes.data();
  P.sec_byte_offset = out_offset;
  P.out = out_bytes + 4 * misalign_words;
  P.tile_end = 0xFFFFFFFFu;
  const unsigned sec_blocks = (unsigned)((nsec + 255) / 256);
  hipsim::launch(pack_tile_count_kernel, dim3(sec_blocks), dim3(256), P);
  hipsim::launch(group_scan_kernel, dim3(1), dim3(kScanThreads), (const uint32_t*)sec_tiles.data(), tile_base.data(), nsec);
  hipsim::launch(pack_tile_plan_kernel, dim3(sec_blocks), dim3(256), P);
  hipsim::launch(pack_tile_measure_kernel, dim3((unsigned)((max_tiles + kPackMeasureTilesPerGroup - 1) / kPackMeasureTilesPerGroup)),
                 dim3(kPackThreads), P);
  hipsim::launch(pack_tile_offsets_kernel,
                 dim3((unsigned)((nsec + kPackOffsetsSectionsPerGroup - 1) / kPackOffsetsSectionsPerGroup)),
                 dim3(64 * kPackOffsetsSectionsPerGroup), P);
  hipsim::launch(group_scan_kernel, dim3(1), dim3(kScanThreads), (const uint32_t*)sec_bytes.data(), out_offset, nsec);
  hipsim::launch(pack_tile_finalize_kernel, dim3((unsigned)((max_tiles + 255) / 256)), dim3(256), P);
  const uint64_t ntiles = tile_base[nsec];
  for (int c = 0; c < nlaunch; c++) {
    const uint64_t t0 = ntiles * c / nlaunch, t1 = ntiles * (c + 1) / nlaunch;
    if (t1 == t0) continue;
    P.tile_first = (uint32_t)t0;
    P.tile_end = (uint32_t)t1;
    hipsim::launch(pack_tile_write_kernel, dim3((unsigned)((t1 - t0 + kPackWriteTilesPerGroup - 1) / kPackWriteTilesPerGroup)),
                   dim3(kPackThreads), P);
  }
  return 0;
}

// The product's hand-over (EnqueueMeasure / EnqueueDeliver of jxlt_capi_pack.hip): the writing pass in `nlaunch` GROWING
// shares of an upper bound of the tile count, pack_tile_finalize_kernel filing the sections every launch completes,
// and behind every launch the copy of exactly those sections from the blob to `dst` (+ dst_shift bytes, any
// alignment) -- a memcpy here, a copy command in the product (the kernel that did it in round 4 is gone): mode 0 =
// launch mode, start-aligned at dst + dst_shift; 1 = launch mode, END-aligned at dst + dst_shift; 2 = run mode, runs
// given as (first, count, dst_offset) triples of 64-bit words, copied behind the last launch.
__attribute__((visibility("default"))) int sim_pack_deliver(const uint8_t* records, const uint64_t* sec_rec_offset,
                                                             int nsec, const uint32_t* code_table, int nlaunch, int mode,
                                                             uint8_t* dst, uint64_t dst_shift, const uint64_t* runs,
                                                             int nruns, int grid, uint64_t* out_offset,
                                                             uint32_t* out_bits, uint32_t* out_flag) {
  const uint64_t total = sec_rec_offset[nsec];
  const size_t max_tiles = total / kPackTile + nsec + 1 + 37;  // (an upper bound, like the product's)
  std::vector<uint32_t> sec_bytes(nsec), sec_tiles(nsec), tile_bits(max_tiles, 0xDEADu), launch_sec_end(kPackMaxLaunches, 0xDEADu);
  std::vector<PackTileInfo> tile_info(max_tiles);
  std::vector<uint64_t> tile_base(nsec + 1);
  std::vector<uint8_t> blob(4 * total + 8 * nsec + 128, 0xEE);
  PackTileArgs P = {};
  P.records = records;
  P.sec_rec_offset = sec_rec_offset;
  P.nsec = nsec;
  P.code_table = code_table;
  P.sec_tiles = sec_tiles.data();
  P.tile_base = tile_base.data();
  P.tile_bits = tile_bits.data();
  P.tile_info = tile_info.data();
  P.sec_bits = out_bits;
  P.sec_bytes = sec_bytes.data();
  P.sec_byte_offset = out_offset;
  P.out = blob.data();
  P.tile_end = 0xFFFFFFFFu;
  P.launches = (uint32_t)nlaunch;
  for (int i = 0; i <= nlaunch; i++)
    P.launch_t0[i] = i == nlaunch ? (uint32_t)max_tiles
                                  : (uint32_t)((double)max_tiles * ((double)((1 << i) - 1) / (double)((1 << nlaunch) - 1)));
  P.launch_sec_end = launch_sec_end.data();
  const unsigned sec_blocks = (unsigned)((nsec + 255) / 256);
  hipsim::launch(pack_tile_count_kernel, dim3(sec_blocks), dim3(256), P);
  hipsim::launch(group_scan_kernel, dim3(1), dim3(kScanThreads), (const uint32_t*)sec_tiles.data(), tile_base.data(), nsec);
  hipsim::launch(pack_tile_plan_kernel, dim3(sec_blocks), dim3(256), P);
  hipsim::launch(pack_tile_measure_kernel, dim3((unsigned)((max_tiles + kPackMeasureTilesPerGroup - 1) / kPackMeasureTilesPerGroup)),
                 dim3(kPackThreads), P);
  hipsim::launch(pack_tile_offsets_kernel,
                 dim3((unsigned)((nsec + kPackOffsetsSectionsPerGroup - 1) / kPackOffsetsSectionsPerGroup)),
                 dim3(64 * kPackOffsetsSectionsPerGroup), P);
  hipsim::launch(group_scan_kernel, dim3(1), dim3(kScanThreads), (const uint32_t*)sec_bytes.data(), out_offset, nsec);
  hipsim::launch(pack_tile_finalize_kernel, dim3((unsigned)((max_tiles + 255) / 256)), dim3(256), P);
  (void)grid;
  uint8_t* const out = dst + dst_shift;
  uint32_t s_lo = 0;
  for (int c = 0; c < nlaunch; c++) {
    PackTileArgs W = P;
    W.tile_first = P.launch_t0[c];
    W.tile_end = P.launch_t0[c + 1];
    if (W.tile_end > W.tile_first)
      hipsim::launch(pack_tile_write_kernel,
                     dim3((unsigned)((W.tile_end - W.tile_first + kPackWriteTilesPerGroup - 1) / kPackWriteTilesPerGroup)),
                     dim3(kPackThreads), W);
    if (mode != 2) {
      // as EnqueueDeliver: the sections this launch has completed, at their offset (END-aligned: against the total)
      const uint32_t s_hi = std::min<uint32_t>((uint32_t)nsec, std::max(s_lo, launch_sec_end[c]));
      const int64_t shift = mode == 1 ? -(int64_t)out_offset[nsec] : 0;
      if (out_offset[s_hi] > out_offset[s_lo])
        memcpy(out + shift + (int64_t)out_offset[s_lo], blob.data() + out_offset[s_lo], out_offset[s_hi] - out_offset[s_lo]);
      s_lo = s_hi;
    }
  }
  if (mode == 2) {
    for (int r = 0; r < nruns; r++) {
      const uint64_t lo = out_offset[runs[3 * r]], hi = out_offset[runs[3 * r] + runs[3 * r + 1]];
      if (hi > lo) memcpy(out + runs[3 * r + 2], blob.data() + lo, hi - lo);
    }
  } else if (s_lo != (uint32_t)nsec) {
    return 1;  // (the last launch must complete every section)
  }
  if (out_flag) *out_flag = 77;
  return 0;
}

// The look-back of the single pass on its own: one wave asks for the start of `tile` given the states of the tiles
// in front of it in its block and of the blocks in front (which the caller makes up).
namespace jxlt_dev {
__global__ void pack_lookback_probe_kernel(const unsigned long long* tile_state, const unsigned long long* block_state, uint32_t tile,
                                           int first, unsigned long long* out) {
  const int lane = (int)threadIdx.x;
  uint32_t windows = 0, reloads = 0;
  const PackWindowAhead mates = pack_block_mates(tile_state, tile, lane, pack_mates_load(tile_state, tile, lane), &reloads);
  const uint32_t block = tile / kPackBlockTiles;
  const unsigned long long block_start =
      pack_block_start(block_state, block, lane, pack_blocks_load(block_state, (long long)block - 1, lane), &windows, &reloads);
  unsigned long long start = pack_ahead_apply(mates, block_start);
  if (first) start = (start + 7) & ~7ull;
  out[lane] = start;
  // what the block does to a position, as its last tile would say it (the tile's own size is made up: 5 bits)
  out[64 + lane] = pack_block_state_of(pack_window_concat(mates, pack_ahead_of_tile(5u, first != 0)));
}
}  // namespace jxlt_dev
__attribute__((visibility("default"))) int sim_pack_lookback(const unsigned long long* tile_state, const unsigned long long* block_state,
                                                              uint32_t tile, int first, unsigned long long* out128) {
  hipsim::launch(jxlt_dev::pack_lookback_probe_kernel, dim3(1), dim3(64), tile_state, block_state, tile, first, out128);
  return 0;
}

// The single pass (pack_tile_stream_kernel, EnqueueStream of jxlt_capi_pack.hip): plan (which also clears the tiles' states
// and the sections' bit counts), then the writing launches over growing shares of an upper bound of the tile count
// -- no measuring pass, every tile takes its position from the tiles in front of it.  `blob` must arrive zeroed
// (the product clears it with a memset in front of the first launch); out_bits: the sections' bit counts;
// out_launch_end: which sections every launch has completed.
__attribute__((visibility("default"))) int sim_pack_stream(const uint8_t* records, const uint64_t* sec_rec_offset, int nsec,
                                                            const uint32_t* code_table, int nlaunch, uint8_t* blob,
                                                            uint32_t* out_bits, uint32_t* out_launch_end) {
  const uint64_t total = sec_rec_offset[nsec];
  const size_t max_tiles = total / kPackTile + nsec + 1 + 37;
  std::vector<uint32_t> sec_bytes(nsec), sec_tiles(nsec);
  std::vector<PackTileInfo> tile_info(max_tiles);
  std::vector<uint64_t> tile_base(nsec + 1), sec_off(nsec + 1, 0);
  std::vector<unsigned long long> tile_state(max_tiles, 0xDEADDEADDEADDEADull);
  std::vector<unsigned long long> block_state(max_tiles / kPackBlockTiles + 2, 0xDEADDEADDEADDEADull);
  for (int i = 0; i < nsec; i++) out_bits[i] = 0xDEADu;
  PackTileArgs P = {};
  P.records = records;
  P.sec_rec_offset = sec_rec_offset;
  P.nsec = nsec;
  P.code_table = code_table;
  P.sec_tiles = sec_tiles.data();
  P.tile_base = tile_base.data();
  P.tile_info = tile_info.data();
  P.sec_bits = out_bits;
  P.sec_bytes = sec_bytes.data();
  P.sec_byte_offset = sec_off.data();
  P.out = blob;
  P.tile_end = 0xFFFFFFFFu;
  P.launches = (uint32_t)nlaunch;
  for (int i = 0; i <= nlaunch; i++)
    P.launch_t0[i] = i == nlaunch ? (uint32_t)max_tiles
                                  : (uint32_t)((double)max_tiles * ((double)((1 << i) - 1) / (double)((1 << nlaunch) - 1)));
  P.launch_sec_end = out_launch_end;
  P.tile_state = tile_state.data();
  P.block_state = block_state.data();
  std::vector<uint32_t> tickets(kPackMaxLaunches, 0xDEADu);
  P.tile_ticket = tickets.data();
  const unsigned sec_blocks = (unsigned)((nsec + 255) / 256);
  if (nsec <= kPackPlanSmallSections) {  // (as EnqueuePlan: the plan of up to 1024 sections is one launch)
    hipsim::launch(pack_tile_plan_small_kernel, dim3(1), dim3(kPackPlanSmallSections), P, tile_base.data());
  } else {
    hipsim::launch(pack_tile_count_kernel, dim3(sec_blocks), dim3(256), P);
    hipsim::launch(group_scan_kernel, dim3(1), dim3(kScanThreads), (const uint32_t*)sec_tiles.data(), tile_base.data(), nsec);
    hipsim::launch(pack_tile_plan_kernel, dim3(sec_blocks), dim3(256), P);
  }
  for (int c = 0; c < nlaunch; c++) {
    PackTileArgs W = P;
    W.tile_first = P.launch_t0[c];
    W.tile_end = P.launch_t0[c + 1];
    W.launch_index = (uint32_t)c;
    if (W.tile_end > W.tile_first)
      hipsim::launch(pack_tile_stream_kernel,
                     dim3((unsigned)((W.tile_end - W.tile_first + kPackStreamTilesPerGroup - 1) / kPackStreamTilesPerGroup)),
                     dim3(kPackThreads), W);
  }
  return 0;
}

// publish_kernel: segments of dwords + a 64-bit word + the sequence word
__attribute__((visibility("default"))) void sim_publish(const uint32_t* a, uint32_t* da, uint32_t na, const uint32_t* b,
                                                         uint32_t* db, uint32_t nb, const unsigned long long* s64,
                                                         unsigned long long* d64, uint32_t* flag, uint32_t seq) {
  PublishArgs P = {};
  P.src[0] = a; P.dst[0] = da; P.words[0] = na;
  P.src[1] = b; P.dst[1] = db; P.words[1] = nb;
  P.src64 = s64;
  P.dst64 = d64;
  P.flag = flag;
  P.seq = seq;
  hipsim::launch(publish_kernel, dim3(1), dim3(kPublishThreads), P);
}

}  // extern "C"

// group_scan_kernel alone: offsets[0..n] = exclusive scan of counts[0..n)
extern "C" __attribute__((visibility("default"))) void sim_group_scan(const uint32_t* counts, int n, uint64_t* offsets) {
  hipsim::launch(group_scan_kernel, dim3(1), dim3(kScanThreads), counts, offsets, n);
}
