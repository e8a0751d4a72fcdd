// jxlt_capi_pack.hip -- libjxltiny_hip.so (include/jxl_tiny_amd.h): the section packing stage (enc_frame.cc:784-800:
// the WriteToken loops over all sections, on the device) and the hand-over of the packed sections.
#include "jxlt_context.h"
#include "jxlt_pack_kernels.h"

using namespace jxlt_dev;
using namespace jxlt_host;

namespace jxlt_host {
// One pass or two.  ONE (pack_tile_stream_kernel: every tile takes its bit position from the tiles in front of it
// while it runs; no measuring pass, none of its offsets / scan / finalize kernels, no second read of the records) up
// to 1024 groups and ~20 000 tiles, TWO (measure, lay out, write: rounds 1-3) above.  The single pass saves the chain of small
// kernels -- 2048^2: 0.41 -> 0.36 ms, 4096^2: 0.62 -> 0.58, 8192^2: 1.56 -> 1.54 -- but its kernel packs 55-85 tiles
// per us where the two-pass form's writing pass does 120 (one tile per workgroup, a ticket and the waits for the
// neighbours' sizes in front of every tile), and from ~20 000 tiles on that costs more than the measuring pass did
// (the AC sections of the 16384^2 bench frame, 25 700 tiles: 5.21-5.26 against 5.14-5.22 ms; 8192^2 of uniform noise,
// 34 700 tiles: 3.24 against 2.88; 8192^2 at d = 0.5, 15 400 tiles: 2.02 against 2.11; 4096^2 at d = 0.1, 9 800 tiles:
// 1.29 against 1.76 -- tools/ab_stream.sh, tools/token_heavy_ab.sh).  What counts is the number of tiles: the AC
// sections' record count is known when their packing is asked for.  JXLT_PACK_TWO_PASS=1 / 0 forces either.
int PackPassesForced() {  // 1 / 2, or 0: by size
  static const int forced = [] {
    const char* two = getenv("JXLT_PACK_TWO_PASS");
    return two ? (atoi(two) != 0 ? 2 : 1) : 0;
  }();
  return forced;
}
// (may a single pass be asked for at all: the plans then prepare the tiles' states)
bool PackSinglePass(const jxlt_context*) { return PackPassesForced() != 2; }
bool PackSinglePassFor(const jxlt_context* ctx, int kind, uint64_t records) {
  if (PackPassesForced() != 0) return PackPassesForced() == 1;
  // The DC-group sections (few, long: 64 at 16384^2): always one pass.  On the large frame their two passes -- nine
  // launches, 0.095 ms -- were still on the device when the AC sections' measuring pass was queued (the AC code is
  // ready 0.09 ms behind token_kernel) and held it back by 0.04 ms; one pass is three launches and 0.05 ms: 16384^2
  // 4.677 / 4.644 / 4.654 against 4.703 / 4.693 / 4.701 ms (round 6, same box, alternating; round 4 had measured the
  // opposite by 0.02 ms -- with a memset and a separate table fetch in front of the single pass, and events between its
  // kernels).
  if (kind == 0) return true;
  // The AC sections: a section has at least one tile -- a frame of 4096 groups is 4096 workgroups with a ticket, a code
  // table and a look-back each even when they hold a handful of records (16384^2 at d = 4, 1 800 tiles' worth of
  // records: 4.55-4.58 ms in one pass, 4.52-4.53 in two) --, and above 80 M records the writing pass's three launches
  // with their copies in between win.
  if ((size_t)ctx->geom.xsize_groups * ctx->geom.ysize_groups > 1024) return false;
  return records <= (80ull << 20);
}

// JXLT_PACK_LAUNCHES=<n>: the number of writing launches of the AC sections (default: 3 in two passes, 1 in one).
int PackLaunchesKnob(int dflt) {
  static const int forced = [] {
    const char* e = getenv("JXLT_PACK_LAUNCHES");
    return e ? std::max(1, std::min(atoi(e), (int)jxlt_context::PackSet::kMaxLaunches)) : 0;
  }();
  return forced ? forced : dflt;
}

// Common argument block of the tile-granular packing kernels for sections of `kind`.
PackTileArgs TileArgsOf(jxlt_context* ctx, int kind, size_t nsec) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  PackTileArgs P;
  memset(&P, 0, sizeof(P));
  P.records = kind == 1 ? ctx->tokens.p : ctx->dc_records.p;
  P.sec_rec_offset = kind == 1 ? ctx->group_off.p : ctx->dc_rec_off.p;
  P.sec_rec_count = kind == 1 ? nullptr : ctx->dc_count.p;
  P.nsec = (int)nsec;
  P.code_table = ps.code_table.p;
  P.sec_tiles = ps.sec_tiles.p;
  P.tile_base = ps.tile_base.p;
  P.tile_bits = ps.tile_bits.p;
  P.tile_info = ps.tile_info.p;
  P.sec_bits = ps.sec_bits(nsec);
  P.sec_bytes = ps.sec_bytes.p;
  P.sec_byte_offset = ps.sec_byte_off.p;
  P.out = ps.packed.p;
  P.tile_first = 0;
  P.tile_end = 0xFFFFFFFFu;
  P.launches = (uint32_t)ps.launches;
  for (int i = 0; i <= ps.launches && i <= kPackMaxLaunches; i++) P.launch_t0[i] = ps.launch_t0[i];
  P.launch_sec_end = ps.launch_sec_end.p;
  P.tile_ticket = ps.launch_sec_end.p ? ps.launch_sec_end.p + kPackMaxLaunches : nullptr;
  P.tile_state = PackSinglePass(ctx) ? ps.tile_state.p : nullptr;
  P.block_state = PackSinglePass(ctx) && ps.tile_state.p ? ps.tile_state.p + ps.state_tiles : nullptr;
  // (JXLT_TRACE_EVENTS=2; the counting slows the pass down)
  P.lookback_stats = TraceLevel() >= 2 && ctx->deliver_counter.p ? ctx->deliver_counter.p + 16 + kind * 4 : nullptr;
  return P;
}

size_t NumSections(const jxlt_context* ctx, int kind) {
  return kind == 1 ? (size_t)ctx->geom.xsize_groups * ctx->geom.ysize_groups
                   : ((ctx->xsize + 2047) / 2048) * ((ctx->ysize + 2047) / 2048);
}

// The tile plan of the sections of `kind` (asynchronous, on `stream`): tiles per section, their scan, the record
// range of every tile.  It needs the sections' record counts only, not a code: for the AC sections it is queued
// right behind the tokenisation (EnqueuePipeline), i.e. it runs while the host builds the AC code.
// rec_bound: an upper bound of the record count (sizes the per-tile arrays).
// (buffers of the plan of `kind`; *nsec_out: its number of sections)
static int PlanEnsure(jxlt_context* ctx, int kind, uint64_t rec_bound, size_t* nsec_out) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  const size_t nsec = NumSections(ctx, kind);
  const size_t max_tiles = (size_t)(rec_bound / kPackTile) + nsec + 1;
  int rc;
#define ENSURE(buf, n) if ((rc = EnsureDevice(ctx, &ps.buf, (n))) != JXLT_OK) return rc
  ENSURE(code_table, 64 * 64);
  ENSURE(sec_bytes, nsec);
  ENSURE(sec_byte_off, jxlt_context::PackSet::SizesWords(nsec));
  ENSURE(sec_tiles, nsec);
  ENSURE(tile_base, nsec + 1);
  ENSURE(tile_bits, max_tiles);
  ENSURE(tile_info, max_tiles);
  ENSURE(launch_sec_end, 2 * kPackMaxLaunches);  // (+ the single pass's tickets)
  if (PackSinglePass(ctx)) {  // (tile states, block states behind them)
    ENSURE(tile_state, max_tiles + max_tiles / kPackBlockTiles + 2);
    ps.state_tiles = max_tiles;
  }
#undef ENSURE
  *nsec_out = nsec;
  return JXLT_OK;
}
int EnqueuePlan(jxlt_context* ctx, int kind, uint64_t rec_bound, hipStream_t stream) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  size_t nsec = 0;
  const int rce = PlanEnsure(ctx, kind, rec_bound, &nsec);
  if (rce != JXLT_OK) return rce;
  const PackTileArgs P = TileArgsOf(ctx, kind, nsec);
  const unsigned sec_blocks = (unsigned)((nsec + 255) / 256);
  if (nsec <= (size_t)kPackPlanSmallSections) {  // (one launch instead of three: count, scan, plan by one workgroup)
    hipLaunchKernelGGL(pack_tile_plan_small_kernel, dim3(1), dim3(kPackPlanSmallSections), 0, stream, P, ps.tile_base.p);
  } else {
    hipLaunchKernelGGL(pack_tile_count_kernel, dim3(sec_blocks), dim3(256), 0, stream, P);
    hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, (const uint32_t*)ps.sec_tiles.p,
                       ps.tile_base.p, (int)nsec);
    hipLaunchKernelGGL(pack_tile_plan_kernel, dim3(sec_blocks), dim3(256), 0, stream, P);
  }
  HIP_TRY(ctx, hipGetLastError());
  // (which sections a launch of the writing pass completes follows from the plan and is worked out on the device --
  // pack_tile_finalize_kernel --: the host does not fetch the plan any more)
  ps.planned = true;
  ps.plan_elsewhere = stream != ps.stream;
  if (ps.plan_elsewhere) HIP_TRY(ctx, hipEventRecord(ps.plan_done, stream));
  return JXLT_OK;
}
// The plans of both kinds in ONE launch (frames of up to 1024 sections of either kind; a lane of a batch: round 6).
int EnqueuePlanBoth(jxlt_context* ctx, uint64_t dc_rec_bound, uint64_t ac_rec_bound, hipStream_t stream) {
  size_t nsec[2] = {0, 0};
  int rc;
  if ((rc = PlanEnsure(ctx, 0, dc_rec_bound, &nsec[0])) != JXLT_OK || (rc = PlanEnsure(ctx, 1, ac_rec_bound, &nsec[1])) != JXLT_OK) return rc;
  if (nsec[0] > (size_t)kPackPlanSmallSections || nsec[1] > (size_t)kPackPlanSmallSections) return JXLT_ERR_INVALID_ARGUMENT;
  const PackTileArgs P0 = TileArgsOf(ctx, 0, nsec[0]), P1 = TileArgsOf(ctx, 1, nsec[1]);
  hipLaunchKernelGGL(pack_tile_plan_small2_kernel, dim3(2), dim3(kPackPlanSmallSections), 0, stream, P0, ctx->pack[0].tile_base.p, P1,
                     ctx->pack[1].tile_base.p);
  HIP_TRY(ctx, hipGetLastError());
  for (int kind = 0; kind < 2; kind++) {
    jxlt_context::PackSet& ps = ctx->pack[kind];
    ps.planned = true;
    ps.plan_elsewhere = stream != ps.stream;
    if (ps.plan_elsewhere) HIP_TRY(ctx, hipEventRecord(ps.plan_done, stream));
  }
  return JXLT_OK;
}

// Measuring pass for the sections of `kind` (asynchronous): exact bit / byte size of every section, byte offsets,
// tile bookkeeping; the sizes are published to the host's page-locked mirror by a kernel (its sequence word:
// HostMail::sizes_seq).  The writing launches follow at once (they need nothing from the host).
int EnqueueWrites(jxlt_context* ctx, int kind);  // (below)
int EnqueueMeasure(jxlt_context* ctx, int kind, const uint32_t* code_table) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  const size_t nsec = NumSections(ctx, kind);
  // upper bound of the record count (the exact per-section counts live on the device)
  const uint64_t rec_bound = kind == 1 ? ctx->h_group_off.p[nsec] : ctx->dc_records.cap / 3;
  const size_t max_tiles = (size_t)(rec_bound / kPackTile) + nsec + 1;
  int rc;
  if (!ps.planned && (rc = EnqueuePlan(ctx, kind, rec_bound, ps.stream)) != JXLT_OK) return rc;
  if (ps.plan_elsewhere) HIP_TRY(ctx, hipStreamWaitEvent(ps.stream, ps.plan_done, 0));
  if ((rc = EnsurePinned(ctx, &ps.h_sec_byte_off, jxlt_context::PackSet::SizesWords(nsec))) != JXLT_OK) return rc;
  // The caller's table is pageable as a rule: an asynchronous copy from it would make this call wait for
  // everything queued on the stream (token_kernel!).  Staged through the context's page-locked copy instead;
  // its previous use (last frame's upload) finished before that frame's sizes were returned.
  if ((rc = EnsurePinned(ctx, &ps.h_code_table, 64 * 64)) != JXLT_OK) return rc;
  memcpy(ps.h_code_table.p, code_table, 64 * 64 * sizeof(uint32_t));
  // (fetched by a kernel that reads the page-locked copy, not by a copy command: a copy command queues behind the
  // other kind's sections on the DMA engine -- the AC measuring pass started 85 us late behind the DC-group sections'
  // download, JXLT_TRACE_EVENTS)
  {
    const PublishSeg seg = {ps.h_code_table.p, ps.code_table.p, 64 * 64};
    if ((rc = EnqueuePublish(ctx, ps.stream, &seg, 1, nullptr, nullptr, nullptr, 0)) != JXLT_OK) return rc;
  }
  // Blob capacity: <= 28 bits per record.  (Allocated before the measuring pass: its last kernel zeroes the
  // dwords in which tiles and sections meet.)
  const uint64_t blob_bound = rec_bound * 4 + nsec * 8 + 64;
  if (ps.packed.cap < blob_bound && (rc = EnsureDevice(ctx, &ps.packed, blob_bound + blob_bound / 8)) != JXLT_OK)
    return rc;
  // The writing pass runs as a few launches over shares of the tile range (an upper bound: the kernels clamp to
  // the real tile count), each followed by the hand-over of the sections it has completed (EnqueueDeliver).
  // The shares GROW (1 : 2 : 4): the kernels write faster than the link carries the bytes away (16384^2: 0.21 ms
  // against 0.35 ms for the 20 MB of AC sections), so the hand-over is the critical path and what it cannot
  // overlap is the FIRST launch; every later share only has to be written before the hand-over in front of it ends.
  // (Time from the AC sizes to the last byte in host memory, tools/pack_sweep.sh: five shrinking shares 0.42 ms,
  // five equal 0.41, five growing 0.39-0.40, four 1:2:4:8 0.38, three 1:2:4 0.37, two 1:4 0.42.)
  // (JXLT_PACK_LAUNCHES=<n> overrides the number of launches: tools/pack_sweep.sh)
  const int ac_launches = PackLaunchesKnob(3);
  const double growth = 2.0;
  const int want = kind == 0 ? 1 : ac_launches;
  ps.launches = (int)std::min<size_t>((size_t)want, std::max<size_t>(1, max_tiles / 64));
  for (int i = 0; i <= ps.launches; i++) {
    const double share = growth > 1.0 ? (std::pow(growth, i) - 1.0) / (std::pow(growth, ps.launches) - 1.0)
                                      : (double)i / ps.launches;
    ps.launch_t0[i] = i == ps.launches ? (uint32_t)max_tiles : (uint32_t)((double)max_tiles * share);
  }
  const PackTileArgs P = TileArgsOf(ctx, kind, nsec);
  TraceMark(ctx, kind ? "AC measure start" : "DC measure start", ps.stream);
  hipLaunchKernelGGL(pack_tile_measure_kernel,
                     dim3((unsigned)((max_tiles + kPackMeasureTilesPerGroup - 1) / kPackMeasureTilesPerGroup)),
                     dim3(kPackThreads), 0, ps.stream, P);
  hipLaunchKernelGGL(pack_tile_offsets_kernel,
                     dim3((unsigned)((nsec + kPackOffsetsSectionsPerGroup - 1) / kPackOffsetsSectionsPerGroup)),
                     dim3(64 * kPackOffsetsSectionsPerGroup), 0, ps.stream, P);
  hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(kScanThreads), 0, ps.stream, (const uint32_t*)ps.sec_bytes.p,
                     ps.sec_byte_off.p, (int)nsec);
  // The sizes are final behind the scan (the last kernel of the pass only moves the tiles to their places): they
  // leave for the host by the auxiliary stream (idle by now), beside that kernel -- offsets and bit counts lie
  // behind each other, one publish_kernel stores them to the page-locked mirror and then the pass's number to the
  // word the host polls.  (Rounds 1-3: hipMemcpyAsync + event; the copy alone took 20 us of device time.)
  TraceMark(ctx, kind ? "AC scan done" : "DC scan done", ps.stream);
  hipLaunchKernelGGL(pack_tile_finalize_kernel, dim3((unsigned)((max_tiles + 255) / 256)), dim3(256), 0, ps.stream, P);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ps.finalized, ps.stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ps.finalized, 0));
  ps.pack_seq++;
  {
    const PublishSeg segs[2] = {{ps.sec_byte_off.p, ps.h_sec_byte_off.p, jxlt_context::PackSet::SizesWords(nsec) * 2},
                                {ps.launch_sec_end.p, ps.h_launch_sec_end, (size_t)kPackMaxLaunches}};
    if ((rc = EnqueuePublish(ctx, ctx->aux_stream, segs, 2, nullptr, nullptr, &ctx->mail.p->sizes_seq[kind][0],
                             ps.pack_seq)) != JXLT_OK)
      return rc;
  }
  // (the plan is used up: the kernel above has replaced every tile's section index by the section's bit position.  A
  // second measuring pass of the same encode plans again; until round 3 it did not, read section offsets at those
  // bit positions and wrote wherever they pointed.)
  ps.planned = false;
  ps.measured_sections = nsec;
  ps.max_tiles = max_tiles;
  ps.writes_queued = false;
  ps.streamed = false;
  return EnqueueWrites(ctx, kind);
}

// Bytes the sections of `kind` take at most with this code: the code lengths of the context's own tokens (its
// histograms are in the host's mirror by now) + the raw bits and the padding a section can add.
uint64_t SectionBytesBound(const jxlt_context* ctx, int kind, const uint32_t* table, size_t nsec) {
  const uint32_t* hist = ctx->h_hist.p + (kind == 1 ? 0 : 64 * 64);
  uint64_t bits = 0;
  for (uint32_t c = 0; c < 64; c++)
    for (uint32_t sym = 0; sym < 64; sym++) {
      const uint32_t n = hist[c * 64 + sym];
      if (n) bits += (uint64_t)n * ((table[c * 64 + sym] >> 16) + (sym >= 16 ? (sym >> 2) - 2u : 0u));
    }
  return bits / 8 + 32 * (uint64_t)nsec + 256;
}

// The single pass over the sections of `kind` (asynchronous; the default, see PackSinglePass): plan (if it is not
// there yet), code table, a zeroed blob, and the launches of pack_tile_stream_kernel over growing shares of the
// tiles.  Behind every launch a publish_kernel on the auxiliary stream carries the sections' bit counts and "which
// sections are complete" to the host's mirror and sets that launch's word (HostMail::stream_seq): the host turns
// bit counts into byte offsets itself (a prefix sum over a few thousand numbers) and issues the copy commands.
int EnqueueStream(jxlt_context* ctx, int kind, const uint32_t* code_table) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  const size_t nsec = NumSections(ctx, kind);
  const uint64_t rec_bound = kind == 1 ? ctx->h_group_off.p[nsec] : ctx->dc_records.cap / 3;
  const size_t max_tiles = (size_t)(rec_bound / kPackTile) + nsec + 1;
  int rc;
  if (!ps.planned && (rc = EnqueuePlan(ctx, kind, rec_bound, ps.stream)) != JXLT_OK) return rc;
  if (ps.plan_elsewhere) HIP_TRY(ctx, hipStreamWaitEvent(ps.stream, ps.plan_done, 0));
  if ((rc = EnsurePinned(ctx, &ps.h_sec_byte_off, jxlt_context::PackSet::SizesWords(nsec))) != JXLT_OK) return rc;
  if ((rc = EnsurePinned(ctx, &ps.h_code_table, 64 * 64)) != JXLT_OK) return rc;
  memcpy(ps.h_code_table.p, code_table, 64 * 64 * sizeof(uint32_t));
  const uint64_t blob_bound = rec_bound * 4 + nsec * 8 + 64;
  if (ps.packed.cap < blob_bound && (rc = EnsureDevice(ctx, &ps.packed, blob_bound + blob_bound / 8)) != JXLT_OK)
    return rc;
  // The code table (fetched by a kernel that reads the page-locked copy: a copy command would queue behind the other
  // kind's sections on the DMA engine) and the zeroed blob (the tiles OR their first and last dwords into it: zero up to
  // where the sections can reach with this code) in ONE launch: as a publish kernel + hipMemsetAsync they were two to
  // three (the runtime fills an odd size with two kernels), and a batch of small frames is bound by its number of
  // launches -- 48 resident 3840x2160 frames over eight lanes: 30.4 -> 33 GP/s with this and the one-launch tile plan.
  ps.zeroed_bytes = std::min<uint64_t>((SectionBytesBound(ctx, kind, code_table, nsec) + 4095) & ~uint64_t(4095), ps.packed.cap & ~uint64_t(15));
  {
    const unsigned groups = (unsigned)std::min<uint64_t>(1024, std::max<uint64_t>(1, ps.zeroed_bytes / (16 * kPackPrepareThreads * 4)));
    hipLaunchKernelGGL(pack_prepare_kernel, dim3(groups), dim3(kPackPrepareThreads), 0, ps.stream,
                       (const uint32_t*)ps.h_code_table.p, ps.code_table.p, ps.packed.p, (unsigned long long)ps.zeroed_bytes);
    HIP_TRY(ctx, hipGetLastError());
  }
  // ONE launch is the default here: the frames the single pass is used for (up to 1024 groups, 6 MB of AC sections)
  // are packed in 0.02-0.1 ms, and every further launch costs a publish kernel, a copy command and a ramp -- 2048^2:
  // 0.355 / 0.377 / 0.382 ms with one / two / three launches, 4096^2: 0.578 / 0.591 / 0.613, 8192^2: 1.562 / 1.568 /
  // 1.596, 48 resident 3840x2160 frames over six lanes: 3423 / 3370 / 3284 frames per second (tools/launches_small.sh).
  const int ac_launches = PackLaunchesKnob(1);
  const double growth = 2.0;
  const int want = kind == 0 ? 1 : ac_launches;
  ps.launches = (int)std::min<size_t>((size_t)want, std::max<size_t>(1, max_tiles / 64));
  for (int i = 0; i <= ps.launches; i++) {
    const double share = growth > 1.0 ? (std::pow(growth, i) - 1.0) / (std::pow(growth, ps.launches) - 1.0)
                                      : (double)i / ps.launches;
    ps.launch_t0[i] = i == ps.launches ? (uint32_t)max_tiles : (uint32_t)((double)max_tiles * share);
  }
  ps.pack_seq++;
  for (int i = 0; i < ps.launches; i++) {
    PackTileArgs W = TileArgsOf(ctx, kind, nsec);
    W.tile_first = ps.launch_t0[i];
    W.tile_end = ps.launch_t0[i + 1];
    W.launch_index = (uint32_t)i;
    TraceMark(ctx, kind ? "AC stream launch start" : "DC stream launch start", ps.stream);
    if (W.tile_end > W.tile_first)
      hipLaunchKernelGGL(pack_tile_stream_kernel,
                         dim3((unsigned)((W.tile_end - W.tile_first + kPackStreamTilesPerGroup - 1) / kPackStreamTilesPerGroup)),
                         dim3(kPackThreads), 0, ps.stream, W);
    HIP_TRY(ctx, hipGetLastError());
    // (the launch's event -- what its sections' copy waits for.  One frame at a time: BEHIND the publication below.  The
    // copy cannot be issued before the host has the sizes anyway, and an event record between the launch and the publish
    // kernel is a barrier packet of its own, 6 us on the way of the sizes to the host.  A lane of a batch: right here,
    // as until round 6 -- with the record behind the publication the resident 3840x2160 batch lost 8 %, 34.0 against
    // 35.9-37.2 GP/s with four lanes on the same box; every other record of R6.10 is neutral there)
    if (ctx->throughput_waits) HIP_TRY(ctx, hipEventRecord(ps.launch_done[i], ps.stream));
    TraceMark(ctx, kind ? "AC stream launch done" : "DC stream launch done", ps.stream);
    // (IN the stream: a one-workgroup kernel on another stream waits for a free slot behind the next launch's
    // workgroups -- the first launch's word arrived when the last launch had ended)
    const PublishSeg segs[2] = {{ps.sec_bits(nsec), ps.h_sec_bits(nsec), nsec},
                                {ps.launch_sec_end.p, ps.h_launch_sec_end, (size_t)kPackMaxLaunches}};
    if ((rc = EnqueuePublish(ctx, ps.stream, segs, 2, nullptr, nullptr, &ctx->mail.p->stream_seq[kind][i][0],
                             ps.pack_seq)) != JXLT_OK)
      return rc;
    if (!ctx->throughput_waits) HIP_TRY(ctx, hipEventRecord(ps.launch_done[i], ps.stream));
  }
  ps.planned = false;
  ps.measured_sections = nsec;
  ps.max_tiles = max_tiles;
  ps.writes_queued = true;
  ps.streamed = true;
  ps.offsets_done_sections = 0;
  ps.h_sec_byte_off.p[0] = 0;
  ps.launches_seen = 0;
  return JXLT_OK;
}

// Single pass: waits for launch `i` and extends the host's byte offsets over the sections that launch completed.
// Returns the number of sections whose offsets are final in *sections_done.
int StreamAdvance(jxlt_context* ctx, int kind, int upto_launch, uint32_t* sections_done) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  const size_t nsec = ps.measured_sections;
  while (ps.launches_seen <= upto_launch) {
    const int i = ps.launches_seen;
    const int rcw = WaitWord(ctx, &ctx->mail.p->stream_seq[kind][i][0], ps.pack_seq, ps.stream, "section packing");
    if (rcw != JXLT_OK) return rcw;
    // (launch_sec_end[i]: 0xFFFFFFFF = the launch had no tile of its own)
    const uint32_t filed = ps.h_launch_sec_end[i];
    uint32_t s_hi = i + 1 == ps.launches ? (uint32_t)nsec
                    : filed == 0xFFFFFFFFu ? ps.offsets_done_sections
                                           : std::min<uint32_t>(filed, (uint32_t)nsec);
    s_hi = std::max(s_hi, ps.offsets_done_sections);
    uint64_t* off = ps.h_sec_byte_off.p;
    const uint32_t* bits = ps.h_sec_bits(nsec);
    for (uint32_t s = ps.offsets_done_sections; s < s_hi; s++) off[s + 1] = off[s] + ((bits[s] + 7u) >> 3);
    ps.offsets_done_sections = s_hi;
    ps.launches_seen++;
  }
  if (ps.offsets_done_sections == nsec && ps.h_sec_byte_off.p[nsec] > ps.zeroed_bytes) {
    ctx->error = "section packing: the sections outgrew the bound computed from the histograms (internal error)";
    return JXLT_ERR_INTERNAL;
  }
  if (sections_done) *sections_done = ps.offsets_done_sections;
  return JXLT_OK;
}

// The writing pass of the sections of `kind` behind their measuring pass (asynchronous).
int EnqueueWrites(jxlt_context* ctx, int kind) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  if (ps.writes_queued) return JXLT_OK;
  ps.writes_queued = true;
  const size_t nsec = ps.measured_sections;
  for (int i = 0; i < ps.launches; i++) {
    PackTileArgs W = TileArgsOf(ctx, kind, nsec);
    W.tile_first = ps.launch_t0[i];
    W.tile_end = ps.launch_t0[i + 1];
    if (W.tile_end > W.tile_first)
      hipLaunchKernelGGL(pack_tile_write_kernel,
                         dim3((unsigned)((W.tile_end - W.tile_first + kPackWriteTilesPerGroup - 1) / kPackWriteTilesPerGroup)),
                         dim3(kPackThreads), 0, ps.stream, W);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ps.launch_done[i], ps.stream));
    TraceMark(ctx, kind ? "AC write launch done" : "DC write launch done", ps.stream);
  }
  return JXLT_OK;
}

bool SizesReady(const jxlt_context* ctx, int kind) {
  const jxlt_context::PackSet& ps = ctx->pack[kind];
  const volatile uint32_t* w = ps.streamed ? &ctx->mail.p->stream_seq[kind][ps.launches - 1][0] : &ctx->mail.p->sizes_seq[kind][0];
  return *w == ps.pack_seq;
}
int WaitSizes(jxlt_context* ctx, int kind) {
  if (ctx->pack[kind].streamed) return StreamAdvance(ctx, kind, ctx->pack[kind].launches - 1, nullptr);
  return WaitWord(ctx, &ctx->mail.p->sizes_seq[kind][0], ctx->pack[kind].pack_seq, ctx->aux_stream, "section measuring");
}

void FillMeasured(jxlt_context* ctx, int kind, jxlt_packed_sections* out) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  out->bytes = nullptr;
  out->section_offset = ps.h_sec_byte_off.p;
  out->section_bits = ps.h_sec_bits(ps.measured_sections);
  out->num_sections = ps.measured_sections;
}

// How the host learns that a hand-over is complete: from a one-workgroup kernel behind the copies that stores the
// hand-over's number to a word the host polls -- or, for the short hand-overs of a small frame (up to 256 groups; and
// for every one of a batch lane, which polls in short sleeps), by asking the copy stream itself (hipStreamQuery in
// WaitDeliveries): the kernel behind a copy costs its launch, ~6 us of a 0.5 ms frame (4096^2 0.494-0.503 -> 0.480-0.488
// ms, 2048^2 0.299-0.306 -> 0.289-0.299), while a host thread that asks the runtime in a loop during the 0.1-0.4 ms of a
// large frame's copies gets in the way of the copy commands now and then (one run in three: 8192^2 1.50 instead of
// 1.33-1.35 ms, 16384^2 5.0 instead of 4.73).  Both kinds of a frame the same way (8192^2 with only the DC-group
// sections' hand-over asked for: 1.35-1.37 against 1.35).
static bool CompletionByQuery(const jxlt_context* ctx) {
  return ctx->throughput_waits || (size_t)ctx->geom.xsize_groups * ctx->geom.ysize_groups <= 256;
}

// hipMemcpyAsync with the time the CALL took on the host: kept per encode (jxlt_encode_stats: a call that meets the
// runtime creating a copy engine's queue takes milliseconds); JXLT_TRACE_EVENTS reports calls of more
// than 0.5 ms (level 2: every call).
hipError_t TimedCopy(jxlt_context* ctx, void* dst, const void* src, size_t bytes, hipStream_t stream, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  ctx->copy_calls++;
  if (us > ctx->longest_copy_call_us) ctx->longest_copy_call_us = (float)us;
  if (TraceEventsOn()) {
    if (us > 500.0) fprintf(stderr, "jxlt slow call: hipMemcpyAsync (%s, %zu bytes) took %.3f ms on the host\n", what, bytes, us * 1e-3);
    if (TraceLevel() >= 2) fprintf(stderr, "jxlt copy call: %s %zu bytes %.1f us on the host\n", what, bytes, us);
  }
  return e;
}

// The hand-over of the measured and (being) written sections of `kind` to `dst` (asynchronous; page-locked host memory
// or device memory): behind every launch of the writing pass the whole sections it completed leave, while later
// launches are still packing.  runs == nullptr: all sections back to back, starting at dst (end_aligned: ENDING at dst).
// The bytes travel by COPY COMMANDS (hipMemcpyAsync: the DMA engines), issued by the host once it has the sizes -- not
// by a kernel that stores to the destination itself, although such a kernel needs no host round trip
// (pack_deliver_kernel, round 4's first version, removed in round 5).  A kernel that stores to host memory and an
// HBM-bound kernel beside it slow each other down badly -- tools/d2h_interfere_probe.hip: the hand-over falls from 54
// to 20-37 GB/s and the other kernel takes 20-35 % longer, whatever the grid, the alignment or the kind of store --
// while a copy command keeps 53 GB/s and costs its neighbour 2 %.  In the frame: AC measuring pass 0.18 instead of
// 0.09 ms, writing launches 1.3-1.6x, step 5.41-5.48 against 5.26-5.31 ms.
int EnqueueDeliver(jxlt_context* ctx, int kind, uint8_t* dst, const jxlt_section_run* runs, size_t nruns, int end_aligned) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  const size_t nsec = ps.measured_sections;
  // (the two kinds leave on streams of their own: two copy commands in flight hide each other's start-up, and the
  // DC-group sections -- a fifth of the bytes -- do not stand in front of the first AC sections)
  const hipStream_t out_stream = kind ? ctx->copy_stream : ctx->dc_copy_stream;
  if (ps.streamed && runs == nullptr && !end_aligned) {
    // single pass, sections back to back from dst on: behind every launch the sections it completed leave
    const uint64_t* off = ps.h_sec_byte_off.p;
    uint32_t s_lo = 0;
    for (int i = 0; i < ps.launches; i++) {
      uint32_t s_hi = 0;
      const int rca = StreamAdvance(ctx, kind, i, &s_hi);
      if (rca != JXLT_OK) return rca;
      if (off[s_hi] > off[s_lo]) {
        // (the launch's word may have been set by the launch itself, before its end: the copy waits for the end)
        HIP_TRY(ctx, hipStreamWaitEvent(out_stream, ps.launch_done[i], 0));
        TraceMark(ctx, kind ? "AC copy start" : "DC copy start", out_stream);
        HIP_TRY(ctx, TimedCopy(ctx, dst + off[s_lo], ps.packed.p + off[s_lo], off[s_hi] - off[s_lo], out_stream, kind ? "AC sections" : "DC-group sections"));
        TraceMark(ctx, kind ? "AC copy done" : "DC copy done", out_stream);
      }
      s_lo = s_hi;
    }
    if (CompletionByQuery(ctx)) {
      ctx->deliver_by_query[kind] = true;  // (completion = the copy stream having drained: WaitDeliveries)
    } else {
      const int rcp = EnqueuePublish(ctx, out_stream, nullptr, 0, nullptr, nullptr, &ctx->mail.p->delivered_seq[kind][0], ++ctx->deliver_seq[kind]);
      if (rcp != JXLT_OK) return rcp;
    }
    ctx->deliveries_pending = true;
    return JXLT_OK;
  }
  {
    const int rcs = WaitSizes(ctx, kind);
    if (rcs != JXLT_OK) return rcs;
    const uint64_t* off = ps.h_sec_byte_off.p;
    if (runs == nullptr) {
      const int64_t shift = end_aligned ? -(int64_t)off[nsec] : 0;
      uint32_t s_lo = 0;
      for (int i = 0; i < ps.launches; i++) {
        const uint32_t s_hi = ps.streamed ? (i + 1 == ps.launches ? (uint32_t)nsec : s_lo)
                                          : std::min<uint32_t>((uint32_t)nsec, std::max(s_lo, ps.h_launch_sec_end[i]));
        if (off[s_hi] > off[s_lo]) {
          HIP_TRY(ctx, hipStreamWaitEvent(out_stream, ps.launch_done[i], 0));
          TraceMark(ctx, kind ? "AC copy start" : "DC copy start", out_stream);
          HIP_TRY(ctx, TimedCopy(ctx, dst + shift + (int64_t)off[s_lo], ps.packed.p + off[s_lo], off[s_hi] - off[s_lo], out_stream,
                                 kind ? "AC sections" : "DC-group sections"));
          TraceMark(ctx, kind ? "AC copy done" : "DC copy done", out_stream);
        }
        s_lo = s_hi;
      }
    } else {
      HIP_TRY(ctx, hipStreamWaitEvent(out_stream, ps.launch_done[ps.launches - 1], 0));
      for (size_t r = 0; r < nruns; r++) {
        if ((size_t)runs[r].first_section + runs[r].num_sections > nsec) {
          ctx->error = "jxlt_pack_deliver: a run names sections the measuring pass did not see";
          return JXLT_ERR_INVALID_ARGUMENT;
        }
        const uint64_t lo = off[runs[r].first_section], hi = off[runs[r].first_section + runs[r].num_sections];
        if (hi > lo)
          HIP_TRY(ctx, TimedCopy(ctx, dst + runs[r].dst_offset, ps.packed.p + lo, hi - lo, out_stream, kind ? "AC sections (run)" : "DC-group sections (run)"));
      }
    }
    // completion: a one-workgroup kernel behind the copies stores the hand-over's number to the word the host polls
    if (CompletionByQuery(ctx)) {
      ctx->deliver_by_query[kind] = true;  // (completion = the copy stream having drained: WaitDeliveries)
    } else {
      const int rcp = EnqueuePublish(ctx, out_stream, nullptr, 0, nullptr, nullptr, &ctx->mail.p->delivered_seq[kind][0], ++ctx->deliver_seq[kind]);
      if (rcp != JXLT_OK) return rcp;
    }
    ctx->deliveries_pending = true;
    return JXLT_OK;
  }
}

// The DC-group sections' hand-over that was asked for before their sizes had arrived: issued now if the sizes are
// there (wait: whether or not -- waits for them).
int IssueDeferred(jxlt_context* ctx, bool wait) {
  jxlt_context::DeferredDeliver& d = ctx->deferred_dc;
  if (!d.pending || ctx->in_deferred) return JXLT_OK;
  if (!wait && !SizesReady(ctx, 0)) return JXLT_OK;
  ctx->in_deferred = true;
  d.pending = false;
  const int rc = EnqueueDeliver(ctx, 0, d.dst, d.runs.empty() ? nullptr : d.runs.data(), d.runs.size(), d.end_aligned);
  ctx->in_deferred = false;
  return rc;
}

}  // namespace jxlt_host

extern "C" {


// ---- the packing stage's three calls (include/jxl_tiny_amd.h) -------------------------------------------------
int jxlt_pack_begin(jxlt_context* ctx, int kind, const uint32_t* code_table) {
  if (!ctx || !code_table || (kind != 0 && kind != 1)) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded || (kind == 1 && !ctx->offsets_fetched)) {
    ctx->error = "jxlt_pack_begin needs jxlt_encode_enqueue (+ jxlt_fetch_histograms for the AC sections) first";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rc0 = ResolveRootTableOverflow(ctx);
    if (rc0 != JXLT_OK) return rc0;
  }
  if (kind == 0 && ctx->deferred_dc.pending) {
    const int rcd = IssueDeferred(ctx, /*wait=*/true);
    if (rcd != JXLT_OK) return rcd;
  }
  if (ctx->deliveries_pending && ctx->pack[kind].measured_sections != 0) {
    // (a second pass of this kind within one encode overwrites the blob the first pass's hand-over reads)
    const int rcw = WaitDeliveries(ctx);
    if (rcw != JXLT_OK) return rcw;
  }
  const uint64_t records = kind == 1 ? ctx->h_group_off.p[NumSections(ctx, 1)] : 0;
  return PackSinglePassFor(ctx, kind, records) ? EnqueueStream(ctx, kind, code_table) : EnqueueMeasure(ctx, kind, code_table);
}

int jxlt_pack_sizes(jxlt_context* ctx, int kind, jxlt_packed_sections* out) {
  if (!ctx || !out || (kind != 0 && kind != 1)) return JXLT_ERR_INVALID_ARGUMENT;
  if (ctx->pack[kind].measured_sections == 0) {
    ctx->error = "jxlt_pack_sizes needs jxlt_pack_begin first";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  const int rc = WaitSizes(ctx, kind);
  if (rc != JXLT_OK) return rc;
  if (kind == 0 && ctx->deferred_dc.pending) {
    const int rcd = IssueDeferred(ctx, /*wait=*/true);
    if (rcd != JXLT_OK) return rcd;
  }
  FillMeasured(ctx, kind, out);
  return JXLT_OK;
}

int jxlt_pack_deliver(jxlt_context* ctx, int kind, uint8_t* dst, const jxlt_section_run* runs, size_t num_runs,
                      int end_aligned) {
  if (!ctx || !dst || (kind != 0 && kind != 1) || (runs == nullptr) != (num_runs == 0) || (runs && end_aligned))
    return JXLT_ERR_INVALID_ARGUMENT;
  if (ctx->pack[kind].measured_sections == 0) {
    ctx->error = "jxlt_pack_deliver needs jxlt_pack_begin first";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // The sections travel by copy commands (hipMemcpyAsync, hipMemcpyDefault), so the destination may be anything a
  // copy command accepts and the caller's pointer is passed through as it is: device memory; page-locked host memory
  // (jxlt_output_buffer / jxlt_pinned_alloc / jxlt_pinned_register / hipHostRegister with or without the mapped flag)
  // -- the copies then run beside the caller, who reads the bytes behind jxlt_synchronize; or ordinary pageable host
  // memory, the SYNCHRONOUS fallback: the runtime stages such a copy and the call that issues it (this one, or for the
  // DC-group sections the library's next wait) returns only when the bytes have arrived.  (Until round 5 a kernel
  // stored the bytes and the destination had to be mapped into the device's address space: ADVICE r5.)
  uint8_t* const dev_dst = dst;
  for (size_t r = 0; r < num_runs; r++) {
    if ((size_t)runs[r].first_section + runs[r].num_sections > ctx->pack[kind].measured_sections) {
      ctx->error = "jxlt_pack_deliver: a run names sections the measuring pass did not see";
      return JXLT_ERR_INVALID_ARGUMENT;
    }
  }
  ctx->delivered_kinds |= 1u << kind;  // (only a request that passed validation counts as a hand-over of the kind)
  if (kind == 0) {
    if (ctx->deferred_dc.pending) {  // (a second hand-over of the kind: the first one first)
      const int rcd = IssueDeferred(ctx, /*wait=*/true);
      if (rcd != JXLT_OK) return rcd;
    }
    if (!SizesReady(ctx, 0)) {
      jxlt_context::DeferredDeliver& d = ctx->deferred_dc;
      d.pending = true;
      d.dst = dev_dst;
      d.end_aligned = end_aligned;
      d.runs.assign(runs, runs + num_runs);
      ctx->deliveries_pending = true;
      return JXLT_OK;
    }
  }
  return EnqueueDeliver(ctx, kind, dev_dst, runs, num_runs, end_aligned);
}

// ---- test-suite forms on top of the three calls (include/jxl_tiny_amd_testing.h) -------------------------------
int jxlt_pack_sections(jxlt_context* ctx, int kind, const uint32_t* code_table, jxlt_packed_sections* out) {
  if (!ctx || !code_table || !out || (kind != 0 && kind != 1)) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded || !ctx->offsets_fetched) {
    ctx->error = "jxlt_pack_sections needs jxlt_encode_enqueue + jxlt_fetch_histograms/side_info first";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  int rc = jxlt_pack_begin(ctx, kind, code_table);
  if (rc != JXLT_OK || (rc = jxlt_pack_sizes(ctx, kind, out)) != JXLT_OK) return rc;
  jxlt_context::PackSet& ps = ctx->pack[kind];
  const uint64_t total_bytes = ps.h_sec_byte_off.p[ps.measured_sections];
  if (ps.h_packed.cap < total_bytes + 16 &&
      (rc = EnsurePinned(ctx, &ps.h_packed, total_bytes + total_bytes / 4 + 4096)) != JXLT_OK)
    return rc;
  if ((rc = jxlt_pack_deliver(ctx, kind, ps.h_packed.p, nullptr, 0, 0)) != JXLT_OK) return rc;
  if ((rc = jxlt_synchronize(ctx)) != JXLT_OK) return rc;
  out->bytes = ps.h_packed.p;
  return JXLT_OK;
}


}  // extern "C"
