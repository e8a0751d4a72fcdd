// jxlt_capi_encode.hip -- libjxltiny_hip.so (include/jxl_tiny_amd.h): the device pipeline of a frame and what the
// host gets back from it.
//
// A whole frame is processed by these launches on the context's streams:
//   tile12_kernel (one workgroup per 64x64 tile)   -> side-band grids, scan-ordered coefficients
//   dc_* kernels  (DC-group tokenisation)
//   token_kernel  (one workgroup per 256x256 group) -> raw 3-byte token records
// and their small results (histograms, counts) reach the host through words that publish_kernel stores to
// page-locked memory and the host polls.
#include "jxlt_context.h"
#include "jxlt_tile_kernel.h"
#include "jxlt_token_kernel.h"
#include "jxlt_dc_kernels.h"
#include "jxlt_publish_kernel.h"
#include "jxlt_host_tables.h"

using namespace jxlt_dev;
using namespace jxlt_host;

namespace jxlt_host {
const char kUnsupportedValues[] =
    "the frame has values the codestream cannot carry (a quantised coefficient beyond 16 bits or a DC value beyond "
    "int16: samples around 1e38 or infinities)";

// (diagnostics, JXLT_TRACE_EVENTS=1: a timed event on `stream`, listed against the encode's first event by jxlt_synchronize)
int TraceLevel() {  // JXLT_TRACE_EVENTS: 1 = device-side event times, 2 = + every copy call and the look-back statistics
  static const int level = [] {
    const char* e = getenv("JXLT_TRACE_EVENTS");
    return e ? atoi(e) : 0;
  }();
  return level;
}
bool TraceEventsOn() { return TraceLevel() != 0; }
void TraceMark(jxlt_context* ctx, const char* name, hipStream_t stream) {
  if (!TraceEventsOn()) return;
  if (ctx->trace_used == ctx->trace.size()) {
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return;
    ctx->trace.push_back({name, ev});
  }
  ctx->trace[ctx->trace_used].name = name;
  (void)hipEventRecord(ctx->trace[ctx->trace_used].ev, stream);
  ctx->trace_used++;
}
void TraceDump(jxlt_context* ctx) {
  if (!TraceEventsOn() || ctx->trace_used == 0) return;
  (void)hipDeviceSynchronize();
  if (ctx->deliver_counter.p) {
    uint32_t st[8];
    if (hipMemcpy(st, ctx->deliver_counter.p + 16, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess) {
      for (int k = 0; k < 2; k++)
        if (st[k * 4])
          fprintf(stderr, "jxlt look-back (%s): %u tiles, %.2f windows per tile (most %u), %.2f reloads per tile\n", k ? "AC" : "DC",
                  st[k * 4], (double)st[k * 4 + 1] / st[k * 4], st[k * 4 + 3], (double)st[k * 4 + 2] / st[k * 4]);
      (void)hipMemset(ctx->deliver_counter.p + 16, 0, sizeof(st));
    }
  }
  for (size_t i = 0; i < ctx->trace_used; i++) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->trace[i].ev) == hipSuccess)
      fprintf(stderr, "jxlt event: %9.3f ms  %s\n", ms, ctx->trace[i].name);
  }
  ctx->trace_used = 0;
}

// Waits until a kernel has stored `want` to a sequence word in page-locked memory (HostMail).  Spins: the waits
// inside a frame are fractions of a millisecond, and the word is seen ~6 us earlier than an event would be
// (tools/d2h_probe.hip).  A device fault would leave the word unwritten for ever: the stream is asked for errors
// every couple of milliseconds, and a wait gives up after two minutes.
int WaitWord(jxlt_context* ctx, const uint32_t* word, uint32_t want, hipStream_t stream, const char* what) {
  const volatile uint32_t* w = word;
  if (*w == want) return JXLT_OK;
  const auto t0 = std::chrono::steady_clock::now();
  // the site's memory (see jxlt_context::WaitSite)
  if (ctx->wait_geometry[0] != ctx->xsize || ctx->wait_geometry[1] != ctx->ysize) {
    for (auto& site : ctx->wait_sites) site = jxlt_context::WaitSite();
    ctx->wait_geometry[0] = ctx->xsize;
    ctx->wait_geometry[1] = ctx->ysize;
  }
  jxlt_context::WaitSite* site = nullptr;
  for (auto& s : ctx->wait_sites) {
    if (s.what == what || s.what == nullptr) {
      s.what = what;
      site = &s;
      break;
    }
  }
  const auto done = [&]() {
    if (site) {
      site->last_us[1] = site->last_us[0];
      site->last_us[0] = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    return JXLT_OK;
  };
  const bool deferred_work = ctx->deferred_dc.pending && !ctx->in_deferred;  // (this wait has something to issue: no sleep)
  // (measured: 4.80-4.83 ms per 16384^2 step with the sleep, 4.80-4.86 without -- the same, minus a spinning core)
  if (!ctx->throughput_waits && site && !deferred_work) {
    const float expect = std::min(site->last_us[0], site->last_us[1]);
    if (expect > 400.0f) {
      std::this_thread::sleep_for(std::chrono::microseconds((long)(expect * 0.7f)));
      if (*w == want) return done();
    }
  }
  auto next_check = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
  int rounds = 0;
  for (;;) {
    for (int spin = 0; spin < 256; spin++) {
      if (*w == want) return done();
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (ctx->deferred_dc.pending && !ctx->in_deferred) {
      const int rcd = IssueDeferred(ctx, /*wait=*/false);
      if (rcd != JXLT_OK) return rcd;
    }
    // (a batch lane: ~5 us of spinning, then short sleeps -- the thread's timer slack is a microsecond, frame_batch.cc)
    if (ctx->throughput_waits && ++rounds >= 2) {
      struct timespec ts = {0, 8000};
      nanosleep(&ts, nullptr);
    }
    const auto now = std::chrono::steady_clock::now();
    if (now < next_check) continue;
    next_check = now + std::chrono::milliseconds(2);
    const hipError_t e = hipStreamQuery(stream);
    if (e != hipSuccess && e != hipErrorNotReady) {
      ctx->error = std::string(what) + ": " + hipGetErrorString(e);
      return JXLT_ERR_NO_DEVICE;
    }
    if (e == hipSuccess && *w != want) {
      // the stream has drained: give the word's store a moment to arrive, then it never will
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
      if (*w == want) return done();
      if (hipStreamQuery(stream) == hipSuccess && *w != want &&
          now - t0 > std::chrono::milliseconds(200)) {
        ctx->error = std::string(what) + ": the device finished without reporting";
        return JXLT_ERR_INTERNAL;
      }
    }
    if (now - t0 > std::chrono::seconds(120)) {
      ctx->error = std::string(what) + ": timed out";
      return JXLT_ERR_INTERNAL;
    }
  }
}

// Every hand-over queued so far (both kinds) has finished.
int WaitDeliveries(jxlt_context* ctx) {
  if (ctx->deferred_dc.pending) {
    const int rcd = IssueDeferred(ctx, /*wait=*/true);
    if (rcd != JXLT_OK) return rcd;
  }
  for (int kind = 0; kind < 2; kind++) {
    const hipStream_t out_stream = kind ? ctx->copy_stream : ctx->dc_copy_stream;
    if (ctx->deliver_by_query[kind]) {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        const hipError_t e = hipStreamQuery(out_stream);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) {
          ctx->error = std::string("section hand-over: ") + hipGetErrorString(e);
          return JXLT_ERR_NO_DEVICE;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
          ctx->error = "section hand-over: timed out";
          return JXLT_ERR_INTERNAL;
        }
        if (ctx->throughput_waits) {
          struct timespec ts = {0, 8000};
          nanosleep(&ts, nullptr);
        }
      }
      ctx->deliver_by_query[kind] = false;
      continue;
    }
    const int rc = WaitWord(ctx, &ctx->mail.p->delivered_seq[kind][0], ctx->deliver_seq[kind], out_stream, "section hand-over");
    if (rc != JXLT_OK) return rc;
  }
  return JXLT_OK;
}

// publish_kernel on `stream`: up to kPublishSegments (device source, host destination, dwords) pairs, an optional
// 64-bit word, then `seq` to the host word `flag`.
int EnqueuePublish(jxlt_context* ctx, hipStream_t stream, const PublishSeg* segs, int nsegs, const unsigned long long* src64,
                   unsigned long long* dst64, uint32_t* flag, uint32_t seq, uint32_t* flag2) {
  PublishArgs P;
  memset(&P, 0, sizeof(P));
  for (int i = 0; i < nsegs && i < kPublishSegments; i++) {
    P.src[i] = static_cast<const uint32_t*>(segs[i].src);
    P.dst[i] = static_cast<uint32_t*>(segs[i].dst);
    P.words[i] = (uint32_t)segs[i].words;
  }
  P.src64 = src64;
  P.dst64 = dst64;
  P.flag = flag;
  P.flag2 = flag2;
  P.seq = seq;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(kPublishThreads), 0, stream, P);
  HIP_TRY(ctx, hipGetLastError());
  return JXLT_OK;
}

int EnqueuePipeline(jxlt_context* ctx, const jxlt_params* params) {
  if (!ctx || !params) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->planes[0]) {
    ctx->error = "no image set";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  if (!(params->distance > 0) || !(params->scale > 0) || params->x_qm_scale < 2 || params->x_qm_scale > 5) {
    ctx->error = "invalid encode parameters";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  // A new encode begins: whatever the previous one left on the context -- "encoded", a refusal of its values -- ends
  // HERE, not at the end of a successful enqueue: a frame that fails on its way in (out of memory, a HIP error) must
  // not be reported with the previous frame's JXLT_ERR_UNSUPPORTED.
  ctx->encoded = false;
  ctx->overflow_checked = true;
  ctx->encode_status = JXLT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->deliveries_pending) {
    // (sections of the previous encode may still be leaving the blobs this encode is about to overwrite)
    const int rcw = WaitDeliveries(ctx);
    if (rcw != JXLT_OK) return rcw;
    ctx->deliveries_pending = false;
  }
  // (the previous encode's DC-group sections may have been packed on their own stream and never handed over: this
  // encode's kernels overwrite what that packing reads)
  if (ctx->pack[0].stream == ctx->dc_pack_stream && ctx->pack[0].launches > 0)
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->pack[0].launch_done[ctx->pack[0].launches - 1], 0));
  const uint32_t frame_seq = ++ctx->seq;  // (what this encode's publish kernels store to the host's sequence words)
  const FrameGeom g = MakeGeom(ctx->xsize, ctx->ysize);
  const size_t nblocks = (size_t)g.xsize_blocks * g.ysize_blocks;
  const size_t ntiles = (size_t)g.xsize_tiles * g.ysize_tiles;
  const size_t ngroups = (size_t)g.xsize_groups * g.ysize_groups;
  const bool debug = (params->flags & JXLT_FLAG_DEBUG_DUMP) != 0;
  int rc;
#define ENSURE(buf, n) if ((rc = EnsureDevice(ctx, &ctx->buf, (n))) != JXLT_OK) return rc
  for (int c = 0; c < 3; c++) {
    ENSURE(quant_dc[c], nblocks);
    ENSURE(nzgrid[c], nblocks);
    if (debug) ENSURE(dbg_xyb[c], nblocks * 64);
  }
  ENSURE(raw_quant, nblocks);
  ENSURE(strategy, nblocks);
  ENSURE(blk_nz, nblocks * 3);
  ENSURE(blk_nscan, nblocks * 3);
  ENSURE(blk_nzmask, nblocks * 6);
  ENSURE(ytox, ntiles);
  ENSURE(ytob, ntiles);
  ENSURE(coef_scan, nblocks * 3 * 64);
  ENSURE(group_ntok, ngroups);
  ENSURE(group_off, ngroups + 1);
  ENSURE(hist, 2 * 64 * 64);
  const size_t ndc = ((ctx->xsize + 2047) / 2048) * ((ctx->ysize + 2047) / 2048);
  // records per DC group, worst case: 2 + 3nb + 2nt + 2nb + nb with nb = 65536, nt = 1024
  const size_t kDcStride = 6 * 65536 + 2 * 1024 + 8;
  ENSURE(dc_records, ndc * kDcStride * 3 + 16);  // (+ slack: tiles are staged with aligned dword loads)
  ENSURE(dc_nac, ndc);
  ENSURE(dc_chain_summary, ndc * kDcChainChunks);
  ENSURE(overflow_tiles, ntiles);
  ENSURE(dc_count, ndc);
  ENSURE(dc_rec_off, ndc + 1);
  if (ctx->dc_rec_off_n != ndc) {
    std::vector<uint64_t> off(ndc + 1);
    for (size_t i = 0; i <= ndc; i++) off[i] = i * kDcStride;
    HIP_TRY(ctx, hipMemcpy(ctx->dc_rec_off.p, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    ctx->dc_rec_off_n = ndc;
  }
  // worst case: every coefficient of every block is a token, plus one nzeros token
  ENSURE(tokens, nblocks * 3 * 64 * 3 + 16);
  const size_t ncells = ((size_t)g.xsize_blocks / 2 + 1) * ((size_t)g.ysize_blocks / 2 + 1);
  if (debug) {
    ENSURE(dbg_qf, nblocks);
    ENSURE(dbg_mask, nblocks);
    ENSURE(dbg_ent8, ncells * 8);
    HIP_TRY(ctx, hipMemsetAsync(ctx->dbg_ent8.p, 0xFF, ncells * 8 * sizeof(float), ctx->stream));  // NaN
  }
  const bool profile = (params->flags & JXLT_FLAG_PROFILE) != 0;
  if (profile) {
    // (kPhaseClockCopies copies of the 16 phase sums, indexed by the workgroup: every tile adding to ONE set of 16
    // addresses serialised the whole launch -- 786 k atomics on 12 addresses doubled the kernel's time, round 6)
    ENSURE(dbg_phase, 16 * kPhaseClockCopies);
    HIP_TRY(ctx, hipMemsetAsync(ctx->dbg_phase.p, 0, 16 * kPhaseClockCopies * sizeof(unsigned long long), ctx->stream));
  }
#undef ENSURE

  if (ctx->tab_scale != params->scale) {
    // Pageable source: the copy is staged by the runtime before the call returns.
    DeviceTables host_tab;
    BuildDeviceTables(params->scale, &host_tab);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_tab, &host_tab, sizeof(host_tab), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tab_scale = params->scale;
  }

  TileArgs A;
  memset(&A, 0, sizeof(A));
  for (int c = 0; c < 3; c++) {
    A.planes[c] = ctx->planes[c];
    A.quant_dc[c] = ctx->quant_dc[c].p;
    A.nzgrid[c] = ctx->nzgrid[c].p;
    A.dbg_xyb[c] = debug ? ctx->dbg_xyb[c].p : nullptr;
  }
  A.pitch = ctx->pitch_floats;
  A.pix_stride = ctx->pix_stride;
  A.byteswap = ctx->byteswap;
  A.g = g;
  A.distance = params->distance;
  A.strategy_distance = ctx->strategy_distance > 0.0f ? ctx->strategy_distance : params->distance;
  A.scale = params->scale;
  A.inv_scale = params->inv_scale;
  A.scale_dc = params->scale_dc;
  A.x_qm_mul = XQmMultiplier(params->x_qm_scale);
  SetStrategyScalars(&A);
  A.flags = (params->flags & JXLT_FLAG_FORCE_DCT8) ? 1u : 0u;
  A.flags |= params->flags & 0x1F00u;  // profiling only: truncate tile_kernel after phase n-1 (tools/profile_phases.py)
  A.tab = ctx->d_tab;
  A.raw_quant = ctx->raw_quant.p;
  A.strategy = ctx->strategy.p;
  A.ytox = ctx->ytox.p;
  A.ytob = ctx->ytob.p;
  A.blk_nz = ctx->blk_nz.p;
  A.blk_nscan = ctx->blk_nscan.p;
  A.blk_nzmask = ctx->blk_nzmask.p;
  A.coef_scan = ctx->coef_scan.p;
  A.group_ntok = ctx->group_ntok.p;
  A.dc_nac = ctx->dc_nac.p;
  A.dbg_qf = debug ? ctx->dbg_qf.p : nullptr;
  A.dbg_mask = debug ? ctx->dbg_mask.p : nullptr;
  A.dbg_ent8 = debug ? ctx->dbg_ent8.p : nullptr;
  A.dbg_phase = profile ? ctx->dbg_phase.p : nullptr;

  TokenArgs K;
  memset(&K, 0, sizeof(K));
  K.g = g;
  K.tab = ctx->d_tab;
  K.strategy = ctx->strategy.p;
  for (int c = 0; c < 3; c++) K.nzgrid[c] = ctx->nzgrid[c].p;
  K.blk_nz = ctx->blk_nz.p;
  K.blk_nscan = ctx->blk_nscan.p;
  K.blk_nzmask = ctx->blk_nzmask.p;
  K.coef_scan = ctx->coef_scan.p;
  K.group_ntok = ctx->group_ntok.p;
  K.group_tok_offset = ctx->group_off.p;
  K.tokens = ctx->tokens.p;
  K.histogram = ctx->hist.p;

  // A frame that is still in page-locked host memory (jxlt_image_attach_host*) is processed in rows of DC groups
  // (2048 pixel rows) while it arrives.  A slab of whole DC-group rows is a frame of its own to tile_kernel
  // (nothing crosses a group boundary): same code, base pointers moved to the slab.
  //   upload stream the rows come over PCIe one by one
  //   main stream   tile_kernel(row 0), tile_kernel(row 1), ... each launch waits for its rows only
  //   aux stream    for every row, as soon as its tile_kernel is done: DC-group tokenisation, token offsets (scan
  //                 chained to the previous row's total), token_kernel
  // so that only the last row's kernels are left when the last byte has arrived.
  // A frame that already is in device memory is ONE launch of each kernel: tile_kernel fills every CU's LDS and
  // register file, so nothing can run beside it, and row-sized launches only add tails (measured at 16384^2:
  // eight tile_kernel launches 5.7 ms instead of 5.3, eight token_kernel launches 1.46 ms instead of 0.82).
  const bool from_host = ctx->host_src_kind != 0;
  const size_t xdc = (ctx->xsize + 2047) / 2048;
  // Pieces (y0, rows) in upload order.  Host frames: whole rows of DC groups, and the LAST row of DC groups in
  // rows of groups (256 pixel rows), so that what is left to compute when the last byte has arrived is a
  // sixteenth of a row's tile_kernel, not all of it.  A piece never crosses a row of DC groups.
  struct Piece {
    size_t y0, rows;
    bool ends_dc_row;  // the tokenisation of its row of DC groups can start behind it
  };
  std::vector<Piece> pieces;
  if (!from_host) {
    pieces.push_back({0, ctx->ysize, true});
  } else {
    const size_t last_row_y0 = ((ctx->ysize - 1) / 2048) * 2048;
    for (size_t y = 0; y < last_row_y0; y += 2048) pieces.push_back({y, 2048, true});
    for (size_t y = last_row_y0; y < ctx->ysize; y += 256)
      pieces.push_back({y, std::min<size_t>(256, ctx->ysize - y), y + 256 >= ctx->ysize});
  }
  const size_t nslabs = pieces.size();
  {
    int rc3;
    // (+ 1: the frame's count of tiles with values the format cannot carry, TileArgs::unsupported)
    if ((rc3 = EnsureDevice(ctx, &ctx->lut_overflow, nslabs + 1)) != JXLT_OK) return rc3;
    if ((rc3 = EnsurePinned(ctx, &ctx->h_lut_overflow, nslabs + 1)) != JXLT_OK) return rc3;
  }
  A.lut_overflow = ctx->lut_overflow.p;
  A.overflow_tiles = ctx->overflow_tiles.p;
  A.unsupported = ctx->lut_overflow.p + nslabs;
  // the frame's counters and histograms start at zero: ONE small kernel (four hipMemsetAsync were four fill kernels,
  // 5-9 us apart, in front of every frame's first tile_kernel launch).  The stage event stands in FRONT of it (an event
  // between it and tile_kernel is 6 us of nothing on the stream): "tile_kernel" of jxlt_kernel_times includes its 3 us.
  // A lane of a batch (jxlt_context_set_wait_mode(ctx, 1)), resident frame of up to 1024 groups: EVERY kernel of the frame on
  // the main stream, in order, and no stage events -- the only events left are the two the sections' copies wait for.
  // What a frame gains from its DC-group kernels, its tile plans and its tokenisation running side by side on three
  // streams (one frame at a time: 0.03-0.05 ms) a batch does not need -- the other lanes' frames fill the device --, and
  // the events that tie the streams together are packets the device works through one after the other: 48 resident
  // 3840x2160 frames over 4 / 6 / 8 lanes 37.8-38.1 / 38.1-38.3 / 37.3-37.9 -> 43.1-44.2 / 45.0-45.8 / 45.1-46.8 GP/s
  // (round 6, same box, alternating; with the stage events kept: 40.0-41.8 / 44.8-46.3 / 44.3-46.1).
  const bool lane_frame = ctx->throughput_waits && !from_host && ngroups <= 1024 && ndc <= 1024;
  if (!lane_frame) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  {
    ClearArgs C;
    C.p[0] = ctx->group_ntok.p;
    C.n[0] = (uint32_t)ngroups;
    C.p[1] = ctx->hist.p;
    C.n[1] = 2 * 64 * 64;
    C.p[2] = ctx->dc_nac.p;
    C.n[2] = (uint32_t)ndc;
    C.p[3] = ctx->lut_overflow.p;
    C.n[3] = (uint32_t)nslabs + 1;
    const uint32_t most = std::max(std::max(C.n[0], C.n[1]), std::max(C.n[2], C.n[3]));
    hipLaunchKernelGGL(clear_counters_kernel, dim3((most + 255) / 256), dim3(256), 0, ctx->stream, C);
    HIP_TRY(ctx, hipGetLastError());
  }
  ctx->overflow_slabs = nslabs;
  while (ctx->slab_ready.size() < nslabs || ctx->tile_done.size() < nslabs) {
    hipEvent_t ev = nullptr;
    std::vector<hipEvent_t>& v = ctx->slab_ready.size() < nslabs ? ctx->slab_ready : ctx->tile_done;
    HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    v.push_back(ev);
  }
  int rc2;
  if ((rc2 = EnsurePinned(ctx, &ctx->h_hist, 2 * 64 * 64)) != JXLT_OK) return rc2;
  if ((rc2 = EnsurePinned(ctx, &ctx->h_group_off, 2 * (ngroups + 1))) != JXLT_OK) return rc2;
  DcArgs D;
  memset(&D, 0, sizeof(D));
  D.g = g;
  D.tab = ctx->d_tab;
  for (int c = 0; c < 3; c++) D.quant_dc[c] = ctx->quant_dc[c].p;
  D.raw_quant = ctx->raw_quant.p;
  D.strategy = ctx->strategy.p;
  D.ytox = ctx->ytox.p;
  D.ytob = ctx->ytob.p;
  D.dc_nac = ctx->dc_nac.p;
  D.dc_rec_offset = ctx->dc_rec_off.p;
  D.records = ctx->dc_records.p;
  D.dc_count = ctx->dc_count.p;
  D.histogram = ctx->hist.p + 64 * 64;
  D.chain_summary = ctx->dc_chain_summary.p;
  const size_t row_bytes = ctx->xsize * sizeof(float);
  const size_t ndc_rows = (ctx->ysize + 2047) / 2048;
  const hipStream_t tok_stream = nslabs == 1 ? ctx->stream : ctx->aux_stream;  // (one launch: nothing to overlap)
  size_t dc_rows_done = 0;  // rows of DC groups whose tokenisation has been queued
  bool merged_hist_publish = false;  // (throughput mode, small frames: the DC histogram leaves with the AC histogram)
  bool dc_kernels_beside = false;    // (the DC-group kernels run on a stream of their own, beside token_kernel)
  for (size_t sl = 0; sl < nslabs; sl++) {
    const size_t y0 = pieces[sl].y0, rows = pieces[sl].rows, y1 = y0 + rows;
    if (from_host) {
      if (ctx->host_src_kind == 1) {
        for (int c = 0; c < 3; c++)
          HIP_TRY(ctx, hipMemcpy2DAsync(ctx->own_planes[c].p + y0 * (size_t)ctx->pitch_floats,
                                        (size_t)ctx->pitch_floats * sizeof(float), ctx->host_src[c] + y0 * ctx->host_pitch_bytes,
                                        ctx->host_pitch_bytes, row_bytes, rows, hipMemcpyHostToDevice, ctx->upload_stream));
      } else {
        // bottom-up payload: image rows [y0, y1) are the payload rows [ysize - y1, ysize - y0)
        const size_t off = (ctx->ysize - y1) * ctx->xsize * 3 * sizeof(float);
        HIP_TRY(ctx, hipMemcpyAsync(reinterpret_cast<uint8_t*>(ctx->own_payload.p) + off, ctx->host_src[0] + off,
                                    rows * ctx->xsize * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->upload_stream));
      }
      HIP_TRY(ctx, hipEventRecord(ctx->slab_ready[sl], ctx->upload_stream));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->slab_ready[sl], 0));
    }
    const TileArgs S = nslabs == 1 ? A : SlabTileArgs(A, y0, rows, ctx->pitch_floats, sl);
    (void)y1;
    const unsigned slab_tiles = (unsigned)((size_t)S.g.xsize_tiles * S.g.ysize_tiles);
    // Behind every launch: the tiles it filed because a quantised magnitude did not fit the root table of its
    // entropy estimates, again with computed roots (enc_ac_strategy.cc:118-126 takes a Sqrt per coefficient) -- a
    // small fixed grid whose workgroups usually find an empty list and leave.
    const unsigned redo_grid = std::min<unsigned>(slab_tiles, kRedoGrid);
#ifdef JXLT_TIMING_MARKS
    if (debug)
#else
    if (debug || profile)
#endif
      hipLaunchKernelGGL(tile12_kernel_debug, dim3(slab_tiles), dim3(kTile12Threads), 0, ctx->stream, S);
    else
      hipLaunchKernelGGL(tile12_kernel, dim3(slab_tiles), dim3(kTile12Threads), 0, ctx->stream, S);
    hipLaunchKernelGGL(tile12_kernel_redo, dim3(redo_grid), dim3(kTile12Threads), 0, ctx->stream, S);
    // (the counts of redone tiles leave with the DC histogram, below)
    // (an event record between two kernels of a stream is a barrier packet of its own: ~6 us in which the stream runs
    // nothing, where two kernels queued back to back follow each other within 0.1 us -- kernel trace, round 6.  The
    // last piece's "tiles done" is therefore the stage event itself, and the records below stand BEHIND the small
    // kernels that carry results to the host, not in front of them.)
    hipEvent_t tiles_done = ctx->tile_done[sl];
    if (sl + 1 == nslabs) {
      tiles_done = ctx->ev[1];
      if (!lane_frame) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    } else {
      HIP_TRY(ctx, hipEventRecord(ctx->tile_done[sl], ctx->stream));
    }
    if (!pieces[sl].ends_dc_row) continue;
    // ---- tokenisation of the row(s) of DC groups this piece completes, on the aux stream
    if (tok_stream != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(tok_stream, tiles_done, 0));
    // DC groups first: their histogram leaves for the host as soon as the last row's is complete, so that the DC
    // code is built while token_kernel is still running
    const size_t dc_row0 = dc_rows_done, dc_row1 = nslabs == 1 ? ndc_rows : dc_row0 + 1;
    dc_rows_done = dc_row1;
    const size_t slab_dc = (dc_row1 - dc_row0) * xdc;  // DC groups of this launch
    D.dcg_first = (int)(dc_row0 * xdc);
    // (a resident frame: the element-wise kernel and the two chain kernels do not depend on each other and none of
    // them fills the chip -- side by side on two streams)
    const bool split = nslabs == 1 && !lane_frame;
    ctx->dc_elementwise_split = split;
    const hipStream_t elem_stream = split ? ctx->aux_stream : tok_stream;
    if (split) HIP_TRY(ctx, hipStreamWaitEvent(elem_stream, tiles_done, 0));
    hipLaunchKernelGGL(dc_elementwise_kernel, dim3((unsigned)(slab_dc * kDcParts)), dim3(256), 0, elem_stream, D);
    if (split) HIP_TRY(ctx, hipEventRecord(ctx->dc_elementwise_done, elem_stream));
    // A resident frame of up to 1024 groups (8192^2): the two chain kernels and the histogram's publication on a
    // stream of their own, submitted in front of token_kernel but not waited for by it -- token_kernel is short there
    // (0.06-0.19 ms) and starts 0.03 ms earlier (4096^2: 0.65-0.69 -> 0.64 ms, 8192^2: 1.59 -> 1.55-1.57).  Larger
    // frames keep the DC-group kernels in FRONT of token_kernel: beside it they do not get a CU before its
    // workgroups retire, the DC histogram arrives with the AC histogram (16384^2: at 4.52 instead of 4.09 ms) and the
    // DC code is built behind token_kernel instead of under it (5.36 against 5.18 ms).
    const bool beside = nslabs == 1 && split && ngroups <= 1024;
    const hipStream_t chain_stream = beside ? ctx->upload_stream : tok_stream;
    if (beside) HIP_TRY(ctx, hipStreamWaitEvent(chain_stream, tiles_done, 0));
    hipLaunchKernelGGL(dc_chain_summary_kernel, dim3((unsigned)(slab_dc * kDcChainChunks)), dim3(kDcChainThreads), 0,
                       chain_stream, D);
    hipLaunchKernelGGL(dc_chain_kernel, dim3((unsigned)std::min<size_t>(slab_dc * kDcChainChunks, kDcChainGrid)), dim3(kDcChainThreads), 0,
                       chain_stream, D, (int)(slab_dc * kDcChainChunks));
    if (lane_frame) {
      merged_hist_publish = true;  // (everything of the frame on one stream, in order: no events between its kernels)
    } else if (beside && ctx->throughput_waits) {
      // (a lane of a batch: the DC histogram leaves WITH the AC histogram, one publication for the two -- token_kernel
      // takes ~10 us on frames this small, and the seven microseconds of a publish kernel count where a batch of small
      // frames is bound by the sum of its small launches, round 6)
      merged_hist_publish = true;
      HIP_TRY(ctx, hipEventRecord(ctx->dc_kernels_done, chain_stream));
    } else if (beside) {
      HIP_TRY(ctx, hipStreamWaitEvent(chain_stream, ctx->dc_elementwise_done, 0));
      const PublishSeg segs[2] = {{ctx->hist.p + 64 * 64, ctx->h_hist.p + 64 * 64, 64 * 64},
                                  {ctx->lut_overflow.p, ctx->h_lut_overflow.p, nslabs + 1}};
      const int rcp = EnqueuePublish(ctx, chain_stream, segs, 2, nullptr, nullptr, &ctx->mail.p->dc_hist_seq, frame_seq);
      if (rcp != JXLT_OK) return rcp;
      ctx->dc_hist_stream = chain_stream;
      HIP_TRY(ctx, hipEventRecord(ctx->dc_kernels_done, chain_stream));
    } else if (sl + 1 == nslabs) {
      // The DC histogram (and the counts of the tiles redone with computed roots) leaves IN FRONT of token_kernel: one
      // small kernel stores both to the host's page-locked memory and then the frame's sequence number to the word
      // the host polls (~6 us on the stream).  Beside token_kernel -- on the copy stream, where rounds 2-3 had the
      // download -- the kernel does not get a wave slot before token_kernel's workgroups begin to retire: the
      // histogram arrived 0.4 ms late and the DC code was built behind the AC code (round 4, first version).
      if (split) HIP_TRY(ctx, hipStreamWaitEvent(tok_stream, ctx->dc_elementwise_done, 0));
      const PublishSeg segs[2] = {{ctx->hist.p + 64 * 64, ctx->h_hist.p + 64 * 64, 64 * 64},
                                  {ctx->lut_overflow.p, ctx->h_lut_overflow.p, nslabs + 1}};
      const int rcp = EnqueuePublish(ctx, tok_stream, segs, 2, nullptr, nullptr, &ctx->mail.p->dc_hist_seq, frame_seq);
      if (rcp != JXLT_OK) return rcp;
      ctx->dc_hist_stream = tok_stream;
      HIP_TRY(ctx, hipEventRecord(ctx->dc_kernels_done, tok_stream));  // (behind the publication)
    }
    const size_t ty0 = dc_row0 * 2048, ty1 = std::min(ctx->ysize, dc_row1 * 2048);  // pixel rows being tokenised
    const size_t g0 = (ty0 / 256) * (size_t)g.xsize_groups;
    const size_t ng = ((ty1 - ty0 + 255) / 256) * (size_t)g.xsize_groups;
    // (every token_kernel workgroup finds its group's token offset itself: the counts of all groups before it,
    // whichever launch tokenised them, are final by now)
    K.group_first = (int)g0;
    if ((uint32_t)g.xsize_blocks > kTokenNarrowWidth)
      hipLaunchKernelGGL(token_kernel_wide, dim3((unsigned)ng), dim3(kTokenThreads), 0, tok_stream, K);
    else
      hipLaunchKernelGGL(token_kernel, dim3((unsigned)ng), dim3(kTokenThreads), 0, tok_stream, K);
    dc_kernels_beside = beside;
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->host_src_kind = 0;  // the frame is resident now (a redo with exact roots must not fetch it again)
  // AC histogram + total token count leave right behind the last token_kernel (publish_kernel: no copy command, no
  // event -- the host polls the sequence word)
  TraceMark(ctx, "token_kernel done", tok_stream);
  if (merged_hist_publish) {
    // (both histograms -- they lie behind each other -- and the counts of redone tiles; the DC-group kernels have run)
    if (!lane_frame) {
      HIP_TRY(ctx, hipStreamWaitEvent(tok_stream, ctx->dc_kernels_done, 0));
      HIP_TRY(ctx, hipStreamWaitEvent(tok_stream, ctx->dc_elementwise_done, 0));
    }
    const PublishSeg segs[2] = {{ctx->hist.p, ctx->h_hist.p, 2 * 64 * 64}, {ctx->lut_overflow.p, ctx->h_lut_overflow.p, nslabs + 1}};
    const int rcp = EnqueuePublish(ctx, tok_stream, segs, 2, reinterpret_cast<const unsigned long long*>(ctx->group_off.p + ngroups),
                                   &ctx->mail.p->token_total, &ctx->mail.p->ac_hist_seq, frame_seq, &ctx->mail.p->dc_hist_seq);
    if (rcp != JXLT_OK) return rcp;
    ctx->ac_hist_stream = ctx->dc_hist_stream = tok_stream;
  } else {
    const PublishSeg seg = {ctx->hist.p, ctx->h_hist.p, 64 * 64};
    const int rcp = EnqueuePublish(ctx, tok_stream, &seg, 1, reinterpret_cast<const unsigned long long*>(ctx->group_off.p + ngroups),
                                   &ctx->mail.p->token_total, &ctx->mail.p->ac_hist_seq, frame_seq);
    if (rcp != JXLT_OK) return rcp;
    ctx->ac_hist_stream = tok_stream;
  }
  // (the tokenisation's stage event, behind the histogram's publication: "tokenisation_after_tile_kernel" of
  // jxlt_kernel_times includes those ~6 us)
  if (!lane_frame) HIP_TRY(ctx, hipEventRecord(ctx->aux_done, tok_stream));
  // (whatever follows on the main stream -- the sections' packing -- reads what the DC-group kernels wrote; behind the
  // AC histogram's publication, which does not)
  if (dc_kernels_beside && !merged_hist_publish) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->dc_kernels_done, 0));
  // whatever is queued on the main stream from here on (section packing) comes after the tokenisation
  if (tok_stream != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->aux_done, 0));
  // (the DC-group sections' packing reads what dc_elementwise_kernel wrote)
  if (nslabs == 1 && ctx->dc_elementwise_split) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->dc_elementwise_done, 0));
  ctx->geom = g;
  // The tile plan of the AC sections needs the groups' token offsets only: it runs now, behind the histogram's
  // way to the host, while the host builds the codes (upper bound of the record count: the buffer's capacity).
  ctx->pack[0].planned = ctx->pack[1].planned = false;
  // Where the sections are packed: the AC sections behind token_kernel on the main stream; the DC-group sections on
  // their own stream, behind the DC-group kernels only (experiment knob JXLT_DC_PACK_STREAM=0: on
  // the main stream as well, rounds 1-3).
  static const int dc_own_stream_knob = [] {
    const char* e = getenv("JXLT_DC_PACK_STREAM");
    return e ? atoi(e) : -1;
  }();
  // (above 1024 groups, where the DC-group kernels stand in front of token_kernel: 16384^2 5.22-5.27 -> 5.18-5.22 ms;
  // below, the DC-group sections' packing is short and the extra stream costs more than it saves, 8192^2 1.55 -> 1.57-1.60)
  const bool dc_own_stream = dc_own_stream_knob >= 0 ? dc_own_stream_knob != 0 : ngroups > 1024;
  ctx->pack[1].stream = ctx->stream;
  ctx->pack[0].stream = dc_own_stream ? ctx->dc_pack_stream : ctx->stream;
  if (dc_own_stream) {
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->dc_pack_stream, ctx->dc_kernels_done, 0));
    if (nslabs == 1 && ctx->dc_elementwise_split) HIP_TRY(ctx, hipStreamWaitEvent(ctx->dc_pack_stream, ctx->dc_elementwise_done, 0));
  }
  {
    // (one launch for the frame: the auxiliary stream is idle, and on the main stream the plan's three small
    // kernels would stand in front of the DC-group sections' packing, which the AC measuring pass queues behind)
    hipStream_t plan_stream = ctx->stream;
    const bool both_plans_at_once = tok_stream == ctx->stream && ctx->throughput_waits && ngroups <= 1024 && ndc <= 1024;
    if (both_plans_at_once && lane_frame) {
      const int rcb = EnqueuePlanBoth(ctx, ctx->dc_records.cap / 3, ctx->tokens.cap / 3, ctx->stream);
      if (rcb != JXLT_OK) return rcb;
    } else if (both_plans_at_once) {
      // (a lane of a batch: the two plans in one launch, behind the tokenisation)
      plan_stream = ctx->aux_stream;
      HIP_TRY(ctx, hipStreamWaitEvent(plan_stream, ctx->dc_kernels_done, 0));
      HIP_TRY(ctx, hipStreamWaitEvent(plan_stream, ctx->aux_done, 0));
      const int rcb = EnqueuePlanBoth(ctx, ctx->dc_records.cap / 3, ctx->tokens.cap / 3, plan_stream);
      if (rcb != JXLT_OK) return rcb;
    } else if (tok_stream == ctx->stream) {
      plan_stream = ctx->aux_stream;
      // (the DC-group sections' plan first, behind the DC-group kernels and beside token_kernel: for a small frame
      // those sections' packing is what the frame waits for last, and the plan is a third of its launches)
      {
        HIP_TRY(ctx, hipStreamWaitEvent(plan_stream, ctx->dc_kernels_done, 0));
        const int rcd = EnqueuePlan(ctx, 0, ctx->dc_records.cap / 3, plan_stream);
        if (rcd != JXLT_OK) return rcd;
      }
      HIP_TRY(ctx, hipStreamWaitEvent(plan_stream, ctx->aux_done, 0));
    }
    if (!both_plans_at_once) {
      const int rcp = EnqueuePlan(ctx, 1, ctx->tokens.cap / 3, plan_stream);
      if (rcp != JXLT_OK) return rcp;
    }
  }
  ctx->encoded = true;
  ctx->offsets_fetched = false;
  ctx->pack[0].measured_sections = ctx->pack[1].measured_sections = 0;
  ctx->pack[0].launches = ctx->pack[1].launches = 0;
  ctx->delivered_kinds = 0;
  ctx->last_flags = params->flags;
  ctx->profiled = !lane_frame;  // (the stage events are recorded for every frame but a batch lane's)
  ctx->last_params = *params;
  ctx->overflow_checked = false;
  ctx->encode_status = JXLT_OK;
  ctx->copy_calls = 0;
  ctx->longest_copy_call_us = 0.0f;
  return JXLT_OK;
}

// First host synchronisation point after an enqueue: how many tiles did the device redo with computed roots
// (statistics only: jxlt_encode_stats; the redo itself needs nothing from the host).
int ResolveRootTableOverflow(jxlt_context* ctx) {
  if (!ctx->encoded) return JXLT_OK;
  if (ctx->overflow_checked) {
    // (an encode that met values the format cannot carry stays refused until the next one is enqueued)
    if (ctx->encode_status != JXLT_OK) ctx->error = kUnsupportedValues;
    return ctx->encode_status;
  }
  {  // (the counts arrive with the DC histogram)
    const int rcw = WaitWord(ctx, &ctx->mail.p->dc_hist_seq, ctx->seq, ctx->dc_hist_stream ? ctx->dc_hist_stream : ctx->stream, "device pipeline");
    if (rcw != JXLT_OK) return rcw;
  }
  ctx->overflow_checked = true;
  uint32_t n = 0;
  for (size_t i = 0; i < ctx->overflow_slabs; i++) n += ctx->h_lut_overflow.p[i];
  ctx->tiles_redone = n;
  if (n != 0) ctx->exact_reruns++;
  if (ctx->h_lut_overflow.p[ctx->overflow_slabs] != 0) {
    // A quantised AC coefficient whose token does not fit the format's 16 bits, or a quantised DC value beyond int16
    // (samples around 1e38, infinities): the reference traps on it in debug builds (enc_bit_writer.cc:120) and writes
    // a stream no decoder accepts otherwise.  Refused, for every later call about this encode.
    ctx->encode_status = JXLT_ERR_UNSUPPORTED;
    ctx->error = kUnsupportedValues;
  }
  return ctx->encode_status;
}
}  // namespace jxlt_host

extern "C" {

int jxlt_encode_enqueue(jxlt_context* ctx, const jxlt_params* params) { return EnqueuePipeline(ctx, params); }

int jxlt_encode_stats(jxlt_context* ctx, jxlt_encode_stats_t* out) {
  if (!ctx || !out) return JXLT_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc0 = ResolveRootTableOverflow(ctx);
  if (rc0 != JXLT_OK) return rc0;
  out->tiles_redone_exact_roots = ctx->tiles_redone;
  out->encodes_with_redone_tiles = ctx->exact_reruns;
  out->copy_calls = ctx->copy_calls;
  out->longest_copy_call_us = ctx->longest_copy_call_us;
  out->tiles = (uint32_t)((size_t)ctx->geom.xsize_tiles * ctx->geom.ysize_tiles);
  return JXLT_OK;
}

int jxlt_set_strategy_distance(jxlt_context* ctx, float first_call_distance) {
  if (!ctx || !(first_call_distance >= 0.0f)) return JXLT_ERR_INVALID_ARGUMENT;
  ctx->strategy_distance = first_call_distance;
  return JXLT_OK;
}

int jxlt_context_set_wait_mode(jxlt_context* ctx, int shared_device) {
  if (!ctx) return JXLT_ERR_INVALID_ARGUMENT;
  ctx->throughput_waits = shared_device != 0;
  return JXLT_OK;
}

int jxlt_synchronize(jxlt_context* ctx) {
  if (!ctx) return JXLT_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rc0 = ResolveRootTableOverflow(ctx);
    if (rc0 != JXLT_OK) return rc0;
  }
  if (ctx->deliveries_pending) {
    // The last hand-over kernel stands behind everything the frame has queued (it waits for the last writing
    // launch, which stands behind the whole pipeline on the main stream): its word is the frame's completion, seen
    // without a call into the runtime.
    const int rcw = WaitDeliveries(ctx);
    if (rcw != JXLT_OK) return rcw;
    ctx->deliveries_pending = false;
    TraceDump(ctx);
    // (the section sizes of both kinds have been published by kernels in front of the hand-over's writes)
    for (int kind = 0; kind < 2; kind++) {
      if (ctx->pack[kind].measured_sections == 0) continue;
      const int rcs = WaitSizes(ctx, kind);
      if (rcs != JXLT_OK) return rcs;
    }
    // (the early return: only when every kind that was packed has also been handed over -- the last hand-over then
    // stands behind everything the frame has queued.  A caller that delivers ONE kind only, a shard participant or
    // jxlt_pack_sections(kind), may still have the other kind's packing, plan and publish kernels running: those are
    // waited for below.  ADVICE r4.)
    bool all_delivered = true;
    for (int kind = 0; kind < 2; kind++)
      if (ctx->pack[kind].measured_sections != 0 && !(ctx->delivered_kinds & (1u << kind))) all_delivered = false;
    if (all_delivered && !ctx->copies_pending) return JXLT_OK;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->pack[0].stream == ctx->dc_pack_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->dc_pack_stream));
  if (ctx->copies_pending) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    ctx->copies_pending = false;
  }
  return JXLT_OK;
}

namespace {

// Copies grids, per-group token offsets and histograms to pinned memory; fills *out
// (tokens left NULL).  Leaves group offsets in TOKENS (not bytes) in h_group_off.
int FetchSideInfo(jxlt_context* ctx, jxlt_frame_result* out) {
  {
    const int rc0 = ResolveRootTableOverflow(ctx);
    if (rc0 != JXLT_OK) return rc0;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const FrameGeom& g = ctx->geom;
  const size_t nblocks = (size_t)g.xsize_blocks * g.ysize_blocks;
  const size_t ntiles = (size_t)g.xsize_tiles * g.ysize_tiles;
  const size_t ngroups = (size_t)g.xsize_groups * g.ysize_groups;
  int rc;
#define ENSUREH(buf, n) if ((rc = EnsurePinned(ctx, &ctx->buf, (n))) != JXLT_OK) return rc
  for (int c = 0; c < 3; c++) ENSUREH(h_quant_dc[c], nblocks);
  ENSUREH(h_raw_quant, nblocks);
  ENSUREH(h_strategy, nblocks);
  ENSUREH(h_ytox, ntiles);
  ENSUREH(h_ytob, ntiles);
  ENSUREH(h_group_off, 2 * (ngroups + 1));  // [0, n]: tokens, [n+1, 2n+1]: bytes
  ENSUREH(h_hist, 2 * 64 * 64);
#undef ENSUREH
#define D2H(dst, src, bytes) HIP_TRY(ctx, hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, ctx->stream))
  D2H(ctx->h_group_off.p, ctx->group_off.p, (ngroups + 1) * sizeof(uint64_t));
  D2H(ctx->h_hist.p, ctx->hist.p, 2 * 64 * 64 * sizeof(uint32_t));
  for (int c = 0; c < 3; c++) D2H(ctx->h_quant_dc[c].p, ctx->quant_dc[c].p, nblocks * sizeof(int16_t));
  D2H(ctx->h_raw_quant.p, ctx->raw_quant.p, nblocks);
  D2H(ctx->h_strategy.p, ctx->strategy.p, nblocks);
  D2H(ctx->h_ytox.p, ctx->ytox.p, ntiles);
  D2H(ctx->h_ytob.p, ctx->ytob.p, ntiles);
#undef D2H
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  uint64_t* byte_off = ctx->h_group_off.p + ngroups + 1;
  for (size_t i = 0; i <= ngroups; i++) byte_off[i] = ctx->h_group_off.p[i] * 3;
  if (ctx->h_group_off.p[ngroups] * 3 > ctx->tokens.cap) {
    ctx->error = "internal error: token count exceeds the worst-case bound";
    return JXLT_ERR_INTERNAL;
  }
  out->xsize = ctx->xsize;
  out->ysize = ctx->ysize;
  out->xsize_blocks = g.xsize_blocks;
  out->ysize_blocks = g.ysize_blocks;
  out->xsize_tiles = g.xsize_tiles;
  out->ysize_tiles = g.ysize_tiles;
  out->num_groups = ngroups;
  for (int c = 0; c < 3; c++) out->quant_dc[c] = ctx->h_quant_dc[c].p;
  out->raw_quant_field = ctx->h_raw_quant.p;
  out->ac_strategy = ctx->h_strategy.p;
  out->ytox_map = ctx->h_ytox.p;
  out->ytob_map = ctx->h_ytob.p;
  out->tokens = nullptr;
  out->group_token_offset = byte_off;
  ctx->offsets_fetched = true;
  return JXLT_OK;
}

}  // namespace

int jxlt_fetch_side_info(jxlt_context* ctx, jxlt_frame_result* out, const uint32_t** ac_histograms) {
  if (!ctx || !out) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded) {
    ctx->error = "nothing encoded yet";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  const int rc = FetchSideInfo(ctx, out);
  if (rc == JXLT_OK && ac_histograms) *ac_histograms = ctx->h_hist.p;
  return rc;
}

int jxlt_fetch_result(jxlt_context* ctx, jxlt_frame_result* out) {
  if (!ctx || !out) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded) {
    ctx->error = "nothing encoded yet";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  int rc = FetchSideInfo(ctx, out);
  if (rc != JXLT_OK) return rc;
  const size_t ngroups = out->num_groups;
  const uint64_t total_bytes = out->group_token_offset[ngroups];
  if (ctx->h_tokens.cap < total_bytes &&
      (rc = EnsurePinned(ctx, &ctx->h_tokens, total_bytes + total_bytes / 4 + 4096)) != JXLT_OK)
    return rc;
  if (total_bytes) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tokens.p, ctx->tokens.p, total_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if ((rc = EnsurePinned(ctx, &ctx->h_tokens, 1)) != JXLT_OK) return rc;
  out->tokens = ctx->h_tokens.p;
  return JXLT_OK;
}

int jxlt_fetch_dc_histogram(jxlt_context* ctx, const uint32_t** dc_histogram) {
  if (!ctx || !dc_histogram) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded) {
    ctx->error = "nothing encoded yet";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rc0 = ResolveRootTableOverflow(ctx);
    if (rc0 != JXLT_OK) return rc0;
  }
  // (ResolveRootTableOverflow above has waited for the word that announces the DC histogram)
  *dc_histogram = ctx->h_hist.p + 64 * 64;
  return JXLT_OK;
}

int jxlt_fetch_histograms(jxlt_context* ctx, const uint32_t** ac_histograms, const uint32_t** dc_histograms) {
  if (!ctx) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded) {
    ctx->error = "nothing encoded yet";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rc0 = ResolveRootTableOverflow(ctx);
    if (rc0 != JXLT_OK) return rc0;
  }
  const FrameGeom& g = ctx->geom;
  const size_t ngroups = (size_t)g.xsize_groups * g.ysize_groups;
  // (both halves were published right behind their kernels, see jxlt_encode_enqueue)
  {
    const int rcw = WaitWord(ctx, &ctx->mail.p->ac_hist_seq, ctx->seq, ctx->ac_hist_stream ? ctx->ac_hist_stream : ctx->stream, "tokenisation");
    if (rcw != JXLT_OK) return rcw;
  }
  ctx->h_group_off.p[ngroups] = ctx->mail.p->token_total;
  ctx->offsets_fetched = true;
  if (ac_histograms) *ac_histograms = ctx->h_hist.p;
  if (dc_histograms) *dc_histograms = ctx->h_hist.p + 64 * 64;
  return JXLT_OK;
}


int jxlt_histograms_ready(jxlt_context* ctx) {
  if (!ctx) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded) {
    ctx->error = "jxlt_histograms_ready needs jxlt_encode_enqueue first";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  // (a read of host memory: no call into the runtime, nothing another encoding thread of the process could wait for)
  return *(const volatile uint32_t*)&ctx->mail.p->ac_hist_seq == ctx->seq ? 1 : 0;
}


int jxlt_kernel_times(jxlt_context* ctx, jxlt_kernel_time* out, int cap) {
  if (!ctx || !out || cap < 0) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->profiled) {
    ctx->error = "jxlt_kernel_times: nothing encoded yet, or the last frame was a batch lane's (jxlt_context_set_wait_mode(ctx, 1): "
                 "such frames carry no stage events)";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rc0 = ResolveRootTableOverflow(ctx);
    if (rc0 != JXLT_OK) return rc0;
  }
  HIP_TRY(ctx, hipEventSynchronize(ctx->aux_done));
  // tile_kernel: first launch's start to last launch's end on the main stream (the launches are back to back).
  // The per-row DC / scan / token kernels run beside them on the aux stream; what the frame pays for them is
  // the time they still need after the last tile_kernel launch has finished.
  static const char* kNames[2] = {"tile_kernel", "tokenisation_after_tile_kernel"};
  hipEvent_t from[2] = {ctx->ev[0], ctx->ev[1]}, to[2] = {ctx->ev[1], ctx->aux_done};
  for (int i = 0; i < 2 && i < cap; i++) {
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, from[i], to[i]));
    out[i].name = kNames[i];
    out[i].milliseconds = ms < 0.0f ? 0.0f : ms;
  }
  return 2;
}

int jxlt_debug_fetch(jxlt_context* ctx, int what, void* host_dst, size_t bytes) {
  if (!ctx || !host_dst) return JXLT_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const int rc0 = ResolveRootTableOverflow(ctx);
    if (rc0 != JXLT_OK) return rc0;
  }
  if (what == 7) {  // how many encodes of this context had to be redone with computed roots
    if (bytes != sizeof(uint32_t)) return JXLT_ERR_INVALID_ARGUMENT;
    memcpy(host_dst, &ctx->exact_reruns, sizeof(uint32_t));
    return JXLT_OK;
  }
  if (what == 6) {  // per-phase shader-cycle totals of tile_kernel (JXLT_FLAG_PROFILE)
    if (!(ctx->last_flags & JXLT_FLAG_PROFILE) || bytes != 16 * sizeof(unsigned long long)) {
      ctx->error = "phase counters need JXLT_FLAG_PROFILE and a 128-byte buffer";
      return JXLT_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> copies(16 * kPhaseClockCopies);
    HIP_TRY(ctx, hipMemcpy(copies.data(), ctx->dbg_phase.p, copies.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long* out = static_cast<unsigned long long*>(host_dst);
    for (int i = 0; i < 16; i++) {
      out[i] = 0;
      for (int c = 0; c < kPhaseClockCopies; c++) out[i] += copies[16 * c + i];
    }
    return JXLT_OK;
  }
  if (!(ctx->last_flags & JXLT_FLAG_DEBUG_DUMP)) {
    ctx->error = "last encode was not run with JXLT_FLAG_DEBUG_DUMP";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  const FrameGeom& g = ctx->geom;
  const size_t nblocks = (size_t)g.xsize_blocks * g.ysize_blocks;
  const size_t ncells = ((size_t)g.xsize_blocks / 2 + 1) * ((size_t)g.ysize_blocks / 2 + 1);
  const void* src = nullptr;
  size_t need = 0;
  if (what >= 0 && what <= 2) {
    src = ctx->dbg_xyb[what].p;
    need = nblocks * 64 * sizeof(float);
  } else if (what == 3) {
    src = ctx->dbg_qf.p;
    need = nblocks * sizeof(float);
  } else if (what == 4) {
    src = ctx->dbg_mask.p;
    need = nblocks * sizeof(float);
  } else if (what == 5) {
    src = ctx->dbg_ent8.p;
    need = ncells * 8 * sizeof(float);
  }
  if (!src || bytes != need) {
    ctx->error = "bad debug selector or size";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(host_dst, src, need, hipMemcpyDeviceToHost));
  return JXLT_OK;
}


}  // extern "C"
