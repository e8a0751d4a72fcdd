// jxlt_context.h -- what the translation units of libjxltiny_hip.so share (not part of the C ABI: see
// include/jxl_tiny_amd.h): the context, its buffers, the device-memory cache, and the helpers one unit offers the others.
//
//   jxlt_capi_context.hip  contexts, streams, memory, frames in / onto the device, the output buffer
//   jxlt_capi_encode.hip   the device pipeline of a frame (tile, DC-group and token kernels), what the host waits
//                          for (polled words), results, statistics, debug outputs
//   jxlt_capi_pack.hip     the section packing stage and the hand-over of the packed sections
// Each unit includes exactly the kernel headers it launches (a kernel is defined in one unit of the library).
#ifndef JXLT_CONTEXT_H_
#define JXLT_CONTEXT_H_

#include <hip/hip_runtime.h>
#include <ctype.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jxl_tiny_amd_testing.h"
#include "jxlt_device_common.h"

namespace jxlt_host {

// Largest frame of the device path, in 8x8 blocks.  The kernels index blocks -- and up to twelve 32-bit words per
// block (jxlt_token_kernel.h: mask_at) -- with 32 bits; coefficients (192 per block) are indexed with 64.
// The limit is what the test-suite exercises (test_frame_above_one_gigapixel: 23.1 M blocks, which crosses
// the 32-bit coefficient index) rounded up to the next power of two, not what the index
// widths would allow on paper (2^28): 2^25 blocks = 2.1 Gpixel, e.g. 46 340 x 46 340.
constexpr size_t kMaxFrameBlocks = size_t(1) << 25;

template <typename T>
struct DeviceBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
};

template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t cap = 0;
};

}  // namespace jxlt_host

using jxlt_host::DeviceBuf;
using jxlt_host::PinnedBuf;
using jxlt_dev::FrameGeom;
using jxlt_dev::DeviceTables;
using jxlt_dev::PackTileInfo;
using jxlt_dev::kPackMaxLaunches;

struct jxlt_context {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string error;

  // input image
  DeviceBuf<float> own_planes[3];
  const float* planes[3] = {nullptr, nullptr, nullptr};
  ptrdiff_t pitch_floats = 0;  // floats per row (negative for a bottom-up PFM payload)
  int pix_stride = 1;          // floats between adjacent samples of a plane (3: interleaved RGB)
  int byteswap = 0;            // big-endian samples
  float strategy_distance = 0.0f;  // jxlt_set_strategy_distance (0: each encode's own distance)
  DeviceBuf<float> own_payload;  // jxlt_image_upload_pfm
  size_t xsize = 0, ysize = 0;
  // jxlt_image_attach_host*: the frame is still in the caller's page-locked memory; the next enqueue
  // uploads it in DC-group rows on `upload_stream`, each row's tile_kernel launch waiting for its rows only
  int host_src_kind = 0;  // 0: frame is in device memory, 1: planar planes, 2: PFM payload
  const uint8_t* host_src[3] = {nullptr, nullptr, nullptr};
  size_t host_pitch_bytes = 0;
  hipStream_t upload_stream = nullptr;
  std::vector<hipEvent_t> slab_ready;
  // DC-group tokenisation + AC tokenisation of one row of DC groups run on `aux_stream` as soon as that row's
  // tile_kernel launch is done, i.e. beside the next row's tile_kernel (latency-bound kernels under a VALU-bound one)
  hipStream_t aux_stream = nullptr;
  // The DC-group sections are packed on a stream of their own: their code is ready while token_kernel is still
  // running, and behind token_kernel on the main stream their packing (and the 5 MB they send over the link, 16384^2)
  // would stand in front of the AC sections'.
  hipStream_t dc_pack_stream = nullptr;
  // A hand-over of the DC-group sections asked for before their sizes have arrived (the copy commands need the sizes):
  // kept here and issued as soon as they are there -- from inside whatever wait of the library comes next (WaitWord),
  // at the latest by jxlt_pack_sizes / jxlt_synchronize.  The caller does not block for it and does not have to
  // come back for it.
  struct DeferredDeliver {
    bool pending = false;
    uint8_t* dst = nullptr;
    int end_aligned = 0;
    std::vector<jxlt_section_run> runs;
  } deferred_dc;
  bool in_deferred = false;  // (IssueDeferred is running: its own waits must not start it again)
  // How WaitWord waits (round 6; until then it spun from the first to the last microsecond of every wait -- a full core for
  // the whole tile kernel of every frame).  Per wait site (the `what` literal) the last waits' lengths are remembered:
  // a site whose waits have been LONG (> 0.4 ms, twice in a row) sleeps through the first 70 % of the expected time
  // and spins for the rest -- the word is still seen within a few microseconds of its store.  The memory goes with the
  // frame's geometry.  throughput_waits (the lanes of a batch encoder: several contexts share a GPU and the host's CPU
  // quota): spin for a few microseconds, then poll in short sleeps -- a lane's wake-up latency is covered by the other
  // lanes' frames, and sixteen spinning threads are not.
  struct WaitSite {
    const char* what = nullptr;
    float last_us[2] = {0.0f, 0.0f};
  } wait_sites[8];
  size_t wait_geometry[2] = {0, 0};  // the frame size the remembered waits belong to
  bool throughput_waits = false;
  // (throughput mode, and frames of up to 256 groups: a hand-over's completion is its copy stream having drained -- asked
  // for with hipStreamQuery, between short sleeps in throughput mode -- instead of a publish kernel behind the copies,
  // jxlt_capi_pack.hip CompletionByQuery)
  bool deliver_by_query[2] = {false, false};
  std::vector<hipEvent_t> tile_done;
  hipEvent_t aux_done = nullptr;   // everything queued on aux_stream for the frame (timing enabled: the token tail)
  // the streams that carry the last encode's publications of the DC / AC histogram: what a wait for their words asks
  // for errors, and whose draining means "the word will never come" (for frames that arrive in slabs the
  // tokenisation runs on aux_stream, and the main stream can drain long before the AC histogram is published)
  hipStream_t dc_hist_stream = nullptr, ac_hist_stream = nullptr;

  // pinned staging ring for uploads from pageable memory
  PinnedBuf<uint8_t> stage[2];
  hipEvent_t stage_done[2] = {nullptr, nullptr};

  // constant tables (rebuilt when `scale` changes)
  DeviceTables* d_tab = nullptr;
  float tab_scale = -1.0f;

  // device outputs / intermediates
  DeviceBuf<int16_t> quant_dc[3];
  DeviceBuf<uint8_t> raw_quant, strategy, nzgrid[3], blk_nz, blk_nscan, tokens;
  DeviceBuf<unsigned long long> blk_nzmask;
  DeviceBuf<int8_t> ytox, ytob;
  DeviceBuf<int16_t> coef_scan;
  DeviceBuf<uint32_t> group_ntok;
  DeviceBuf<uint64_t> group_off;
  DeviceBuf<float> dbg_xyb[3], dbg_qf, dbg_mask, dbg_ent8;
  DeviceBuf<unsigned long long> dbg_phase;
  DeviceBuf<uint32_t> hist;  // [0,4096): AC, [4096,8192): DC symbol histograms
  // DC-group record streams (fixed stride per DC group)
  DeviceBuf<uint8_t> dc_records;
  DeviceBuf<uint32_t> dc_nac, dc_count;
  DeviceBuf<uint64_t> dc_rec_off;
  size_t dc_rec_off_n = 0;
  // section packing, [0] = DC groups, [1] = AC groups
  struct PackSet {
    DeviceBuf<uint32_t> code_table, sec_bytes, sec_tiles, tile_bits;
    // sec_byte_off: [nsec + 1] byte offsets of the sections, and right behind them [nsec] 32-bit bit counts -- what
    // the host needs of a measuring pass, in one piece (one download)
    DeviceBuf<uint64_t> sec_byte_off, tile_base;
    DeviceBuf<PackTileInfo> tile_info;
    DeviceBuf<uint8_t> packed;  // the sections at their final byte offsets
    PinnedBuf<uint64_t> h_sec_byte_off;  // (the same layout as sec_byte_off; filled by publish_kernel)
    DeviceBuf<uint32_t> launch_sec_end;  // sections complete behind each writing launch (pack_tile_finalize_kernel)
    uint32_t pack_seq = 0;               // measuring passes of this kind so far: what the sizes' flag carries
    uint32_t* h_launch_sec_end = nullptr;  // host mirror of launch_sec_end (inside the context's HostMail)
    DeviceBuf<unsigned long long> tile_state;  // single pass: what every tile tells the tiles behind it (PackTileState)
    bool streamed = false;          // the last pass of this kind was a single pass (no measuring pass)
    // (single pass: the host turns the sections' bit counts into byte offsets, launch by launch)
    size_t state_tiles = 0;               // tile_state: the block states start behind this many tile states
    hipStream_t stream = nullptr;         // where this kind's packing kernels are queued (set by jxlt_pack_begin)
    int launches_seen = 0;                // launches whose word the host has seen
    uint32_t offsets_done_sections = 0;   // sections whose byte offsets the host has worked out
    uint64_t zeroed_bytes = 0;            // how much of the blob was zeroed in front of the pass
    static size_t SizesWords(size_t nsec) { return nsec + 1 + (nsec + 1) / 2; }
    uint32_t* sec_bits(size_t nsec) const { return reinterpret_cast<uint32_t*>(sec_byte_off.p + nsec + 1); }
    uint32_t* h_sec_bits(size_t nsec) const { return reinterpret_cast<uint32_t*>(h_sec_byte_off.p + nsec + 1); }
    size_t max_tiles = 0;        // of the measuring pass (bounds the writing launches)
    bool writes_queued = false;  // the writing launches of the last measuring pass are queued
    PinnedBuf<uint8_t> h_packed;
    PinnedBuf<uint32_t> h_code_table;  // staging of the caller's table (asynchronous upload needs page-locked memory)
    size_t measured_sections = 0;  // sections of the last measuring pass (0: none for this frame)
    bool planned = false;          // the frame's tile plan (count / scan / plan kernels) has been queued
    // The writing kernels are queued right behind the measuring kernels (they need nothing from
    // the host): launch i covers tiles [launch_t0[i], launch_t0[i + 1]) and signals launch_done[i].
    static constexpr int kMaxLaunches = 8;
    int launches = 0;
    uint32_t launch_t0[kMaxLaunches + 1] = {};
    hipEvent_t launch_done[kMaxLaunches] = {};
    hipEvent_t finalized = nullptr; // the measuring pass's kernels are done (the mirrors' copies wait for it)
    hipEvent_t plan_done = nullptr; // the tile plan, when it was queued on another stream than the measuring pass
    bool plan_elsewhere = false;
  } pack[2];
  PinnedBuf<uint8_t> h_output;  // jxlt_output_buffer
  // What kernels tell the host without a copy command and an event in between (publish_kernel):
  // sequence words in page-locked memory that the host polls, every word in a cache line of its own.
  struct HostMail {
    uint32_t dc_hist_seq;  // = seq: the DC histogram (h_hist + 4096) and the root-table overflow counts are there
    uint32_t pad0[15];
    uint32_t ac_hist_seq;  // = seq: the AC histogram (h_hist) and token_total are there
    uint32_t pad1[15];
    uint32_t sizes_seq[2][16];  // [kind][0] = pack[kind].pack_seq: h_sec_byte_off of that kind is complete
    uint32_t delivered_seq[2][16];  // [kind][0] = deliver_seq[kind]: every hand-over of that kind queued so far has finished
    unsigned long long token_total;  // records of all AC groups (sizes the packing's tile arrays)
    uint32_t launch_sec_end[2][16];  // [kind]: sections complete behind each writing launch (published with the sizes)
    uint32_t stream_seq[2][kPackMaxLaunches][16];  // [kind][launch][0] = pack_seq: that launch of a single pass is done,
                                                   // the bit counts of the sections it completed are in the host's mirror
    uint32_t denormal_probe;  // bits of 3.0f x 2^-147 as the device computes it (jxlt_context_create's self-check)
  };
  PinnedBuf<HostMail> mail;
  uint32_t seq = 0;          // encodes enqueued on this context
  uint32_t deliver_seq[2] = {0, 0};  // hand-overs queued so far, per kind (each kind leaves on a stream of its own)
  bool deliveries_pending = false;
  unsigned delivered_kinds = 0;  // bit k: a hand-over of kind k has been asked for since the last encode was enqueued
  DeviceBuf<uint32_t> deliver_counter;  // (the single pass's look-back statistics, JXLT_TRACE_EVENTS=2: words 16..23)

  // pinned host mirrors
  PinnedBuf<int16_t> h_quant_dc[3];
  PinnedBuf<uint8_t> h_raw_quant, h_strategy, h_tokens;
  PinnedBuf<int8_t> h_ytox, h_ytob;
  PinnedBuf<uint64_t> h_group_off;
  PinnedBuf<uint32_t> h_hist;
  bool offsets_fetched = false;

  FrameGeom geom = {};
  bool encoded = false;
  uint32_t last_flags = 0;

  // profiling
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // jxlt_pack_write: packing on `stream`, chunked copies to the destination on `copy_stream`
  hipStream_t copy_stream = nullptr;
  hipStream_t dc_copy_stream = nullptr;  // the DC-group sections' hand-over (beside the AC sections' on copy_stream)
  bool dc_elementwise_split = false;
  hipEvent_t dc_elementwise_done = nullptr;  // (resident frames: dc_elementwise_kernel runs beside the two chain kernels)
  hipEvent_t dc_kernels_done = nullptr;  // (the small downloads wait for their kernels on the copy stream, not in front of the next kernel)
  // Root-table overflow of tile_kernel (a quantised magnitude >= kSqrtLutSize): the tiles concerned are redone by
  // tile*_kernel_redo right behind it; their number reaches the host with the first synchronisation point.
  DeviceBuf<uint32_t> lut_overflow;    // per tile_kernel launch of the frame: tiles redone with computed roots
  DeviceBuf<uint32_t> overflow_tiles;  // their indices
  DeviceBuf<uint32_t> dc_chain_summary;
  PinnedBuf<uint32_t> h_lut_overflow;
  jxlt_params last_params = {};
  uint32_t copy_calls = 0;            // copy commands issued for the last encode's sections ...
  float longest_copy_call_us = 0.0f;  // ... and the longest of those calls on the host (jxlt_encode_stats)
  bool overflow_checked = true;
  int encode_status = JXLT_OK;  // JXLT_ERR_UNSUPPORTED: the last encode met values the format cannot carry
  size_t overflow_slabs = 0;   // launches of the last encode
  uint32_t exact_reruns = 0;   // encodes of this context in which some tile was redone
  uint32_t tiles_redone = 0;   // ... tiles of the last encode
  bool copies_pending = false;  // (hipMemcpyAsync on the copy stream: the raw-token / debug routes only)
  bool profiled = false;
  bool counted = false;  // DeviceBlockCache knows this context as a living one
  // JXLT_TRACE_EVENTS=1 (tools/): timed events at points of interest of the last encode, printed by jxlt_synchronize
  struct TraceEvent {
    const char* name;
    hipEvent_t ev;
  };
  std::vector<TraceEvent> trace;
  size_t trace_used = 0;
};


namespace jxlt_host {

#define HIP_TRY(ctx, expr)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
      return e_ == hipErrorOutOfMemory ? JXLT_ERR_OUT_OF_MEMORY : JXLT_ERR_NO_DEVICE;       \
    }                                                                                       \
  } while (0)

// Device memory of destroyed contexts is kept for the next context of the process instead of going back to the
// runtime: memory that hipFree has seen and hipMalloc hands out again is SLOWER on this stack (ROCm 7.2, MI355X) --
// a context created after another one of the same frame size was destroyed ran tile_kernel 8 % slower (4.22 ->
// 4.57 ms at 16384^2) and its downloads at half the rate (59 MB in 2.1 instead of 0.96 ms), tools/seq_probe.py; with
// the first context's buffers leaked instead of freed the second was as fast as the first.  Blocks of 1 MB and more,
// per device, handed out again for requests of their size (up to a quarter less).
// What is kept, and for how long (memory a co-resident allocator cannot see must not outlive its use):
//  * default: blocks are kept only while the device has another LIVING context of this library (a pipeline lane that
//    is re-created, a batch encoder's lanes, contexts of several sizes side by side); when the last context of a
//    device is destroyed everything kept for that device goes back to the runtime.  At most 8 GB are held.
//  * JXLT_DEVICE_CACHE_MB=<n> (environment) opts in to keeping up to n MB beyond the last context -- what a process
//    that creates and destroys encoders in a row wants (tools/config_table.py, tools/soak.py); 0 keeps nothing, ever.
//  jxlt_release_cached_memory() returns everything at once in either mode.
class DeviceBlockCache {
 public:
  static DeviceBlockCache& Get() {
    static DeviceBlockCache* cache = new DeviceBlockCache;  // (never destroyed: the runtime may be gone by then)
    return *cache;
  }
  void ContextCreated(int device) {
    std::lock_guard<std::mutex> lock(mu_);
    if (device >= 0) {
      if (live_.size() <= (size_t)device) live_.resize((size_t)device + 1, 0);
      live_[(size_t)device]++;
    }
  }
  // The last context of a device is gone: what was kept for its successors goes back to the runtime, unless the
  // process asked for a cache that outlives its contexts.
  void ContextDestroyed(int device) {
    bool release = false;
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (device >= 0 && (size_t)device < live_.size() && live_[(size_t)device] > 0)
        release = --live_[(size_t)device] == 0 && !persistent_;
    }
    if (release) (void)Release(device);
  }
  void* Take(int device, size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> lock(mu_);
    size_t best = blocks_.size();
    for (size_t i = 0; i < blocks_.size(); i++) {
      const Block& b = blocks_[i];
      if (b.device != device || b.bytes < bytes || b.bytes - bytes > bytes / 4) continue;
      if (best == blocks_.size() || b.bytes < blocks_[best].bytes) best = i;
    }
    if (best == blocks_.size()) return nullptr;
    void* p = blocks_[best].p;
    *got = blocks_[best].bytes;
    total_ -= blocks_[best].bytes;
    blocks_.erase(blocks_.begin() + static_cast<ptrdiff_t>(best));
    return p;
  }
  // (the calling thread's current device is the block's)
  void Give(void* p, size_t bytes) {
    int device = 0;
    if (bytes < kMinBytes || limit_ == 0 || hipGetDevice(&device) != hipSuccess) {
      (void)hipFree(p);
      return;
    }
    std::lock_guard<std::mutex> lock(mu_);
    blocks_.push_back({p, bytes, device});
    total_ += bytes;
    while (total_ > limit_ && !blocks_.empty()) {  // the oldest first
      total_ -= blocks_.front().bytes;
      (void)hipFree(blocks_.front().p);
      blocks_.erase(blocks_.begin());
    }
  }
  // Returns every block of `device` (-1: of every device) to the runtime; the bytes released.
  size_t Release(int device) {
    std::lock_guard<std::mutex> lock(mu_);
    int current = 0;
    const bool have_current = hipGetDevice(&current) == hipSuccess;
    size_t released = 0;
    for (size_t i = 0; i < blocks_.size();) {
      if (device >= 0 && blocks_[i].device != device) {
        i++;
        continue;
      }
      (void)hipSetDevice(blocks_[i].device);
      (void)hipFree(blocks_[i].p);
      released += blocks_[i].bytes;
      total_ -= blocks_[i].bytes;
      blocks_.erase(blocks_.begin() + static_cast<ptrdiff_t>(i));
    }
    if (have_current) (void)hipSetDevice(current);
    return released;
  }

 private:
  DeviceBlockCache() {
    const char* e = getenv("JXLT_DEVICE_CACHE_MB");
    persistent_ = e != nullptr && *e != '\0';
    limit_ = (persistent_ ? static_cast<size_t>(atoll(e)) : size_t(8192)) << 20;
  }
  static constexpr size_t kMinBytes = size_t(1) << 20;
  struct Block {
    void* p;
    size_t bytes;
    int device;
  };
  std::mutex mu_;
  std::vector<Block> blocks_;
  std::vector<int> live_;  // living contexts per device
  size_t total_ = 0, limit_ = 0;
  bool persistent_ = false;  // JXLT_DEVICE_CACHE_MB given: blocks outlive the last context
};

template <typename T>
int EnsureDevice(jxlt_context* ctx, DeviceBuf<T>* b, size_t n) {
  if (b->cap >= n && b->p) return JXLT_OK;
  if (b->p) HIP_TRY(ctx, hipFree(b->p));  // (a buffer that grows: nobody will ask for its old size again)
  b->p = nullptr;
  b->cap = 0;
  const size_t bytes = (n ? n : 1) * sizeof(T);
  size_t got = 0;
  if (void* cached = DeviceBlockCache::Get().Take(ctx->device, bytes, &got)) {
    b->p = static_cast<T*>(cached);
    b->cap = got / sizeof(T);
    return JXLT_OK;
  }
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&b->p), bytes);
  if (e == hipErrorOutOfMemory) {
    // (what destroyed contexts left behind must not stand in the way of a living one)
    (void)hipGetLastError();
    b->p = nullptr;
    if (DeviceBlockCache::Get().Release(ctx->device) != 0) e = hipMalloc(reinterpret_cast<void**>(&b->p), bytes);
  }
  HIP_TRY(ctx, e);
  b->cap = n;
  return JXLT_OK;
}

template <typename T>
int EnsurePinned(jxlt_context* ctx, PinnedBuf<T>* b, size_t n) {
  if (b->cap >= n && b->p) return JXLT_OK;
  if (b->p) HIP_TRY(ctx, hipHostFree(b->p));
  b->p = nullptr;
  b->cap = 0;
  HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&b->p), (n ? n : 1) * sizeof(T), hipHostMallocDefault));
  b->cap = n;
  return JXLT_OK;
}

template <typename T>
void FreeDevice(DeviceBuf<T>* b) {
  if (b->p) DeviceBlockCache::Get().Give(b->p, b->cap * sizeof(T));
  b->p = nullptr;
  b->cap = 0;
}
template <typename T>
void FreePinned(PinnedBuf<T>* b) {
  if (b->p) (void)hipHostFree(b->p);
  b->p = nullptr;
  b->cap = 0;
}

// ---- what one unit offers the others ------------------------------------------------------------------------
// jxlt_capi_encode.hip
int TraceLevel();  // JXLT_TRACE_EVENTS: 1 = device-side event times, 2 = + every copy call and the look-back statistics
bool TraceEventsOn();
void TraceMark(jxlt_context* ctx, const char* name, hipStream_t stream);
void TraceDump(jxlt_context* ctx);
int WaitWord(jxlt_context* ctx, const uint32_t* word, uint32_t want, hipStream_t stream, const char* what);
int WaitDeliveries(jxlt_context* ctx);
struct PublishSeg {
  const void* src;
  void* dst;
  size_t words;
};
int EnqueuePublish(jxlt_context* ctx, hipStream_t stream, const PublishSeg* segs, int nsegs, const unsigned long long* src64,
                   unsigned long long* dst64, uint32_t* flag, uint32_t seq, uint32_t* flag2 = nullptr);
int ResolveRootTableOverflow(jxlt_context* ctx);
// jxlt_capi_pack.hip
int EnqueuePlan(jxlt_context* ctx, int kind, uint64_t rec_bound, hipStream_t stream);
int EnqueuePlanBoth(jxlt_context* ctx, uint64_t dc_rec_bound, uint64_t ac_rec_bound, hipStream_t stream);
int WaitSizes(jxlt_context* ctx, int kind);
int IssueDeferred(jxlt_context* ctx, bool wait);

}  // namespace jxlt_host

#endif  // JXLT_CONTEXT_H_
