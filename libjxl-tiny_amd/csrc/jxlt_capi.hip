// jxlt_capi.hip -- C ABI of libjxltiny_hip.so (see include/jxl_tiny_amd.h).
//
// One jxlt_context = one HIP device + one stream + all device/pinned buffers of
// the per-group pipeline.  A whole frame is processed by these launches on the
// context's stream:
//   tile_kernel  (one workgroup per 64x64 tile)  -> side-band grids, coefficients
//   dc_* kernels (DC-group tokenisation)
//   token_kernel (one workgroup per 256x256 group) -> raw 3-byte token records
// There is no CPU fallback: without a usable HIP device every entry point
// returns JXLT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cmath>

#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/jxl_tiny_amd_testing.h"
#include "jxlt_device.h"
#include "jxlt_host_tables.h"

using namespace jxlt_dev;

namespace {

thread_local std::string g_create_error;

// Largest frame of the device path, in 8x8 blocks.  The kernels index blocks -- and up to twelve 32-bit words per
// block (jxlt_token_kernel.h: mask_at) -- with 32 bits; coefficients (192 per block) are indexed with 64.
// The limit is what the test-suite exercises (test_frame_above_one_gigapixel: 23.1 M blocks, which crosses
// kTokenNarrowBlocks and the 32-bit coefficient index) rounded up to the next power of two, not what the index
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

}  // namespace

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
  // JXLT_WAIT_SLEEP=1: the long wait of a frame -- for the DC histogram, i.e. for tile_kernel -- is slept through when
  // the last frame was the same frame (geometry, parameters): the thread wakes 0.3 ms before the histogram is due
  // and polls from there.  (A polling thread is a CPU; the boxes of this pool grant a process 16 at a time, and a batch encoder or
  // eight ranks of a sharded frame have six to eight such threads.)
  std::chrono::steady_clock::time_point enqueued_at;
  uint64_t wait_key = 0;          // what was enqueued: geometry and parameters
  uint64_t last_wait_key = 0;     // ... and for which frame last_dc_wait_us was measured
  double last_dc_wait_us = 0.0;   // from the end of jxlt_encode_enqueue to the DC histogram's word
  std::vector<hipEvent_t> tile_done;
  hipEvent_t aux_done = nullptr;   // everything queued on aux_stream for the frame (timing enabled: the token tail)

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
  // What kernels tell the host without a copy command and an event in between (publish_kernel, pack_deliver_kernel):
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
  };
  PinnedBuf<HostMail> mail;
  uint32_t seq = 0;          // encodes enqueued on this context
  uint32_t deliver_seq[2] = {0, 0};  // hand-overs queued so far, per kind (each kind leaves on a stream of its own)
  bool deliveries_pending = false;
  DeviceBuf<uint32_t> deliver_counter;

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

namespace {

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
  // (experiment knob: JXLT_PINNED_FLAGS=<hipHostMalloc flags, hex or decimal>)
  static const unsigned pinned_flags = [] {
    const char* e = getenv("JXLT_PINNED_FLAGS");
    return e ? (unsigned)strtoul(e, nullptr, 0) : (unsigned)hipHostMallocDefault;
  }();
  HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&b->p), (n ? n : 1) * sizeof(T), pinned_flags));
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

int CheckImageArgs(jxlt_context* ctx, const void* const planes[3], size_t pitch_bytes, size_t xsize,
                   size_t ysize) {
  if (!ctx) return JXLT_ERR_INVALID_ARGUMENT;
  if (!planes || !planes[0] || !planes[1] || !planes[2] || xsize == 0 || ysize == 0 ||
      xsize > 0x3FFFFFFFull || ysize > 0x3FFFFFFFull || pitch_bytes < xsize * sizeof(float) ||
      pitch_bytes % sizeof(float) != 0) {
    ctx->error = "invalid image arguments";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  if (((xsize + 7) / 8) * ((ysize + 7) / 8) > kMaxFrameBlocks) {
    // (the kernels' 32-bit block indices would reach 2^28 blocks; frames above 2^25 -- 2.1 Gpixel -- are refused
    // because nothing larger has ever been run through them)
    ctx->error = "frames above 2^25 8x8 blocks (2.1 Gpixel) are not supported by the device path";
    return JXLT_ERR_UNSUPPORTED;
  }
  if (xsize <= 8 && ysize <= 8) {
    // The reference traps on images that fit a single 8x8 block.
    ctx->error = "images of at most one 8x8 block are not supported";
    return JXLT_ERR_UNSUPPORTED;
  }
  return JXLT_OK;
}

}  // namespace

extern "C" {

namespace {
__global__ void delay_kernel(unsigned long long cycles) {
  const unsigned long long t0 = clock64();
  while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
// The runtime creates the queue of a copy engine (SDMA) the first time it turns to that engine, inside the
// hipMemcpyAsync call that needs it: 5-7 ms on the host, during which nothing else is issued.  Which engine a copy gets
// depends on what is in flight when it is issued, so a context met such calls in its first frame (+10 ms) and ONCE
// MORE in one of frames 2 to 6 -- the first frame whose DC-group sections leave beside its AC sections: a step of 12 ms
// among steps of 5.2, followed by three or four slow ones (the GPU's clock coming back up), in the driver's warm-up
// or in its timed steps as luck had it (tools/outlier_probe.sh: 5 of 8 runs; JXLT_TRACE_EVENTS names the call; with
// HSA_ENABLE_SDMA=0 no such step ever, but every step 5.7 ms -- blit kernels beside the packing kernels).  So the copy
// commands of a frame are issued once when the first context of a device is made -- device-to-host copies of section
// size on the hand-over streams, side by side, each WAITING for an event of the main stream that has not happened yet
// (mode 2: 2 of 8 runs still met an engine for the first time later), and more of them in flight than a frame ever
// has, over four streams (mode 3, the default: 0 of 16 runs; first frame 11 instead of 22-25 ms).
void CopyWarmup(jxlt_context* ctx, int mode) {
  const size_t n = (size_t)12 << 20;
  uint8_t *dsrc = nullptr, *hdst = nullptr;
  hipEvent_t ev = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&dsrc), 3 * n) == hipSuccess &&
      hipHostMalloc(reinterpret_cast<void**>(&hdst), 3 * n, hipHostMallocDefault) == hipSuccess &&
      hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
    for (int rep = 0; rep < 2; rep++) {
      if (mode >= 2) {
        hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, ctx->stream, 200000ull);  // ~2 ms at 100 MHz
        (void)hipEventRecord(ev, ctx->stream);
        (void)hipStreamWaitEvent(ctx->dc_copy_stream, ev, 0);
        (void)hipStreamWaitEvent(ctx->copy_stream, ev, 0);
      }
      (void)hipMemcpyAsync(hdst + n, dsrc + n, n * 2 / 3, hipMemcpyDefault, ctx->dc_copy_stream);
      (void)hipMemcpyAsync(hdst, dsrc, n / 4, hipMemcpyDefault, ctx->copy_stream);
      (void)hipMemcpyAsync(hdst + n / 4, dsrc + n / 4, n / 2, hipMemcpyDefault, ctx->copy_stream);
      (void)hipMemcpyAsync(hdst + 2 * n, dsrc + 2 * n, n, hipMemcpyDefault, ctx->copy_stream);
      if (mode >= 3) {  // (more copies in flight than a frame ever has: every engine the runtime may turn to)
        (void)hipStreamWaitEvent(ctx->aux_stream, ev, 0);
        (void)hipStreamWaitEvent(ctx->upload_stream, ev, 0);
        const hipStream_t four[4] = {ctx->aux_stream, ctx->upload_stream, ctx->dc_copy_stream, ctx->copy_stream};
        // (twenty more, all issued while the event they wait for is still out: a copy goes to an engine that is idle
        // when it is issued, and a frame at d = 0.5 -- copies of 8 / 16 / 32 MB, longer in flight -- still met new
        // engines after a warm-up of eight: 8.2 instead of 6.7 ms per frame over ten frames)
        for (int k = 0; k < 20; k++)
          (void)hipMemcpyAsync(hdst + (size_t)k * (n / 8), dsrc + (size_t)k * (n / 8), n / 8, hipMemcpyDefault, four[k & 3]);
        // (... and the other direction: frames that come over PCIe are uploaded in rows, several copies in flight)
        for (int k = 0; k < 8; k++)
          (void)hipMemcpyAsync(dsrc + (size_t)k * (n / 8), hdst + (size_t)k * (n / 8), n / 8, hipMemcpyDefault, four[k & 1]);
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamSynchronize(ctx->upload_stream);
      }
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipStreamSynchronize(ctx->copy_stream);
      (void)hipStreamSynchronize(ctx->dc_copy_stream);
    }
  }
  if (ev) (void)hipEventDestroy(ev);
  if (dsrc) (void)hipFree(dsrc);
  if (hdst) (void)hipHostFree(hdst);
  (void)hipGetLastError();
}
}  // namespace

int jxlt_context_create(int device_ordinal, jxlt_context** out) {
  if (!out) return JXLT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_create_error = std::string("no HIP device available: ") +
                     (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return JXLT_ERR_NO_DEVICE;
  }
  if (device_ordinal < 0 || device_ordinal >= count) {
    g_create_error = "device ordinal out of range";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  jxlt_context* ctx = new jxlt_context;
  ctx->device = device_ordinal;
  {
    // (experiment knob JXLT_STREAM_SKEW=<k>: k streams created -- and kept -- in front of the context's own, which shifts
    // the runtime's round-robin assignment of streams to its four hardware queues)
    static const int skew = [] {
      const char* e2 = getenv("JXLT_STREAM_SKEW");
      return e2 ? atoi(e2) : 0;
    }();
    if (hipSetDevice(device_ordinal) == hipSuccess)
      for (int k = 0; k < skew; k++) {
        hipStream_t dummy = nullptr;
        (void)hipStreamCreateWithFlags(&dummy, hipStreamNonBlocking);
      }
  }
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&ctx->d_tab), sizeof(DeviceTables))) != hipSuccess) {
    g_create_error = std::string("context setup failed: ") + hipGetErrorString(e);
    delete ctx;
    return JXLT_ERR_NO_DEVICE;
  }
  // every stream / event of the context; a failure anywhere releases what exists so far
  e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->dc_copy_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    // An ORDINARY stream.  Until the end of round 4 it was created with the device's highest priority -- which bought
    // nothing (the DC-group sections' kernels do not get in beside token_kernel either way) and could
    // cost 18 %: in a process whose FIRST HIP streams are this context's (the context made before the application's
    // own first allocation or kernel), every kernel of the main stream ran slower -- tile12_kernel 4.75 instead of
    // 4.05 ms, the same wave-cycles in 16 % more busy cycles -- as long as that high-priority stream existed
    // (bench.py with BENCH_CONTEXT_FIRST=1, tools/alloc_order_probe.sh; neither the address translation nor the
    // instruction cache counters move).  (experiment knob: JXLT_DC_STREAM_PRIORITY=1)
    static const bool with_priority = [] {
      const char* e2 = getenv("JXLT_DC_STREAM_PRIORITY");
      return e2 && atoi(e2) != 0;
    }();
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    e = with_priority ? hipStreamCreateWithPriority(&ctx->dc_pack_stream, hipStreamNonBlocking, greatest) : hipErrorNotSupported;
    if (e != hipSuccess) {  // (a runtime without stream priorities: an ordinary stream does the same work)
      (void)hipGetLastError();
      e = hipStreamCreateWithFlags(&ctx->dc_pack_stream, hipStreamNonBlocking);
    }
  }
  if (e == hipSuccess) {
    // (experiment knob JXLT_STREAM_ROLES=<five digits>: which of the five streams made above -- in the order copy,
    // DC copy, upload, auxiliary, DC packing, i.e. on the runtime's hardware queues 1, 2, 3, 0, 1 when the main stream
    // is on 0 -- plays which role; "01234" is the order as made)
    const char* roles = getenv("JXLT_STREAM_ROLES");
    if (roles && strlen(roles) == 5) {
      const hipStream_t made[5] = {ctx->copy_stream, ctx->dc_copy_stream, ctx->upload_stream, ctx->aux_stream, ctx->dc_pack_stream};
      bool used[5] = {false, false, false, false, false};
      bool ok = true;
      for (int k = 0; k < 5; k++) {
        const int d = roles[k] - '0';
        if (d < 0 || d > 4 || used[d]) ok = false; else used[d] = true;
      }
      if (ok) {
        ctx->copy_stream = made[roles[0] - '0'];
        ctx->dc_copy_stream = made[roles[1] - '0'];
        ctx->upload_stream = made[roles[2] - '0'];
        ctx->aux_stream = made[roles[3] - '0'];
        ctx->dc_pack_stream = made[roles[4] - '0'];
      }
    }
  }
  if (e == hipSuccess) e = hipEventCreate(&ctx->aux_done);
  for (auto& ev : ctx->ev)
    if (e == hipSuccess) e = hipEventCreate(&ev);
  for (auto& ev : ctx->stage_done)
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->dc_kernels_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->dc_elementwise_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&ctx->mail.p), sizeof(jxlt_context::HostMail), hipHostMallocDefault);
  if (e == hipSuccess) {
    ctx->mail.cap = 1;
    memset(ctx->mail.p, 0, sizeof(jxlt_context::HostMail));
    for (int k = 0; k < 2; k++) ctx->pack[k].h_launch_sec_end = ctx->mail.p->launch_sec_end[k];
    e = hipMalloc(reinterpret_cast<void**>(&ctx->deliver_counter.p), 128);  // (+ 64 bytes of look-back statistics)
  }
  if (e == hipSuccess) {
    ctx->deliver_counter.cap = 16;
    e = hipMemset(ctx->deliver_counter.p, 0, 128);
  }
  for (auto& ps : ctx->pack) {
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ps.finalized, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ps.plan_done, hipEventDisableTiming);
    for (auto& ev : ps.launch_done)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    g_create_error = std::string("context setup failed: ") + hipGetErrorString(e);
    jxlt_context_destroy(ctx);  // (handles partially built contexts: every handle is checked for null)
    return e == hipErrorOutOfMemory ? JXLT_ERR_OUT_OF_MEMORY : JXLT_ERR_NO_DEVICE;
  }
  // The copy pattern of a frame, and then some, once per device and process (CopyWarmup; JXLT_COPY_WARMUP=0: not at
  // all, 1 / 2: the weaker forms that were tried first).
  static const int copy_warmup = [] {
    const char* e2 = getenv("JXLT_COPY_WARMUP");
    return e2 ? atoi(e2) : 3;
  }();
  static std::atomic<bool> warmed[64];
  if (copy_warmup && device_ordinal >= 0 && device_ordinal < 64 && !warmed[device_ordinal].exchange(true)) CopyWarmup(ctx, copy_warmup);
  ctx->counted = true;
  DeviceBlockCache::Get().ContextCreated(ctx->device);
  *out = ctx;
  return JXLT_OK;
}

void jxlt_context_destroy(jxlt_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  // (everything the context has queued on any of ITS streams -- other contexts, lanes and frameworks on the device
  // are not waited for: its device buffers may be kept for the next context, DeviceBlockCache, and are then not
  // synchronised by a hipFree)
  for (hipStream_t st : {ctx->stream, ctx->aux_stream, ctx->dc_pack_stream, ctx->copy_stream, ctx->dc_copy_stream, ctx->upload_stream})
    if (st) (void)hipStreamSynchronize(st);
  FreeDevice(&ctx->own_payload);
  for (int c = 0; c < 3; c++) {
    FreeDevice(&ctx->own_planes[c]);
    FreeDevice(&ctx->quant_dc[c]);
    FreeDevice(&ctx->nzgrid[c]);
    FreeDevice(&ctx->dbg_xyb[c]);
    FreePinned(&ctx->h_quant_dc[c]);
  }
  FreeDevice(&ctx->raw_quant);
  FreeDevice(&ctx->strategy);
  FreeDevice(&ctx->blk_nz);
  FreeDevice(&ctx->blk_nscan);
  FreeDevice(&ctx->blk_nzmask);
  FreeDevice(&ctx->tokens);
  FreeDevice(&ctx->ytox);
  FreeDevice(&ctx->ytob);
  FreeDevice(&ctx->coef_scan);
  FreeDevice(&ctx->group_ntok);
  FreeDevice(&ctx->group_off);
  FreeDevice(&ctx->dbg_qf);
  FreeDevice(&ctx->dbg_mask);
  FreeDevice(&ctx->dbg_ent8);
  FreeDevice(&ctx->dbg_phase);
  FreeDevice(&ctx->hist);
  FreeDevice(&ctx->dc_records);
  FreeDevice(&ctx->dc_nac);
  FreeDevice(&ctx->dc_count);
  FreeDevice(&ctx->dc_rec_off);
  FreePinned(&ctx->h_hist);
  for (auto& ps : ctx->pack) {
    FreeDevice(&ps.code_table);
    FreeDevice(&ps.sec_bytes);
    FreeDevice(&ps.sec_byte_off);
    FreeDevice(&ps.sec_tiles);
    FreeDevice(&ps.tile_bits);
    FreeDevice(&ps.tile_base);
    FreeDevice(&ps.tile_info);
    FreeDevice(&ps.packed);
    FreeDevice(&ps.launch_sec_end);
    FreeDevice(&ps.tile_state);
    FreePinned(&ps.h_sec_byte_off);
    FreePinned(&ps.h_packed);
    FreePinned(&ps.h_code_table);
  }
  FreePinned(&ctx->h_raw_quant);
  FreePinned(&ctx->h_strategy);
  FreePinned(&ctx->h_tokens);
  FreePinned(&ctx->h_ytox);
  FreePinned(&ctx->h_ytob);
  FreePinned(&ctx->h_group_off);
  if (ctx->d_tab) (void)hipFree(ctx->d_tab);
  for (auto& st : ctx->stage) FreePinned(&st);
  FreePinned(&ctx->h_output);
  for (auto& ev : ctx->stage_done)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ps : ctx->pack) {
    if (ps.finalized) (void)hipEventDestroy(ps.finalized);
    if (ps.plan_done) (void)hipEventDestroy(ps.plan_done);
    for (auto& ev : ps.launch_done)
      if (ev) (void)hipEventDestroy(ev);
  }
  FreePinned(&ctx->mail);
  if (ctx->deliver_counter.p) (void)hipFree(ctx->deliver_counter.p);
  ctx->deliver_counter.p = nullptr;
  if (ctx->dc_kernels_done) (void)hipEventDestroy(ctx->dc_kernels_done);
  if (ctx->dc_elementwise_done) (void)hipEventDestroy(ctx->dc_elementwise_done);
  FreeDevice(&ctx->lut_overflow);
  FreeDevice(&ctx->overflow_tiles);
  FreeDevice(&ctx->dc_chain_summary);
  FreePinned(&ctx->h_lut_overflow);
  for (hipEvent_t ev : ctx->slab_ready)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->tile_done)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->aux_done) (void)hipEventDestroy(ctx->aux_done);
  if (ctx->aux_stream) {
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamDestroy(ctx->aux_stream);
  }
  if (ctx->upload_stream) {
    (void)hipStreamSynchronize(ctx->upload_stream);
    (void)hipStreamDestroy(ctx->upload_stream);
  }
  if (ctx->dc_pack_stream) (void)hipStreamDestroy(ctx->dc_pack_stream);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  if (ctx->dc_copy_stream) (void)hipStreamDestroy(ctx->dc_copy_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  const bool counted = ctx->counted;
  const int device = ctx->device;
  delete ctx;
  if (counted) DeviceBlockCache::Get().ContextDestroyed(device);
}

const char* jxlt_last_error(const jxlt_context* ctx) {
  return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

int jxlt_context_device(const jxlt_context* ctx) { return ctx ? ctx->device : -1; }

int jxlt_device_count(void) {
  int count = 0;
  return hipGetDeviceCount(&count) == hipSuccess && count > 0 ? count : 0;
}

// The CPUs next to a device: /sys/bus/pci/devices/<bus id>/local_cpulist ("0-63,128-191").
int jxlt_bind_thread_near_device(int device_ordinal) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device_ordinal) != hipSuccess) return JXLT_ERR_NO_DEVICE;
  for (char* p = bus; *p; ++p) *p = (char)tolower((unsigned char)*p);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
  FILE* f = fopen(path, "r");
  if (!f) return JXLT_ERR_UNSUPPORTED;
  char list[4096] = {0};
  const bool got = fgets(list, sizeof(list), f) != nullptr;
  fclose(f);
  if (!got) return JXLT_ERR_UNSUPPORTED;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n = 0;
  for (const char* p = list; *p && *p != '\n';) {
    char* end = nullptr;
    const long lo = strtol(p, &end, 10);
    if (end == p) break;
    long hi = lo;
    p = end;
    if (*p == '-') {
      hi = strtol(p + 1, &end, 10);
      p = end;
    }
    for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) {
      CPU_SET((int)c, &set);
      ++n;
    }
    if (*p == ',') ++p;
  }
  if (n == 0) return JXLT_ERR_UNSUPPORTED;
  return sched_setaffinity(0, sizeof(set), &set) == 0 ? JXLT_OK : JXLT_ERR_UNSUPPORTED;
}

}  // extern "C"

namespace {
// Staging of pageable host memory through the context's two page-locked buffers: `nthreads` host
// threads (the caller is one of them) live for the whole upload and fill band after band --
// fill(band, t, nthreads, stage) copies thread t's share -- while the previous band is in flight;
// issue(band, stage) enqueues the band's host-to-device copy.  (Spawning threads per band cost
// more than the copies of a 32 MB band.)
// Host threads per staged upload (JXLT_STAGE_THREADS overrides; capped by the machine).
int StageThreads() {
  static const int n = [] {
    const char* e = getenv("JXLT_STAGE_THREADS");
    int v = e ? atoi(e) : 8;
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && v > hw) v = hw;
    return v < 1 ? 1 : v > 64 ? 64 : v;
  }();
  return n;
}

template <typename Fill, typename Issue>
int StagedUpload(jxlt_context* ctx, size_t nbands, int nthreads, const Fill& fill, const Issue& issue) {
  std::atomic<size_t> released(0), finished(0);
  std::atomic<bool> aborted(false);
  auto worker = [&](int t) {
    for (size_t b = 0; b < nbands; b++) {
      while (released.load(std::memory_order_acquire) <= b) {
        if (aborted.load(std::memory_order_relaxed)) return;
        std::this_thread::yield();
      }
      if (aborted.load(std::memory_order_relaxed)) return;
      fill(b, t, nthreads, ctx->stage[b & 1].p);
      finished.fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthreads; t++) pool.emplace_back(worker, t);
  int rc = JXLT_OK;
  for (size_t b = 0; b < nbands && rc == JXLT_OK; b++) {
    uint8_t* stage = ctx->stage[b & 1].p;
    if (hipEventSynchronize(ctx->stage_done[b & 1]) != hipSuccess) {  // previous use of this buffer
      rc = JXLT_ERR_NO_DEVICE;
      break;
    }
    released.store(b + 1, std::memory_order_release);
    fill(b, 0, nthreads, stage);
    while (finished.load(std::memory_order_acquire) < (b + 1) * (size_t)(nthreads - 1)) std::this_thread::yield();
    rc = issue(b, stage);
    if (rc == JXLT_OK && hipEventRecord(ctx->stage_done[b & 1], ctx->stream) != hipSuccess) rc = JXLT_ERR_NO_DEVICE;
  }
  if (rc != JXLT_OK) {
    aborted.store(true);
    released.store(nbands);
    ctx->error = "staged upload failed";
  }
  for (auto& th : pool) th.join();
  return rc;
}
}  // namespace

extern "C" {

int jxlt_image_upload(jxlt_context* ctx, const float* const planes[3], size_t pitch_bytes,
                      size_t xsize, size_t ysize) {
  int rc = CheckImageArgs(ctx, reinterpret_cast<const void* const*>(planes), pitch_bytes, xsize, ysize);
  if (rc != JXLT_OK) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t pitch_floats = (xsize + 63) & ~size_t(63);
  const size_t row_bytes = xsize * sizeof(float);
  // Pinned / registered host memory goes straight over PCIe.  Pageable memory is staged
  // through two pinned buffers: host threads copy a band of rows while the previous band
  // is in flight (a pageable hipMemcpy2D is synchronous and runs at a few GB/s).
  hipPointerAttribute_t attr;
  const bool pinned = hipPointerGetAttributes(&attr, planes[0]) == hipSuccess &&
                      attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();  // a pageable pointer is not an error
  constexpr size_t kStageBytes = size_t(32) << 20;
  const size_t band_rows = std::max<size_t>(1, std::min(ysize, kStageBytes / row_bytes));
  if (!pinned)
    for (auto& st : ctx->stage)
      if ((rc = EnsurePinned(ctx, &st, band_rows * row_bytes)) != JXLT_OK) return rc;
  for (int c = 0; c < 3; c++) {
    rc = EnsureDevice(ctx, &ctx->own_planes[c], pitch_floats * ysize);
    if (rc != JXLT_OK) return rc;
    ctx->planes[c] = ctx->own_planes[c].p;
    if (pinned)
      HIP_TRY(ctx, hipMemcpy2DAsync(ctx->own_planes[c].p, pitch_floats * sizeof(float), planes[c], pitch_bytes,
                                    row_bytes, ysize, hipMemcpyHostToDevice, ctx->stream));
  }
  if (!pinned) {
    // bands of all three planes in one staged sequence
    const size_t bands_per_plane = (ysize + band_rows - 1) / band_rows;
    const int nthreads = ysize * row_bytes > (size_t(4) << 20) ? StageThreads() : 1;
    rc = StagedUpload(
        ctx, 3 * bands_per_plane, nthreads,
        [&](size_t band, int t, int nt, uint8_t* stage) {
          const size_t c = band / bands_per_plane, y0 = (band % bands_per_plane) * band_rows;
          const size_t rows = std::min(band_rows, ysize - y0);
          const uint8_t* src = reinterpret_cast<const uint8_t*>(planes[c]) + y0 * pitch_bytes;
          for (size_t y = rows * t / nt; y < rows * (t + 1) / nt; y++)
            memcpy(stage + y * row_bytes, src + y * pitch_bytes, row_bytes);
        },
        [&](size_t band, uint8_t* stage) {
          const size_t c = band / bands_per_plane, y0 = (band % bands_per_plane) * band_rows;
          const size_t rows = std::min(band_rows, ysize - y0);
          return hipMemcpy2DAsync(ctx->own_planes[c].p + y0 * pitch_floats, pitch_floats * sizeof(float), stage,
                                  row_bytes, row_bytes, rows, hipMemcpyHostToDevice, ctx->stream) == hipSuccess
                     ? JXLT_OK
                     : JXLT_ERR_NO_DEVICE;
        });
    if (rc != JXLT_OK) return rc;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // caller may reuse its buffers
  ctx->host_src_kind = 0;
  ctx->pitch_floats = (ptrdiff_t)pitch_floats;
  ctx->pix_stride = 1;
  ctx->byteswap = 0;
  ctx->xsize = xsize;
  ctx->ysize = ysize;
  ctx->encoded = false;
  return JXLT_OK;
}

void* jxlt_pinned_alloc(size_t bytes) {
  void* p = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return nullptr;
  }
  // portable: every device of the process may DMA from / to it (frames and outputs shared by several GPUs)
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}

int jxlt_pinned_register(void* p, size_t bytes) {
  if (!p || !bytes) return JXLT_ERR_INVALID_ARGUMENT;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    return JXLT_ERR_NO_DEVICE;
  }
  if (hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {
    (void)hipGetLastError();
    return JXLT_ERR_NO_DEVICE;
  }
  return JXLT_OK;
}

void jxlt_pinned_unregister(void* p) {
  if (p) (void)hipHostUnregister(p);
}

void jxlt_pinned_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int jxlt_image_set_device(jxlt_context* ctx, const void* const device_planes[3], size_t pitch_bytes,
                          size_t xsize, size_t ysize) {
  int rc = CheckImageArgs(ctx, device_planes, pitch_bytes, xsize, ysize);
  if (rc != JXLT_OK) return rc;
  for (int c = 0; c < 3; c++) ctx->planes[c] = static_cast<const float*>(device_planes[c]);
  ctx->host_src_kind = 0;
  ctx->pitch_floats = (ptrdiff_t)(pitch_bytes / sizeof(float));
  ctx->pix_stride = 1;
  ctx->byteswap = 0;
  ctx->xsize = xsize;
  ctx->ysize = ysize;
  ctx->encoded = false;
  return JXLT_OK;
}

namespace {
// The frame is the sample payload of a PFM file at `payload` (device memory): interleaved RGB
// f32, bottom row first, byte-reversed if big endian (read_pfm.cc:199-209).  tile_kernel reads
// it in place: no de-interleaving pass anywhere.
int SetPfmView(jxlt_context* ctx, const float* payload, size_t xsize, size_t ysize, int big_endian) {
  ctx->host_src_kind = 0;
  for (int c = 0; c < 3; c++) ctx->planes[c] = payload + (ysize - 1) * xsize * 3 + c;
  ctx->pitch_floats = -(ptrdiff_t)(xsize * 3);
  ctx->pix_stride = 3;
  ctx->byteswap = big_endian ? 1 : 0;
  ctx->xsize = xsize;
  ctx->ysize = ysize;
  ctx->encoded = false;
  return JXLT_OK;
}
int CheckPfmArgs(jxlt_context* ctx, const void* payload, size_t xsize, size_t ysize) {
  if (!ctx) return JXLT_ERR_INVALID_ARGUMENT;
  const void* const three[3] = {payload, payload, payload};
  return CheckImageArgs(ctx, three, xsize * 3 * sizeof(float), xsize, ysize);
}
}  // namespace

int jxlt_image_set_device_pfm(jxlt_context* ctx, const void* device_payload, size_t xsize, size_t ysize,
                              int big_endian) {
  const int rc = CheckPfmArgs(ctx, device_payload, xsize, ysize);
  if (rc != JXLT_OK) return rc;
  return SetPfmView(ctx, static_cast<const float*>(device_payload), xsize, ysize, big_endian);
}

int jxlt_image_upload_pfm(jxlt_context* ctx, const void* host_payload, size_t xsize, size_t ysize,
                          int big_endian) {
  int rc = CheckPfmArgs(ctx, host_payload, xsize, ysize);
  if (rc != JXLT_OK) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t nfloats = xsize * ysize * 3;
  if ((rc = EnsureDevice(ctx, &ctx->own_payload, nfloats)) != JXLT_OK) return rc;
  // Page-locked memory goes over PCIe in one piece.  Pageable memory (e.g. the mmap of the
  // file) is staged through the two pinned buffers: host threads fill one while the other is
  // in flight, so the file's pages are touched once, by several cores, overlapped with the DMA.
  const size_t nbytes = nfloats * sizeof(float);
  const uint8_t* src = static_cast<const uint8_t*>(host_payload);
  uint8_t* dst = reinterpret_cast<uint8_t*>(ctx->own_payload.p);
  hipPointerAttribute_t attr;
  const bool pinned = hipPointerGetAttributes(&attr, host_payload) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();  // a pageable pointer is not an error
  if (pinned) {
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, ctx->stream));
  } else {
    constexpr size_t kStageBytes = size_t(32) << 20;
    for (auto& st : ctx->stage)
      if ((rc = EnsurePinned(ctx, &st, std::min(kStageBytes, nbytes))) != JXLT_OK) return rc;
    const size_t nbands = (nbytes + kStageBytes - 1) / kStageBytes;
    rc = StagedUpload(
        ctx, nbands, nbytes > (size_t(4) << 20) ? StageThreads() : 1,
        [&](size_t band, int t, int nt, uint8_t* stage) {
          const size_t o = band * kStageBytes, n = std::min(kStageBytes, nbytes - o);
          memcpy(stage + n * t / nt, src + o + n * t / nt, n * (t + 1) / nt - n * t / nt);
        },
        [&](size_t band, uint8_t* stage) {
          const size_t o = band * kStageBytes, n = std::min(kStageBytes, nbytes - o);
          return hipMemcpyAsync(dst + o, stage, n, hipMemcpyHostToDevice, ctx->stream) == hipSuccess ? JXLT_OK
                                                                                                     : JXLT_ERR_NO_DEVICE;
        });
    if (rc != JXLT_OK) return rc;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // caller may reuse its buffer
  return SetPfmView(ctx, ctx->own_payload.p, xsize, ysize, big_endian);
}

namespace {
bool IsPageLocked(const void* p) {
  hipPointerAttribute_t attr;
  const bool pinned = hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost;
  (void)hipGetLastError();  // a pageable pointer is not an error
  return pinned;
}
}  // namespace

int jxlt_image_attach_host(jxlt_context* ctx, const float* const planes[3], size_t pitch_bytes, size_t xsize,
                           size_t ysize) {
  int rc = CheckImageArgs(ctx, reinterpret_cast<const void* const*>(planes), pitch_bytes, xsize, ysize);
  if (rc != JXLT_OK) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!IsPageLocked(planes[0]) || !IsPageLocked(planes[1]) || !IsPageLocked(planes[2])) {
    ctx->error = "jxlt_image_attach_host needs page-locked memory (jxlt_pinned_alloc / jxlt_pinned_register)";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  const size_t pitch_floats = (xsize + 63) & ~size_t(63);
  for (int c = 0; c < 3; c++) {
    if ((rc = EnsureDevice(ctx, &ctx->own_planes[c], pitch_floats * ysize)) != JXLT_OK) return rc;
    ctx->planes[c] = ctx->own_planes[c].p;
    ctx->host_src[c] = reinterpret_cast<const uint8_t*>(planes[c]);
  }
  ctx->host_pitch_bytes = pitch_bytes;
  ctx->pitch_floats = (ptrdiff_t)pitch_floats;
  ctx->pix_stride = 1;
  ctx->byteswap = 0;
  ctx->xsize = xsize;
  ctx->ysize = ysize;
  ctx->encoded = false;
  ctx->host_src_kind = 1;
  return JXLT_OK;
}

int jxlt_image_attach_host_pfm(jxlt_context* ctx, const void* host_payload, size_t xsize, size_t ysize,
                               int big_endian) {
  int rc = CheckPfmArgs(ctx, host_payload, xsize, ysize);
  if (rc != JXLT_OK) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!IsPageLocked(host_payload)) {
    ctx->error = "jxlt_image_attach_host_pfm needs page-locked memory (jxlt_pinned_alloc / jxlt_pinned_register)";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  if ((rc = EnsureDevice(ctx, &ctx->own_payload, xsize * ysize * 3)) != JXLT_OK) return rc;
  rc = SetPfmView(ctx, ctx->own_payload.p, xsize, ysize, big_endian);
  ctx->host_src[0] = static_cast<const uint8_t*>(host_payload);
  ctx->host_src_kind = 2;
  return rc;
}

int jxlt_image_size(const jxlt_context* ctx, size_t* xsize, size_t* ysize) {
  if (!ctx || !xsize || !ysize || !ctx->planes[0]) return JXLT_ERR_INVALID_ARGUMENT;
  *xsize = ctx->xsize;
  *ysize = ctx->ysize;
  return JXLT_OK;
}

namespace {
const char kUnsupportedValues[] =
    "the frame has values the codestream cannot carry (a quantised coefficient beyond 16 bits or a DC value beyond "
    "int16: samples around 1e38 or infinities)";
int EnqueuePlan(jxlt_context* ctx, int kind, uint64_t rec_bound, hipStream_t stream);  // (below)
int WaitSizes(jxlt_context* ctx, int kind);  // (below)

// (diagnostics, JXLT_TRACE_EVENTS=1: a timed event on `stream`, listed against the encode's first event by jxlt_synchronize)
bool TraceEventsOn() {
  static const bool on = [] {
    const char* e = getenv("JXLT_TRACE_EVENTS");
    return e && atoi(e) != 0;
  }();
  return on;
}
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
int IssueDeferred(jxlt_context* ctx, bool wait);  // (below)
int WaitWord(jxlt_context* ctx, const uint32_t* word, uint32_t want, hipStream_t stream, const char* what) {
  const volatile uint32_t* w = word;
  if (*w == want) return JXLT_OK;
  const auto t0 = std::chrono::steady_clock::now();
  auto next_check = t0 + std::chrono::milliseconds(2);
  for (;;) {
    for (int spin = 0; spin < 256; spin++) {
      if (*w == want) return JXLT_OK;
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (ctx->deferred_dc.pending && !ctx->in_deferred) {
      const int rcd = IssueDeferred(ctx, /*wait=*/false);
      if (rcd != JXLT_OK) return rcd;
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
      if (*w == want) return JXLT_OK;
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
    const int rc = WaitWord(ctx, &ctx->mail.p->delivered_seq[kind][0], ctx->deliver_seq[kind],
                            kind ? ctx->copy_stream : ctx->dc_copy_stream, "section hand-over");
    if (rc != JXLT_OK) return rc;
  }
  return JXLT_OK;
}

// publish_kernel on `stream`: up to kPublishSegments (device source, host destination, dwords) pairs, an optional
// 64-bit word, then `seq` to the host word `flag`.
struct PublishSeg {
  const void* src;
  void* dst;
  size_t words;
};
int EnqueuePublish(jxlt_context* ctx, hipStream_t stream, const PublishSeg* segs, int nsegs, const unsigned long long* src64,
                   unsigned long long* dst64, uint32_t* flag, uint32_t seq) {
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
    ENSURE(dbg_phase, 16);
    HIP_TRY(ctx, hipMemsetAsync(ctx->dbg_phase.p, 0, 16 * sizeof(unsigned long long), ctx->stream));
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
  // 5-9 us apart, in front of every frame's first tile_kernel launch)
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
  HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
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
    // (experiment knob, tools/: JXLT_TILE_EXTRA_LDS=<bytes> of unused dynamic LDS per workgroup lowers the number
    // of resident workgroups per CU -- how much of tile_kernel's speed comes from the second one?)
    static const unsigned extra_lds = [] {
      const char* e = getenv("JXLT_TILE_EXTRA_LDS");
      return e ? (unsigned)atoi(e) : 0u;
    }();
    // Behind every launch: the tiles it filed because a quantised magnitude did not fit the root table of its
    // entropy estimates, again with computed roots (enc_ac_strategy.cc:118-126 takes a Sqrt per coefficient) -- a
    // small fixed grid whose workgroups usually find an empty list and leave.
    const unsigned redo_grid = std::min<unsigned>(slab_tiles, kRedoGrid);
    if (debug || profile)
      hipLaunchKernelGGL(tile12_kernel_debug, dim3(slab_tiles), dim3(kTile12Threads), extra_lds, ctx->stream, S);
    else
      hipLaunchKernelGGL(tile12_kernel, dim3(slab_tiles), dim3(kTile12Threads), extra_lds, ctx->stream, S);
    hipLaunchKernelGGL(tile12_kernel_redo, dim3(redo_grid), dim3(kTile12Threads), 0, ctx->stream, S);
    // (the counts of redone tiles leave with the DC histogram, below)
    if (sl + 1 == nslabs) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->tile_done[sl], ctx->stream));
    if (!pieces[sl].ends_dc_row) continue;
    // ---- tokenisation of the row(s) of DC groups this piece completes, on the aux stream
    if (tok_stream != ctx->stream) HIP_TRY(ctx, hipStreamWaitEvent(tok_stream, ctx->tile_done[sl], 0));
    // DC groups first: their histogram leaves for the host as soon as the last row's is complete, so that the DC
    // code is built while token_kernel is still running
    const size_t dc_row0 = dc_rows_done, dc_row1 = nslabs == 1 ? ndc_rows : dc_row0 + 1;
    dc_rows_done = dc_row1;
    const size_t slab_dc = (dc_row1 - dc_row0) * xdc;  // DC groups of this launch
    D.dcg_first = (int)(dc_row0 * xdc);
    // (a resident frame: the element-wise kernel and the two chain kernels do not depend on each other and none of
    // them fills the chip -- side by side on two streams; experiment knob JXLT_DC_SPLIT=0: one after the other)
    static const bool dc_split = [] {
      const char* e = getenv("JXLT_DC_SPLIT");
      return !e || atoi(e) != 0;
    }();
    const bool split = dc_split && nslabs == 1;
    ctx->dc_elementwise_split = split;
    const hipStream_t elem_stream = split ? ctx->aux_stream : tok_stream;
    if (split) HIP_TRY(ctx, hipStreamWaitEvent(elem_stream, ctx->tile_done[sl], 0));
    hipLaunchKernelGGL(dc_elementwise_kernel, dim3((unsigned)(slab_dc * kDcParts)), dim3(256), 0, elem_stream, D);
    if (split) HIP_TRY(ctx, hipEventRecord(ctx->dc_elementwise_done, elem_stream));
    // A resident frame of up to 1024 groups (8192^2): the two chain kernels and the histogram's publication on a
    // stream of their own, submitted in front of token_kernel but not waited for by it -- token_kernel is short there
    // (0.06-0.19 ms) and starts 0.03 ms earlier (4096^2: 0.65-0.69 -> 0.64 ms, 8192^2: 1.59 -> 1.55-1.57).  Larger
    // frames keep the DC-group kernels in FRONT of token_kernel: beside it they do not get a CU before its
    // workgroups retire, the DC histogram arrives with the AC histogram (16384^2: at 4.52 instead of 4.09 ms) and the
    // DC code is built behind token_kernel instead of under it (5.36 against 5.18 ms).
    // (experiment knob: JXLT_DC_BESIDE_TOKEN=0 / 1 forces either)
    static const int dc_beside = [] {
      const char* e = getenv("JXLT_DC_BESIDE_TOKEN");
      return e ? atoi(e) : -1;
    }();
    const bool beside = nslabs == 1 && split && (dc_beside >= 0 ? dc_beside != 0 : ngroups <= 1024);
    const hipStream_t chain_stream = beside ? ctx->upload_stream : tok_stream;
    if (beside) HIP_TRY(ctx, hipStreamWaitEvent(chain_stream, ctx->tile_done[sl], 0));
    hipLaunchKernelGGL(dc_chain_summary_kernel, dim3((unsigned)(slab_dc * kDcChainChunks)), dim3(kDcChainThreads), 0,
                       chain_stream, D);
    hipLaunchKernelGGL(dc_chain_kernel, dim3((unsigned)(slab_dc * kDcChainChunks)), dim3(kDcChainThreads), 0,
                       chain_stream, D);
    if (beside) {
      HIP_TRY(ctx, hipStreamWaitEvent(chain_stream, ctx->dc_elementwise_done, 0));
      const PublishSeg segs[2] = {{ctx->hist.p + 64 * 64, ctx->h_hist.p + 64 * 64, 64 * 64},
                                  {ctx->lut_overflow.p, ctx->h_lut_overflow.p, nslabs + 1}};
      const int rcp = EnqueuePublish(ctx, chain_stream, segs, 2, nullptr, nullptr, &ctx->mail.p->dc_hist_seq, frame_seq);
      if (rcp != JXLT_OK) return rcp;
      HIP_TRY(ctx, hipEventRecord(ctx->dc_kernels_done, chain_stream));
    } else if (sl + 1 == nslabs) {
      // The DC histogram (and the counts of the tiles redone with computed roots) leaves IN FRONT of token_kernel: one
      // small kernel stores both to the host's page-locked memory and then the frame's sequence number to the word
      // the host polls (~6 us on the stream).  Beside token_kernel -- on the copy stream, where rounds 2-3 had the
      // download -- the kernel does not get a wave slot before token_kernel's workgroups begin to retire: the
      // histogram arrived 0.4 ms late and the DC code was built behind the AC code (round 4, first version).
      HIP_TRY(ctx, hipEventRecord(ctx->dc_kernels_done, tok_stream));
      if (split) HIP_TRY(ctx, hipStreamWaitEvent(tok_stream, ctx->dc_elementwise_done, 0));
      const PublishSeg segs[2] = {{ctx->hist.p + 64 * 64, ctx->h_hist.p + 64 * 64, 64 * 64},
                                  {ctx->lut_overflow.p, ctx->h_lut_overflow.p, nslabs + 1}};
      const int rcp = EnqueuePublish(ctx, tok_stream, segs, 2, nullptr, nullptr, &ctx->mail.p->dc_hist_seq, frame_seq);
      if (rcp != JXLT_OK) return rcp;
    }
    const size_t ty0 = dc_row0 * 2048, ty1 = std::min(ctx->ysize, dc_row1 * 2048);  // pixel rows being tokenised
    const size_t g0 = (ty0 / 256) * (size_t)g.xsize_groups;
    const size_t ng = ((ty1 - ty0 + 255) / 256) * (size_t)g.xsize_groups;
    // (every token_kernel workgroup finds its group's token offset itself: the counts of all groups before it,
    // whichever launch tokenised them, are final by now)
    K.group_first = (int)g0;
    // (experiment knob, tools/: JXLT_TOKEN_EXTRA_LDS=<bytes> of unused dynamic LDS per workgroup lowers the
    // number of resident workgroups per CU)
    static const unsigned token_extra_lds = [] {
      const char* e = getenv("JXLT_TOKEN_EXTRA_LDS");
      return e ? (unsigned)atoi(e) : 0u;
    }();
    if (nblocks > kTokenNarrowBlocks)
      hipLaunchKernelGGL(token_kernel_wide, dim3((unsigned)ng), dim3(kTokenThreads), token_extra_lds, tok_stream, K);
    else
      hipLaunchKernelGGL(token_kernel, dim3((unsigned)ng), dim3(kTokenThreads), token_extra_lds, tok_stream, K);
    // (whatever follows on the main stream -- the sections' packing -- reads what the DC-group kernels wrote)
    if (beside) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->dc_kernels_done, 0));
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->host_src_kind = 0;  // the frame is resident now (a redo with exact roots must not fetch it again)
  // AC histogram + total token count leave right behind the last token_kernel (publish_kernel: no copy command, no
  // event -- the host polls the sequence word)
  HIP_TRY(ctx, hipEventRecord(ctx->aux_done, tok_stream));
  TraceMark(ctx, "token_kernel done", tok_stream);
  {
    const PublishSeg seg = {ctx->hist.p, ctx->h_hist.p, 64 * 64};
    const int rcp = EnqueuePublish(ctx, tok_stream, &seg, 1, reinterpret_cast<const unsigned long long*>(ctx->group_off.p + ngroups),
                                   &ctx->mail.p->token_total, &ctx->mail.p->ac_hist_seq, frame_seq);
    if (rcp != JXLT_OK) return rcp;
  }
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
    if (tok_stream == ctx->stream) {
      plan_stream = ctx->aux_stream;
      // (the DC-group sections' plan first, behind the DC-group kernels and beside token_kernel: for a small frame
      // those sections' packing is what the frame waits for last, and the plan is a third of its launches)
      static const bool dc_plan_early = [] {
        const char* e = getenv("JXLT_DC_PLAN_EARLY");  // (experiment knob)
        return !e || atoi(e) != 0;
      }();
      if (dc_plan_early) {
        HIP_TRY(ctx, hipStreamWaitEvent(plan_stream, ctx->dc_kernels_done, 0));
        const int rcd = EnqueuePlan(ctx, 0, ctx->dc_records.cap / 3, plan_stream);
        if (rcd != JXLT_OK) return rcd;
      }
      HIP_TRY(ctx, hipStreamWaitEvent(plan_stream, ctx->aux_done, 0));
    }
    const int rcp = EnqueuePlan(ctx, 1, ctx->tokens.cap / 3, plan_stream);
    if (rcp != JXLT_OK) return rcp;
  }
  ctx->encoded = true;
  ctx->enqueued_at = std::chrono::steady_clock::now();
  {
    uint32_t dbits, sbits;
    memcpy(&dbits, &params->distance, 4);
    memcpy(&sbits, &params->scale, 4);
    ctx->wait_key = ((uint64_t)ctx->xsize * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)ctx->ysize << 21) ^ ((uint64_t)dbits << 7) ^ sbits ^
                    ((uint64_t)params->flags << 50) ^ ((uint64_t)ctx->host_src_kind << 60) ^ 1u;
  }
  ctx->offsets_fetched = false;
  ctx->pack[0].measured_sections = ctx->pack[1].measured_sections = 0;
  ctx->pack[0].launches = ctx->pack[1].launches = 0;
  ctx->last_flags = params->flags;
  ctx->profiled = true;  // the five stage events are always recorded (a few microseconds per frame)
  ctx->last_params = *params;
  ctx->overflow_checked = false;
  ctx->encode_status = JXLT_OK;
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
    // (off unless JXLT_WAIT_SLEEP=1: with it the 16384^2 step is 5.19-5.20 against 5.15-5.19 ms, one run 5.59 -- a
    // sleeping thread depends on the host's scheduler for its wake-up --, 48 resident 3840x2160 frames over eight lanes
    // 3624 against 3743 frames per second; it saves 0.7 of a CPU per encoding thread, tools/wait_sleep_ab.sh)
    static const bool sleep_allowed = [] {
      const char* e = getenv("JXLT_WAIT_SLEEP");
      return e && atoi(e) != 0;
    }();
    const volatile uint32_t* w = &ctx->mail.p->dc_hist_seq;
    if (sleep_allowed && *w != ctx->seq && ctx->wait_key == ctx->last_wait_key && ctx->last_dc_wait_us > 600.0) {
      const auto wake = ctx->enqueued_at + std::chrono::microseconds((long long)(ctx->last_dc_wait_us - 300.0));
      // (in pieces: a frame that is done early -- a faster clock, a lighter load -- is noticed within 0.2 ms)
      while (*w != ctx->seq) {
        const auto now = std::chrono::steady_clock::now();
        if (now >= wake) break;
        std::this_thread::sleep_for(std::min<std::chrono::steady_clock::duration>(wake - now, std::chrono::microseconds(200)));
      }
    }
    const int rcw = WaitWord(ctx, &ctx->mail.p->dc_hist_seq, ctx->seq, ctx->stream, "device pipeline");
    if (rcw != JXLT_OK) return rcw;
    const double waited = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ctx->enqueued_at).count();
    // (the shortest of the recent frames: a frame that was held up must not make the next one oversleep)
    ctx->last_dc_wait_us = ctx->wait_key == ctx->last_wait_key && ctx->last_dc_wait_us > 0.0 ? std::min(waited, ctx->last_dc_wait_us * 1.02) : waited;
    ctx->last_wait_key = ctx->wait_key;
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
}  // namespace

int jxlt_encode_enqueue(jxlt_context* ctx, const jxlt_params* params) { return EnqueuePipeline(ctx, params); }

int jxlt_encode_stats(jxlt_context* ctx, jxlt_encode_stats_t* out) {
  if (!ctx || !out) return JXLT_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc0 = ResolveRootTableOverflow(ctx);
  if (rc0 != JXLT_OK) return rc0;
  out->tiles_redone_exact_roots = ctx->tiles_redone;
  out->encodes_with_redone_tiles = ctx->exact_reruns;
  out->tiles = (uint32_t)((size_t)ctx->geom.xsize_tiles * ctx->geom.ysize_tiles);
  return JXLT_OK;
}

int jxlt_set_strategy_distance(jxlt_context* ctx, float first_call_distance) {
  if (!ctx || !(first_call_distance >= 0.0f)) return JXLT_ERR_INVALID_ARGUMENT;
  ctx->strategy_distance = first_call_distance;
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
    if (!ctx->copies_pending) return JXLT_OK;
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
    const int rcw = WaitWord(ctx, &ctx->mail.p->ac_hist_seq, ctx->seq, ctx->stream, "tokenisation");
    if (rcw != JXLT_OK) return rcw;
  }
  ctx->h_group_off.p[ngroups] = ctx->mail.p->token_total;
  ctx->offsets_fetched = true;
  if (ac_histograms) *ac_histograms = ctx->h_hist.p;
  if (dc_histograms) *dc_histograms = ctx->h_hist.p + 64 * 64;
  return JXLT_OK;
}

namespace {
// One pass or two.  ONE (pack_tile_stream_kernel: every tile takes its bit position from the tiles in front of it
// while it runs; no measuring pass, none of its offsets / scan / finalize kernels, no second read of the records) up
// to 1024 groups and ~20 000 tiles, TWO (measure, lay out, write: rounds 1-3) above.  The single pass saves the chain of small
// kernels -- 2048^2: 0.41 -> 0.36 ms, 4096^2: 0.62 -> 0.58, 8192^2: 1.56 -> 1.54 -- but its kernel packs 55-85 tiles
// per us where the two-pass form's writing pass does 120 (one tile per workgroup, a ticket and the waits for the
// neighbours' sizes in front of every tile), and from ~20 000 tiles on that costs more than the measuring pass did
// (the AC sections of the 16384^2 bench frame, 25 700 tiles: 5.21-5.26 against 5.14-5.22 ms; 8192^2 of uniform noise,
// 34 700 tiles: 3.24 against 2.88; 8192^2 at d = 0.5, 15 400 tiles: 2.02 against 2.11; 4096^2 at d = 0.1, 9 800 tiles:
// 1.29 against 1.76 -- tools/ab_stream.sh, tools/token_heavy_ab.sh).  What counts is the number of tiles: the AC
// sections' record count is known when their packing is asked for.  JXLT_PACK_TWO_PASS=1 / 0 forces either; the kernel hand-over (JXLT_DELIVER_KERNEL=1)
// needs two passes.
int PackPassesForced() {  // 1 / 2, or 0: by size
  static const int forced = [] {
    const char* two = getenv("JXLT_PACK_TWO_PASS");
    const char* kern = getenv("JXLT_DELIVER_KERNEL");
    if (kern && atoi(kern) != 0) return 2;
    return two ? (atoi(two) != 0 ? 2 : 1) : 0;
  }();
  return forced;
}
// (may a single pass be asked for at all: the plans then prepare the tiles' states)
bool PackSinglePass(const jxlt_context*) { return PackPassesForced() != 2; }
bool PackSinglePassFor(const jxlt_context* ctx, int kind, uint64_t records) {
  if (PackPassesForced() != 0) return PackPassesForced() == 1;
  // (a section has at least one tile: a frame of 4096 groups is 4096 workgroups with a ticket, a code table and a
  // look-back each even when they hold a handful of records -- 16384^2 at d = 4, 1 800 tiles' worth of records: 4.55-4.58
  // ms in one pass, 4.52-4.53 in two; the DC-group sections of the 16384^2 frame alone in one pass: 5.26-5.28
  // against 5.24-5.26, tools/mixed_ab.sh)
  if ((size_t)ctx->geom.xsize_groups * ctx->geom.ysize_groups > 1024) return false;
  return kind == 0 || records <= (80ull << 20);
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
  P.sized_count = ps.launch_sec_end.p ? ps.launch_sec_end.p + 2 * kPackMaxLaunches : nullptr;
  P.tile_state = PackSinglePass(ctx) ? ps.tile_state.p : nullptr;
  P.block_state = PackSinglePass(ctx) && ps.tile_state.p ? ps.tile_state.p + ps.state_tiles : nullptr;
  static const bool stats_on = [] {
    const char* e = getenv("JXLT_LOOKBACK_STATS");  // (with JXLT_TRACE_EVENTS; the counting slows the pass down)
    return e && atoi(e) != 0;
  }();
  P.lookback_stats = stats_on && TraceEventsOn() && ctx->deliver_counter.p ? ctx->deliver_counter.p + 16 + kind * 4 : nullptr;
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
int EnqueuePlan(jxlt_context* ctx, int kind, uint64_t rec_bound, hipStream_t stream) {
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
  ENSURE(launch_sec_end, 3 * kPackMaxLaunches);  // (+ the single pass's tickets and size counters)
  if (PackSinglePass(ctx)) {  // (tile states, block states behind them)
    ENSURE(tile_state, max_tiles + max_tiles / kPackBlockTiles + 2);
    ps.state_tiles = max_tiles;
  }
#undef ENSURE
  const PackTileArgs P = TileArgsOf(ctx, kind, nsec);
  const unsigned sec_blocks = (unsigned)((nsec + 255) / 256);
  hipLaunchKernelGGL(pack_tile_count_kernel, dim3(sec_blocks), dim3(256), 0, stream, P);
  hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, (const uint32_t*)ps.sec_tiles.p,
                     ps.tile_base.p, (int)nsec);
  hipLaunchKernelGGL(pack_tile_plan_kernel, dim3(sec_blocks), dim3(256), 0, stream, P);
  HIP_TRY(ctx, hipGetLastError());
  // (which sections a launch of the writing pass completes follows from the plan and is worked out on the device --
  // pack_tile_finalize_kernel --: the host does not fetch the plan any more)
  ps.planned = true;
  ps.plan_elsewhere = stream != ps.stream;
  if (ps.plan_elsewhere) HIP_TRY(ctx, hipEventRecord(ps.plan_done, stream));
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
  // (experiment knobs, tools/: JXLT_PACK_LAUNCHES=<n>, JXLT_PACK_GROWTH=<percent, each share against the one before>)
  static const int ac_launches = [] {
    const char* e = getenv("JXLT_PACK_LAUNCHES");
    return e ? std::max(1, std::min(atoi(e), (int)jxlt_context::PackSet::kMaxLaunches)) : 3;
  }();
  static const double growth = [] {
    const char* e = getenv("JXLT_PACK_GROWTH");
    return e ? std::max(1.0, atoi(e) / 100.0) : 2.0;
  }();
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
  {
    const PublishSeg seg = {ps.h_code_table.p, ps.code_table.p, 64 * 64};
    if ((rc = EnqueuePublish(ctx, ps.stream, &seg, 1, nullptr, nullptr, nullptr, 0)) != JXLT_OK) return rc;
  }
  const uint64_t blob_bound = rec_bound * 4 + nsec * 8 + 64;
  if (ps.packed.cap < blob_bound && (rc = EnsureDevice(ctx, &ps.packed, blob_bound + blob_bound / 8)) != JXLT_OK)
    return rc;
  // (the tiles OR their first and last dwords into the blob: zero up to where the sections can reach with this code)
  ps.zeroed_bytes = std::min<uint64_t>(SectionBytesBound(ctx, kind, code_table, nsec), ps.packed.cap);
  HIP_TRY(ctx, hipMemsetAsync(ps.packed.p, 0, ps.zeroed_bytes, ps.stream));
  // ONE launch is the default here: the frames the single pass is used for (up to 1024 groups, 6 MB of AC sections)
  // are packed in 0.02-0.1 ms, and every further launch costs a publish kernel, a copy command and a ramp -- 2048^2:
  // 0.355 / 0.377 / 0.382 ms with one / two / three launches, 4096^2: 0.578 / 0.591 / 0.613, 8192^2: 1.562 / 1.568 /
  // 1.596, 48 resident 3840x2160 frames over six lanes: 3423 / 3370 / 3284 frames per second (tools/launches_small.sh).
  static const int ac_launches = [] {
    const char* e = getenv("JXLT_PACK_LAUNCHES");
    return e ? std::max(1, std::min(atoi(e), (int)jxlt_context::PackSet::kMaxLaunches)) : 1;
  }();
  static const double growth = [] {
    const char* e = getenv("JXLT_PACK_GROWTH");
    return e ? std::max(1.0, atoi(e) / 100.0) : 2.0;
  }();
  const int want = kind == 0 ? 1 : ac_launches;
  ps.launches = (int)std::min<size_t>((size_t)want, std::max<size_t>(1, max_tiles / 64));
  for (int i = 0; i <= ps.launches; i++) {
    const double share = growth > 1.0 ? (std::pow(growth, i) - 1.0) / (std::pow(growth, ps.launches) - 1.0)
                                      : (double)i / ps.launches;
    ps.launch_t0[i] = i == ps.launches ? (uint32_t)max_tiles : (uint32_t)((double)max_tiles * share);
  }
  ps.pack_seq++;
  // (experiment knob JXLT_PACK_SIZES_IN_KERNEL=1, one launch only: the tile that is the last to say its size stores the
  // sections' bit counts to the host's mirror and sets the launch's word itself, while the launch is still packing --
  // no publish kernel behind the launch, the copy command queued behind the launch's event before the launch ends.
  // Measured and not kept: the counter every tile draws from needs acquire + release at agent scope, i.e. an L2
  // write-back and invalidate per tile -- 8192^2 1.66-1.67 against 1.55-1.58 ms, 1024^2 ... 4096^2 and the 4K
  // batch inside the noise, tools/sizes_in_kernel_ab.sh)
  static const bool sizes_in_kernel = [] {
    const char* e = getenv("JXLT_PACK_SIZES_IN_KERNEL");
    return e && atoi(e) != 0;
  }();
  const bool in_kernel_sizes = sizes_in_kernel && ps.launches == 1;
  for (int i = 0; i < ps.launches; i++) {
    PackTileArgs W = TileArgsOf(ctx, kind, nsec);
    W.tile_first = ps.launch_t0[i];
    W.tile_end = ps.launch_t0[i + 1];
    W.launch_index = (uint32_t)i;
    if (in_kernel_sizes) {
      W.host_sec_bits = ps.h_sec_bits(nsec);
      W.host_flag = &ctx->mail.p->stream_seq[kind][i][0];
      W.host_seq = ps.pack_seq;
    }
    TraceMark(ctx, kind ? "AC stream launch start" : "DC stream launch start", ps.stream);
    if (W.tile_end > W.tile_first)
      hipLaunchKernelGGL(pack_tile_stream_kernel,
                         dim3((unsigned)((W.tile_end - W.tile_first + kPackStreamTilesPerGroup - 1) / kPackStreamTilesPerGroup)),
                         dim3(kPackThreads), 0, ps.stream, W);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ps.launch_done[i], ps.stream));
    TraceMark(ctx, kind ? "AC stream launch done" : "DC stream launch done", ps.stream);
    if (in_kernel_sizes) continue;  // (the launch reports the sizes itself, see PackTileArgs::host_flag)
    // (IN the stream: a one-workgroup kernel on another stream waits for a free slot behind the next launch's
    // workgroups -- the first launch's word arrived when the last launch had ended)
    const PublishSeg segs[2] = {{ps.sec_bits(nsec), ps.h_sec_bits(nsec), nsec},
                                {ps.launch_sec_end.p, ps.h_launch_sec_end, (size_t)kPackMaxLaunches}};
    if ((rc = EnqueuePublish(ctx, ps.stream, segs, 2, nullptr, nullptr, &ctx->mail.p->stream_seq[kind][i][0],
                             ps.pack_seq)) != JXLT_OK)
      return rc;
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

// The DC-group sections of a two-pass frame leave by pack_deliver_kernel (experiment knob JXLT_DC_DELIVER_KERNEL=1):
// queued at once behind their writing launch, byte ranges read on the device -- no host in between.
bool DcDeliverByKernel(const jxlt_context* ctx, int kind) {
  static const bool on = [] {
    const char* e = getenv("JXLT_DC_DELIVER_KERNEL");
    return e && atoi(e) != 0;
  }();
  return on && kind == 0 && !ctx->pack[0].streamed;
}

// hipMemcpyAsync with the time the CALL took on the host (JXLT_TRACE_EVENTS: calls of more than 0.5 ms are reported)
hipError_t TimedCopy(void* dst, const void* src, size_t bytes, hipStream_t stream, const char* what) {
  if (!TraceEventsOn()) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream);
  const auto t0 = std::chrono::steady_clock::now();
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (ms > 0.5) fprintf(stderr, "jxlt slow call: hipMemcpyAsync (%s, %zu bytes) took %.3f ms on the host\n", what, bytes, ms);
  static const bool every = getenv("JXLT_TRACE_COPY_CALLS") != nullptr;
  if (every) fprintf(stderr, "jxlt copy call: %s %zu bytes %.1f us on the host\n", what, bytes, ms * 1e3);
  return e;
}

// The hand-over of the measured and (being) written sections of `kind` to `dst` (asynchronous, kernels on the copy
// stream that store to the destination themselves -- page-locked host memory or device memory): behind every launch
// of the writing pass the whole sections it completed leave, while later launches are still packing.  The kernels
// read their byte ranges from the device-side layout: the host does not have to know a size before the bytes leave.
// runs == nullptr: all sections back to back, starting at dst (end_aligned: ENDING at dst).
int EnqueueDeliver(jxlt_context* ctx, int kind, uint8_t* dst, const jxlt_section_run* runs, size_t nruns, int end_aligned) {
  jxlt_context::PackSet& ps = ctx->pack[kind];
  const size_t nsec = ps.measured_sections;
  // (the two kinds leave on streams of their own: two copy commands in flight hide each other's start-up, and the
  // DC-group sections -- a fifth of the bytes -- do not stand in front of the first AC sections)
  const hipStream_t out_stream = kind ? ctx->copy_stream : ctx->dc_copy_stream;
  DeliverArgs D;
  memset(&D, 0, sizeof(D));
  D.blob = ps.packed.p;
  D.sec_byte_offset = ps.sec_byte_off.p;
  D.launch_sec_end = ps.launch_sec_end.p;
  D.dst = dst;
  D.nsec = (int)nsec;
  D.end_aligned = end_aligned;
  D.counter = ctx->deliver_counter.p + kind * 8;  // (a counter per kind: the two kinds may be handed over side by side)
  // workgroups per hand-over kernel: the link is saturated from 64 on (tools/d2h_probe.hip); small shares take fewer
  // (experiment knob, tools/: JXLT_DELIVER_WGS)
  static const size_t max_wgs = [] {
    const char* e = getenv("JXLT_DELIVER_WGS");
    return e ? (size_t)std::max(1, std::min(atoi(e), 1024)) : (size_t)8;
  }();
  auto grid_for = [&](size_t tiles) { return (unsigned)std::min<size_t>(max_wgs, std::max<size_t>(2, tiles / 8)); };
  // A hand-over workgroup gets a CU to ITSELF: it reserves (unused) LDS so that no workgroup of the kernels that
  // run beside the hand-over fits next to it.  A compute workgroup that shares its CU's memory pipeline with waves
  // that wait for the PCIe link falls behind the rest of its kernel, and the kernel ends with its slowest workgroup:
  // beside eight unreserved hand-over workgroups the AC measuring pass took 0.18 ms instead of 0.09 (JXLT_TRACE_EVENTS).
  // (experiment knob: JXLT_DELIVER_LDS=<bytes>)
  static const unsigned deliver_lds = [] {
    const char* e = getenv("JXLT_DELIVER_LDS");
    const unsigned v = e ? (unsigned)atoi(e) : 147456u;
    if (v > 65536u) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pack_deliver_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)v);
    return v;
  }();
  // The bytes travel by COPY COMMANDS (hipMemcpyAsync: the DMA engines), issued by the host once it has the sizes
  // -- not by a kernel that stores to the destination itself, although such a kernel needs no host round trip
  // (pack_deliver_kernel: byte ranges read on the device; round 4's first version).  A kernel that stores to host
  // memory and an HBM-bound kernel beside it slow each other down badly -- tools/d2h_interfere_probe.hip: the
  // hand-over falls from 54 to 20-37 GB/s and the other kernel takes 20-35 % longer, whatever the grid, the
  // alignment or the kind of store -- while a copy command keeps 53 GB/s and costs its neighbour 2 %.  In the
  // frame: AC measuring pass 0.18 instead of 0.09 ms, writing launches 1.3-1.6x, step 5.41-5.48 against 5.26-5.31 ms.
  // (experiment knob: JXLT_DELIVER_KERNEL=1 selects the kernel)
  static const bool by_kernel = [] {
    const char* e = getenv("JXLT_DELIVER_KERNEL");
    return e && atoi(e) != 0;
  }();
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
        HIP_TRY(ctx, TimedCopy(dst + off[s_lo], ps.packed.p + off[s_lo], off[s_hi] - off[s_lo], out_stream, kind ? "AC sections" : "DC-group sections"));
        TraceMark(ctx, kind ? "AC copy done" : "DC copy done", out_stream);
      }
      s_lo = s_hi;
    }
    const int rcp = EnqueuePublish(ctx, out_stream, nullptr, 0, nullptr, nullptr, &ctx->mail.p->delivered_seq[kind][0], ++ctx->deliver_seq[kind]);
    if (rcp != JXLT_OK) return rcp;
    ctx->deliveries_pending = true;
    return JXLT_OK;
  }
  if (!(by_kernel || DcDeliverByKernel(ctx, kind)) || ps.streamed) {
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
          HIP_TRY(ctx, TimedCopy(dst + shift + (int64_t)off[s_lo], ps.packed.p + off[s_lo], off[s_hi] - off[s_lo], out_stream,
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
          HIP_TRY(ctx, hipMemcpyAsync(dst + runs[r].dst_offset, ps.packed.p + lo, hi - lo, hipMemcpyDefault, out_stream));
      }
    }
    // completion: a one-workgroup kernel behind the copies stores the hand-over's number to the word the host polls
    const int rcp = EnqueuePublish(ctx, out_stream, nullptr, 0, nullptr, nullptr, &ctx->mail.p->delivered_seq[kind][0], ++ctx->deliver_seq[kind]);
    if (rcp != JXLT_OK) return rcp;
    ctx->deliveries_pending = true;
    return JXLT_OK;
  }
  if (runs == nullptr) {
    for (int i = 0; i < ps.launches; i++) {
      HIP_TRY(ctx, hipStreamWaitEvent(out_stream, ps.launch_done[i], 0));
      D.launch = i;
      D.nruns = 0;
      const bool last = i + 1 == ps.launches;
      D.flag = last ? &ctx->mail.p->delivered_seq[kind][0] : nullptr;
      D.seq = last ? ++ctx->deliver_seq[kind] : 0;
      TraceMark(ctx, kind ? "AC deliver start" : "DC deliver start", out_stream);
      hipLaunchKernelGGL(pack_deliver_kernel, dim3(grid_for(ps.launch_t0[i + 1] - ps.launch_t0[i])), dim3(kDeliverThreads), deliver_lds,
                         out_stream, D);
      HIP_TRY(ctx, hipGetLastError());
      TraceMark(ctx, kind ? "AC deliver done" : "DC deliver done", out_stream);
    }
  } else {
    for (size_t r = 0; r < nruns; r++) {
      if ((size_t)runs[r].first_section + runs[r].num_sections > nsec) {
        ctx->error = "jxlt_pack_deliver: a run names sections the measuring pass did not see";
        return JXLT_ERR_INVALID_ARGUMENT;
      }
    }
    HIP_TRY(ctx, hipStreamWaitEvent(out_stream, ps.launch_done[ps.launches - 1], 0));
    D.launch = -1;
    for (size_t r0 = 0; r0 < nruns; r0 += kDeliverMaxRuns) {
      const size_t n = std::min<size_t>(kDeliverMaxRuns, nruns - r0);
      D.nruns = (int)n;
      for (size_t r = 0; r < n; r++) {
        D.runs[r].first = runs[r0 + r].first_section;
        D.runs[r].count = runs[r0 + r].num_sections;
        D.runs[r].dst_offset = runs[r0 + r].dst_offset;
      }
      const bool last = r0 + n >= nruns;
      D.flag = last ? &ctx->mail.p->delivered_seq[kind][0] : nullptr;
      D.seq = last ? ++ctx->deliver_seq[kind] : 0;
      hipLaunchKernelGGL(pack_deliver_kernel, dim3(grid_for(ps.max_tiles)), dim3(kDeliverThreads), deliver_lds, out_stream, D);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  ctx->deliveries_pending = true;
  return JXLT_OK;
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

}  // namespace

int jxlt_histograms_ready(jxlt_context* ctx) {
  if (!ctx) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->encoded) {
    ctx->error = "jxlt_histograms_ready needs jxlt_encode_enqueue first";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  // (a read of host memory: no call into the runtime, nothing another encoding thread of the process could wait for)
  return *(const volatile uint32_t*)&ctx->mail.p->ac_hist_seq == ctx->seq ? 1 : 0;
}

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
  // The destination must be memory a kernel can store to: page-locked host memory or device memory.  (The context's
  // own output buffer is known to be; anything else is looked up, every time -- a range that was page-locked at the
  // last call may have been freed since.)
  uint8_t* dev_dst = dst;
  const bool own_output = ctx->h_output.p && dst >= ctx->h_output.p && dst <= ctx->h_output.p + ctx->h_output.cap;
  if (!own_output) {
    hipPointerAttribute_t attr;
    const hipError_t pe = hipPointerGetAttributes(&attr, dst);
    (void)hipGetLastError();
    if (pe != hipSuccess || (attr.type != hipMemoryTypeHost && attr.type != hipMemoryTypeDevice)) {
      ctx->error = "jxlt_pack_deliver needs page-locked host memory (jxlt_output_buffer / jxlt_pinned_alloc / "
                   "jxlt_pinned_register) or device memory as destination";
      return JXLT_ERR_INVALID_ARGUMENT;
    }
    if (attr.type == hipMemoryTypeHost) {
      void* mapped = nullptr;
      if (hipHostGetDevicePointer(&mapped, dst, 0) != hipSuccess || !mapped) {
        (void)hipGetLastError();
        ctx->error = "jxlt_pack_deliver: the destination is page-locked but not mapped into the device's address space";
        return JXLT_ERR_INVALID_ARGUMENT;
      }
      dev_dst = static_cast<uint8_t*>(mapped);
    }
  }
  if (kind == 0) {
    if (ctx->deferred_dc.pending) {  // (a second hand-over of the kind: the first one first)
      const int rcd = IssueDeferred(ctx, /*wait=*/true);
      if (rcd != JXLT_OK) return rcd;
    }
    if (!SizesReady(ctx, 0) && !DcDeliverByKernel(ctx, 0)) {
      jxlt_context::DeferredDeliver& d = ctx->deferred_dc;
      d.pending = true;
      d.dst = dev_dst;
      d.end_aligned = end_aligned;
      d.runs.assign(runs, runs + num_runs);
      for (size_t r = 0; r < num_runs; r++) {
        if ((size_t)runs[r].first_section + runs[r].num_sections > ctx->pack[0].measured_sections) {
          d.pending = false;
          ctx->error = "jxlt_pack_deliver: a run names sections the measuring pass did not see";
          return JXLT_ERR_INVALID_ARGUMENT;
        }
      }
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

size_t jxlt_release_cached_memory(int device_ordinal) { return DeviceBlockCache::Get().Release(device_ordinal); }

int jxlt_output_buffer(jxlt_context* ctx, size_t bytes, uint8_t** out) {
  if (!ctx || !out) return JXLT_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->h_output.cap < bytes) {
    if ((ctx->copies_pending || ctx->deliveries_pending) && ctx->h_output.p) {
      // sections may be on their way into the buffer (jxlt_pack_deliver): it grows with its contents
      if (ctx->deliveries_pending) {
        const int rcw = WaitDeliveries(ctx);
        if (rcw != JXLT_OK) return rcw;
        ctx->deliveries_pending = false;
      }
      if (ctx->copies_pending) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
      PinnedBuf<uint8_t> grown;
      const int rc = EnsurePinned(ctx, &grown, bytes + bytes / 8 + 65536);
      if (rc != JXLT_OK) return rc;
      memcpy(grown.p, ctx->h_output.p, ctx->h_output.cap);
      FreePinned(&ctx->h_output);
      ctx->h_output = grown;
    } else {
      const int rc = EnsurePinned(ctx, &ctx->h_output, bytes + bytes / 8 + 65536);
      if (rc != JXLT_OK) return rc;
    }
  }
  *out = ctx->h_output.p;
  return JXLT_OK;
}


int jxlt_kernel_times(jxlt_context* ctx, jxlt_kernel_time* out, int cap) {
  if (!ctx || !out || cap < 0) return JXLT_ERR_INVALID_ARGUMENT;
  if (!ctx->profiled) {
    ctx->error = "nothing encoded yet";
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
    HIP_TRY(ctx, hipMemcpy(host_dst, ctx->dbg_phase.p, bytes, hipMemcpyDeviceToHost));
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
