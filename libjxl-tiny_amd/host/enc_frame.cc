// EncodeFrame: device pixel pipeline (through the C ABI of libjxltiny_hip.so)
// followed by host bitstream assembly.  Counterpart of
// /root/reference/encoder/enc_frame.cc:818-860 with the per-DC-group loop
// (:839-844) replaced by one device pass over all groups.
#include "encoder/enc_frame.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/jxl_tiny_amd.h"
#include "entropy_coder.h"
#include "frame_assembler.h"
#include "host_internal.h"

namespace jxlt {

namespace {
FrameView ViewOf(const jxlt_frame_result& res, const uint8_t* const* group_ptr, const size_t* group_len) {
  FrameView view;
  view.xsize = res.xsize;
  view.ysize = res.ysize;
  for (int c = 0; c < 3; ++c) view.quant_dc[c] = res.quant_dc[c];
  view.raw_quant_field = res.raw_quant_field;
  view.ac_strategy = res.ac_strategy;
  view.ytox_map = res.ytox_map;
  view.ytob_map = res.ytob_map;
  view.group_tokens = group_ptr;
  view.group_token_bytes = group_len;
  return view;
}
}  // namespace

// Device pipeline for the image set on `ctx`, then assembly.  Multi-section
// frames keep the raw tokens in HBM: the device returns symbol histograms, the
// host builds the prefix codes (and, concurrently, the DC-group sections), the
// device packs the AC sections.  Single-group frames (bit-concatenated sections,
// enc_frame.cc:805-811) take the raw-token route.
// Appends the frame to `writer` (if non-null); otherwise asks `placer(frame_bytes)` for the
// destination and writes the frame there (the AC blob comes straight from the device).
namespace {
std::atomic<bool> g_emulate_static_constants{false};
std::atomic<float> g_first_distance{0.0f};  // latched by the first frame while the emulation is on
}  // namespace

void SetStaticConstantEmulation(bool on) {
  g_emulate_static_constants.store(on);
  if (!on) g_first_distance.store(0.0f);
}

// The reference's function-local static constants (enc_ac_strategy.cc:178-185): while the emulation is on,
// every encode of the process uses the FIRST frame's distance for the two multipliers.  Off (default): the
// context keeps whatever its owner set through jxlt_set_strategy_distance (0 = each encode's own distance).
void ApplyStrategyDistanceEmulation(jxlt_context* ctx, float distance) {
  static thread_local jxlt_context* touched = nullptr;  // a context this thread switched to a latched distance
  if (g_emulate_static_constants.load()) {
    float expected = 0.0f;
    g_first_distance.compare_exchange_strong(expected, distance);  // first frame wins
    jxlt_set_strategy_distance(ctx, g_first_distance.load());
    touched = ctx;
  } else if (touched == ctx) {
    jxlt_set_strategy_distance(ctx, 0.0f);  // emulation was switched off again: undo our own setting only
    touched = nullptr;
  }
}

// One helper thread per encoding thread: builds the DC code while the encoding thread waits for the AC histogram
// and, when that arrives first (small frames, frames that came over PCIe, slabs: their AC tokenisation is shorter
// than a DC code construction), builds the AC code at the same time.  Sleeps between frames.
class CodeWorker {
 public:
  CodeWorker() : thread_([this] { Loop(); }) {}
  ~CodeWorker() {
    {
      std::lock_guard<std::mutex> lock(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    thread_.join();
  }
  void Start(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      job_ = std::move(job);
      done_.store(false, std::memory_order_relaxed);
      pending_ = true;
    }
    cv_.notify_all();
  }
  bool Done() const { return done_.load(std::memory_order_acquire); }
  void Wait() {
    for (int spin = 0; spin < 20000 && !Done(); ++spin) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    if (Done()) return;
    std::unique_lock<std::mutex> lock(mu_);
    cv_.wait(lock, [this] { return done_.load(std::memory_order_acquire); });
  }

 private:
  void Loop() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lock(mu_);
        cv_.wait(lock, [this] { return pending_ || quit_; });
        if (quit_) return;
        pending_ = false;
        job = std::move(job_);
      }
      job();
      {
        std::lock_guard<std::mutex> lock(mu_);
        done_.store(true, std::memory_order_release);
      }
      cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::function<void()> job_;
  bool pending_ = false, quit_ = false;
  std::atomic<bool> done_{true};
  std::thread thread_;
};

// (jxlt_last_frame_timeline: the stage times of the calling thread's last frame)
thread_local jxlt_frame_timeline g_last_timeline = {0, 0, 0, 0, 0};
thread_local bool g_have_timeline = false;
bool LastFrameTimeline(jxlt_frame_timeline* out) {
  if (!g_have_timeline) return false;
  *out = g_last_timeline;
  return true;
}

bool EncodeFrameOnContext(jxlt_context* ctx, float distance, int num_threads, jxl::BitWriter* writer,
                          const std::function<uint8_t*(size_t)>* placer, ContextOutput* in_context) {
  static const bool trace = getenv("JXLT_TRACE") != nullptr;
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  const DistanceParams distp = ComputeDistanceParams(distance);
  jxlt_params params;
  params.distance = distp.distance;
  params.scale = distp.scale;
  params.inv_scale = distp.inv_scale;
  params.scale_dc = distp.scale_dc;
  params.x_qm_scale = distp.x_qm_scale;
  params.flags = 0;
  ApplyStrategyDistanceEmulation(ctx, distp.distance);
  if (jxlt_encode_enqueue(ctx, &params) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: device encode failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  size_t xsize = 0, ysize = 0;
  if (jxlt_image_size(ctx, &xsize, &ysize) != JXLT_OK) return false;
  const size_t num_groups = ((xsize + 255) / 256) * ((ysize + 255) / 256);
  const size_t num_dc_groups = ((xsize + 2047) / 2048) * ((ysize + 2047) / 2048);
  if (num_groups + num_dc_groups == 2) {
    jxlt_frame_result res;
    if (jxlt_fetch_result(ctx, &res) != JXLT_OK) return false;
    const uint8_t* ptr = res.tokens;
    const size_t len = static_cast<size_t>(res.group_token_offset[1]);
    jxl::BitWriter local;
    jxl::BitWriter* w = writer ? writer : &local;
    if (!AssembleFrame(ViewOf(res, &ptr, &len), distp, w, num_threads)) return false;
    if (in_context) {
      const std::vector<uint8_t>& b = local.Bytes();
      const size_t pre = in_context->prefix ? in_context->prefix->size() : 0;
      uint8_t* buf = nullptr;
      if (jxlt_output_buffer(ctx, pre + b.size(), &buf) != JXLT_OK) return false;
      if (pre) memcpy(buf, in_context->prefix->data(), pre);
      memcpy(buf + pre, b.data(), b.size());
      in_context->data = buf;
      in_context->size = pre + b.size();
    } else if (!writer) {
      const std::vector<uint8_t>& b = local.Bytes();
      uint8_t* dst = (*placer)(b.size());
      if (!dst) return false;
      memcpy(dst, b.data(), b.size());
    }
    return true;
  }
  // The DC histogram arrives first (the DC-group tokenisation runs ahead of the AC one): the DC
  // code is built while the device is still tokenising.  The arrival times are the previous
  // frame's, to a good approximation (same context, usually same geometry); the helper threads
  // of the code construction are woken just before (entropy_coder.h).
  // (the expectation belongs to a frame size: a frame of another size starts without one)
  static thread_local double expected_dc_ms = 0.0, expected_ac_ms = 0.0;
  static thread_local size_t expected_for_pixels = 0;
  // (a code construction that worked alone last time is not announced to the helper threads)
  static thread_local bool dc_shared = true, ac_shared = true;
  if (expected_for_pixels != xsize * ysize) {
    expected_for_pixels = xsize * ysize;
    expected_dc_ms = expected_ac_ms = 0.0;
  }
  EntropyCode ac_code, dc_code;
  std::vector<uint32_t> ac_table(64 * 64), dc_table(64 * 64);
  if (expected_dc_ms > 1.0 && dc_shared) WarmCodeConstruction(expected_dc_ms - 0.5, expected_dc_ms + 1.5);
  const uint32_t *ac_hist = nullptr, *dc_hist = nullptr;
  if (jxlt_fetch_dc_histogram(ctx, &dc_hist) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: fetch failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  const auto t0a = now();
  expected_dc_ms = ms(t0, t0a);
  // The DC code is built by the helper thread.  Whatever is ready first goes to the device first:
  //  * the DC code (large resident frames: token_kernel runs for longer than a DC code construction): the DC-group
  //    sections are packed behind token_kernel and leave for the host while this thread builds the AC code;
  //  * the AC histogram (small frames, frames that came over PCIe, slabs): this thread builds the AC code at once,
  //    the AC sections are packed and leave; the DC-group sections follow when their code exists.
  // Neither kind waits for the other, because nothing in the codestream is placed from the left: the AC sections
  // start at a position that is fixed before any size is known (E0 + ACGlobal, E0 = a bound of what stands in front),
  // the DC-group sections END at E0 -- the device right-aligns them itself (jxlt_pack_deliver, end_aligned) --, and
  // the head (file header, frame header, TOC, DCGlobal) is set against them from the right when the sizes have
  // arrived.  The codestream then starts a little way into the buffer.
  static thread_local std::unique_ptr<CodeWorker> worker;
  if (!worker) worker.reset(new CodeWorker);
  FrameGlobals globals;
  bool dc_shared_now = false;
  double dc_job_ms[3] = {0, 0, 0};  // (trace: start of the helper's job, code, table + DCGlobal; from the DC histogram's arrival)
  worker->Start([&] {
    dc_job_ms[0] = ms(t0a, now());
    (void)TakeClusteringShared();
    BuildDcCode(dc_hist, &dc_code);
    dc_shared_now = TakeClusteringShared();
    dc_job_ms[1] = ms(t0a, now());
    FillCodeTable(dc_code, dc_table.data());
    globals.dc_global = BuildDcGlobal(xsize, ysize, distp, dc_code);
    dc_job_ms[2] = ms(t0a, now());
  });
  // (a worker that is still busy at a return would write to this frame's locals)
  struct Joiner {
    CodeWorker* w;
    ~Joiner() { w->Wait(); }
  } joiner{worker.get()};
  bool ac_first = false;
  for (;;) {
    if (worker->Done()) break;
    const int ready = jxlt_histograms_ready(ctx);  // (a read of host memory)
    if (ready < 0) {
      fprintf(stderr, "jxl_tiny_amd: device encode failed: %s\n", jxlt_last_error(ctx));
      return false;
    }
    if (ready == 1) {
      ac_first = true;
      break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  const size_t pre = (in_context && in_context->prefix) ? in_context->prefix->size() : 0;
  const auto align = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  // Bytes the sections of a kind can take at most: their tokens' code lengths from the histogram -- with the code's
  // own lengths when it exists (exact but for the padding of every section to a byte and the handful of raw bits a
  // DC-group section starts with), with the longest code word possible (15 bits) when it does not.
  const auto section_bytes_bound = [](const uint32_t* hist, const uint32_t* table, size_t nsec) {
    uint64_t bits = 0;
    for (size_t ctx_i = 0; ctx_i < 64; ++ctx_i)
      for (size_t sym = 0; sym < 64; ++sym) {
        const uint32_t n = hist[ctx_i * 64 + sym];
        if (n == 0) continue;
        const uint32_t extra = sym >= 16 ? static_cast<uint32_t>(sym >> 2) - 2u : 0u;  // (token.h:32-48)
        bits += static_cast<uint64_t>(n) * ((table ? table[ctx_i * 64 + sym] >> 16 : 15u) + extra);
      }
    return static_cast<size_t>(bits / 8) + 16 * nsec + 64;
  };
  static thread_local size_t last_frame_bytes = 0;  // (sizes the output buffer before the AC sections' size is known)
  size_t e0 = 0;       // where the DC-group sections end and ACGlobal starts
  uint8_t* buf = nullptr;
  bool dc_begun = false;
  // (the DC-group sections' hand-over is asked for at once: the library issues the copy commands when the sections'
  // sizes arrive -- from inside its next wait, e.g. the one for the AC histogram -- and does not block for them)
  const auto begin_dc = [&]() -> bool {
    // (the DC code, and with it globals.dc_global, is complete here)
    dc_shared = dc_shared_now;
    if (jxlt_pack_begin(ctx, 0, dc_table.data()) != JXLT_OK) {
      fprintf(stderr, "jxl_tiny_amd: section packing failed: %s\n", jxlt_last_error(ctx));
      return false;
    }
    dc_begun = true;
    if (jxlt_pack_deliver(ctx, 0, buf + e0, nullptr, 0, /*end_aligned=*/1) != JXLT_OK) {
      fprintf(stderr, "jxl_tiny_amd: section hand-over failed: %s\n", jxlt_last_error(ctx));
      return false;
    }
    return true;
  };
  if (!ac_first) {
    e0 = align(pre + HeadSizeBound(xsize, ysize, globals) + section_bytes_bound(dc_hist, dc_table.data(), num_dc_groups));
    if (jxlt_output_buffer(ctx, std::max(e0 + 4096, last_frame_bytes), &buf) != JXLT_OK) {
      fprintf(stderr, "jxl_tiny_amd: output buffer: %s\n", jxlt_last_error(ctx));
      return false;
    }
    if (!begin_dc()) return false;
  }
  const auto t0b = now();
  {
    const double left = expected_ac_ms - ms(t0, t0b);  // until the AC histogram is expected
    if (!ac_first && expected_ac_ms > 1.0 && ac_shared)
      WarmCodeConstruction(left > 0.5 ? left - 0.5 : 0.0, (left > 0.0 ? left : 0.0) + 1.5);
  }
  if (jxlt_fetch_histograms(ctx, &ac_hist, nullptr) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: fetch failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  const auto t1 = now();
  expected_ac_ms = ms(t0, t1);
  BuildAcCode(ac_hist, &ac_code);
  const auto t1a = now();
  ac_shared = TakeClusteringShared();
  FillCodeTable(ac_code, ac_table.data());
  const auto t1b = now();
  // (the device needs the table only: the AC sections' packing is queued before ACGlobal -- the code's serialisation,
  // 10 us -- is written, and runs beside it)
  if (jxlt_pack_begin(ctx, 1, ac_table.data()) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: section packing failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  globals.ac_global = BuildAcGlobal(xsize, ysize, ac_code);
  const auto t2 = now();
  if (trace)
    fprintf(stderr, "jxlt trace: AC code: clustering + Huffman %.3f ms, table %.3f, ACGlobal %.3f\n", ms(t1, t1a), ms(t1a, t1b),
            ms(t1b, t2));
  const size_t acg_bytes = globals.ac_global.size();
  if (ac_first) {
    // (DCGlobal does not exist yet: its bound -- context tree + one clustered code for 45 contexts -- is 16 KB)
    FrameGlobals none;
    e0 = align(pre + HeadSizeBound(xsize, ysize, none) + 16384 + section_bytes_bound(dc_hist, nullptr, num_dc_groups));
  }
  const size_t ac_bound = section_bytes_bound(ac_hist, ac_table.data(), num_groups);
  // (with DC-group sections on their way the buffer grows with its contents; normally it has its size from the last frame)
  if (jxlt_output_buffer(ctx, e0 + acg_bytes + ac_bound + 16, &buf) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: output buffer: %s\n", jxlt_last_error(ctx));
    return false;
  }
  // (AC code first: the DC code was started before it on the other thread and is shorter -- its sections are queued
  // BEFORE the AC sections' hand-over is asked for, which returns when the last AC launch has reported: 4096^2, the
  // DC-group sections' kernel started 0.04 ms behind the last AC launch instead of right behind it)
  if (!dc_begun) {
    worker->Wait();
    if (!begin_dc()) return false;
  }
  if (jxlt_pack_deliver(ctx, 1, buf + e0 + acg_bytes, nullptr, 0, /*end_aligned=*/0) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: section hand-over failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  memcpy(buf + e0, globals.ac_global.data(), acg_bytes);
  if (trace)
    fprintf(stderr, "jxlt trace: helper: job started %.3f ms after the DC histogram, DC code %.3f, table + DCGlobal %.3f\n",
            dc_job_ms[0], dc_job_ms[1], dc_job_ms[2]);
  if (trace)
    fprintf(stderr, "jxlt trace: dc histogram after %.2f ms, %s first | ac histogram after %.2f ms, ac code %.3f ms\n",
            ms(t0, t0a), ac_first ? "AC code" : "DC code", ms(t0, t1), ms(t1, t2));
  // The sizes of both kinds (all the TOC needs) arrive while the sections are being written and handed over.
  jxlt_packed_sections dcm, acm;
  if (jxlt_pack_sizes(ctx, 0, &dcm) != JXLT_OK || jxlt_pack_sizes(ctx, 1, &acm) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: section measuring failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  const auto t3 = now();
  const PackedSections dc = {nullptr, dcm.section_offset, dcm.section_bits, dcm.num_sections};
  const PackedSections ac = {nullptr, acm.section_offset, acm.section_bits, acm.num_sections};
  const size_t dc_bytes = static_cast<size_t>(dc.offset[dc.n]), ac_bytes = static_cast<size_t>(ac.offset[ac.n]);
  std::vector<uint8_t> head;
  if (!BuildFrameHead(xsize, ysize, distp, globals, dc, ac, &head)) return false;
  if (pre + head.size() + dc_bytes > e0 || ac_bytes > ac_bound) {
    // (cannot happen: both bounds are sums of the code lengths of the very tokens that were packed)
    fprintf(stderr, "jxl_tiny_amd: internal error: sections larger than their bound\n");
    (void)jxlt_synchronize(ctx);
    return false;
  }
  uint8_t* const frame_at = buf + e0 - dc_bytes - head.size();
  const size_t frame_bytes = head.size() + dc_bytes + acg_bytes + ac_bytes;
  if (pre) memcpy(frame_at - pre, in_context->prefix->data(), pre);
  memcpy(frame_at, head.data(), head.size());
  last_frame_bytes = e0 + acg_bytes + ac_bytes + 16;  // (the LAST frame: a thread that once had a large frame does not ask for its size for ever)
  bool ok = jxlt_synchronize(ctx) == JXLT_OK;
  if (ok) {
    if (in_context) {
      in_context->data = frame_at - pre;
      in_context->size = pre + frame_bytes;
    } else if (writer) {
      // BitWriter API of the drop-in EncodeFrame: the frame is appended from the page-locked buffer
      writer->Reserve(frame_bytes);
      writer->AppendBytes(frame_at, frame_bytes);
    } else {
      // placer(frame_bytes) returns where the frame must be written
      uint8_t* dst = (*placer)(frame_bytes);
      ok = dst != nullptr;
      if (ok) memcpy(dst, frame_at, frame_bytes);
    }
  }
  const auto t4 = now();
  g_last_timeline = {ms(t0, t0a), ms(t0, t1), ms(t0, t2), ms(t0, t3), ms(t0, t4)};
  g_have_timeline = true;
  if (trace)
    fprintf(stderr, "jxlt trace: device+histograms %.3f ms | codes %.3f | sizes %.3f | head + hand-over %.3f\n",
            ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
  (void)num_threads;
  return ok;
}

}  // namespace jxlt

namespace jxl {
namespace {

thread_local int g_device = 0;

// One device context per host thread, re-created when the device changes.
struct ThreadContext {
  jxlt_context* ctx = nullptr;
  int device = -1;
  ~ThreadContext() {
    if (ctx) jxlt_context_destroy(ctx);
  }
};
thread_local ThreadContext g_tls;

}  // namespace

jxlt_context* AcquireContextForThread() {
  if (g_tls.ctx && g_tls.device == g_device) return g_tls.ctx;
  if (g_tls.ctx) {
    jxlt_context_destroy(g_tls.ctx);
    g_tls.ctx = nullptr;
  }
  if (jxlt_context_create(g_device, &g_tls.ctx) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: cannot create device context: %s\n", jxlt_last_error(nullptr));
    g_tls.ctx = nullptr;
    return nullptr;
  }
  g_tls.device = g_device;
  return g_tls.ctx;
}

void SetEncoderDevice(int device_ordinal) { g_device = device_ordinal; }

void EmulateReferenceStaticConstants(bool on) { jxlt::SetStaticConstantEmulation(on); }
void EmulateReferenceSingleSymbolCodes(bool on) { jxlt::SetReferenceSingleSymbolEmulation(on); }

}  // namespace jxl

namespace jxlt {
jxlt_context* AcquireThreadContext() { return jxl::AcquireContextForThread(); }
}  // namespace jxlt

namespace jxl {

namespace {
// The device list for frames over several GPUs and the encoder built on it belong to the CALLING THREAD, like the
// single device of SetEncoderDevice and its context: every thread that encodes over a list owns a
// jxlt_multi_encoder (contexts + worker threads) of its own, made at its first such frame and destroyed with the
// thread or when its list changes.  Two threads with two lists -- or with the same list -- encode side by side;
// nothing is process-wide, nothing is locked.  (Until round 4: ONE process-wide list and encoder behind a mutex,
// concurrent callers serialised)  A thread that has not called SetEncoderDevices takes the
// list of the environment variable JXLT_DEVICES ("0,1,2,3" or "all"), read once.
struct ThreadDeviceList {
  bool explicit_list = false;   // SetEncoderDevices was called on this thread
  std::vector<int> devices;     // ... with this list
  jxlt_multi_encoder* enc = nullptr;
  std::vector<int> enc_devices;  // what `enc` was made for
  void Drop() {
    if (enc) jxlt_multi_encoder_destroy(enc);
    enc = nullptr;
    enc_devices.clear();
  }
  ~ThreadDeviceList() { Drop(); }
};
thread_local ThreadDeviceList t_list;

const std::vector<int>& DevicesFromEnvironment() {
  static const std::vector<int> devices = [] {
    std::vector<int> v;
    const char* e = getenv("JXLT_DEVICES");
    if (!e || !*e) return v;
    if (strcmp(e, "all") == 0) {
      const int n = jxlt_device_count();
      for (int d = 0; d < n && d < 64; ++d) v.push_back(d);
      return v;
    }
    for (const char* p = e; *p;) {
      char* end = nullptr;
      const long d = strtol(p, &end, 10);
      if (end == p) break;
      if (d >= 0 && d < 1024) v.push_back(static_cast<int>(d));
      p = *end == ',' ? end + 1 : end;
    }
    return v;
  }();
  return devices;
}
}  // namespace

void SetEncoderDevices(const int* device_ordinals, int n) {
  t_list.explicit_list = true;  // an explicit call wins over the environment (also an empty list)
  t_list.devices.assign(device_ordinals, device_ordinals + (n > 0 && device_ordinals ? n : 0));
  if (t_list.enc && t_list.enc_devices != t_list.devices) t_list.Drop();
}

}  // namespace jxl

namespace jxlt {
// The frame over the calling thread's device list (jxl::SetEncoderDevices / JXLT_DEVICES) when that names several
// GPUs and the frame has more than one DC group (a PFM payload: more than one row of DC groups).  *used = false: not applicable, nothing done.
// Otherwise the complete codestream (file header + frame) is in *codestream, or false is returned and
// *failure_code says why (the participants' own code: JXLT_ERR_UNSUPPORTED when a device refused the frame's values).
bool EncodeOnDeviceList(const float* const planes[3], size_t pitch_bytes, const void* pfm_payload, int big_endian,
                        size_t xsize, size_t ysize, float distance, std::vector<uint8_t>* codestream, bool* used,
                        int* failure_code) {
  using namespace jxl;
  if (failure_code) *failure_code = JXLT_ERR_INTERNAL;
  const std::vector<int>& devices = t_list.explicit_list ? t_list.devices : DevicesFromEnvironment();
  *used = devices.size() > 1 && (ysize > 2048 || (xsize > 2048 && pfm_payload == nullptr));
  if (!*used) return true;
  if (t_list.enc && t_list.enc_devices != devices) t_list.Drop();
  if (!t_list.enc) {
    if (jxlt_multi_encoder_create(devices.data(), static_cast<int>(devices.size()), &t_list.enc) != JXLT_OK) {
      fprintf(stderr, "jxl_tiny_amd: cannot create device contexts: %s\n", jxlt_last_error(nullptr));
      t_list.enc = nullptr;
      if (failure_code) *failure_code = JXLT_ERR_NO_DEVICE;
      return false;  // no CPU fallback by design
    }
    t_list.enc_devices = devices;
  }
  jxlt_multi_encoder* const list_encoder = t_list.enc;
  const uint8_t* bytes = nullptr;
  size_t size = 0;
  const int rc = pfm_payload
                     ? jxlt_multi_encoder_encode_pfm(list_encoder, pfm_payload, xsize, ysize, big_endian, distance, &bytes, &size)
                     : jxlt_multi_encoder_encode(list_encoder, planes, pitch_bytes, xsize, ysize, distance, &bytes, &size);
  if (rc != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: sharded encode failed: %s\n", jxlt_multi_encoder_last_error(list_encoder));
    if (failure_code) *failure_code = rc;
    return false;
  }
  codestream->assign(bytes, bytes + size);
  return true;
}
}  // namespace jxlt

namespace jxl {

Status EncodeFrame(const float distance, const Image3F& linear, ThreadPool* pool,
                   BitWriter* writer) {
  if (linear.xsize() == 0 || linear.ysize() == 0 || !(distance > 0)) return false;
  const float* planes[3] = {linear.ConstPlaneRow(0, 0), linear.ConstPlaneRow(1, 0),
                            linear.ConstPlaneRow(2, 0)};
  {
    std::vector<uint8_t> whole;
    bool used = false;
    if (!jxlt::EncodeOnDeviceList(planes, linear.bytes_per_row(), nullptr, 0, linear.xsize(), linear.ysize(), distance,
                                  &whole, &used))
      return false;
    if (used) {
      // the device list returns the whole codestream; the caller's writer already holds the file header
      BitWriter header;
      if (!jxlt::WriteFileHeader(linear.xsize(), linear.ysize(), &header)) return false;
      const size_t skip = header.TakeBytes().size();
      writer->AppendBytes(whole.data() + skip, whole.size() - skip);
      return true;
    }
  }
  jxlt_context* ctx = AcquireContextForThread();
  if (!ctx) return false;  // no CPU fallback by design
  if (jxlt_image_upload(ctx, planes, linear.bytes_per_row(), linear.xsize(), linear.ysize()) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: upload failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  return jxlt::EncodeFrameOnContext(ctx, distance, pool ? pool->NumThreads() : 0, writer, nullptr);
}

}  // namespace jxl
