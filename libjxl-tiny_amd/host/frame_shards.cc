// One frame over several GPUs: row slabs of whole DC groups, one participant (device context + host
// thread or process) per slab.  See include/jxl_tiny_amd.h ("one frame sharded over several GPUs").
//
// Counterpart in the reference: the DC-group loop /root/reference/encoder/enc_frame.cc:839-844 (independent
// units), the code optimisation over ALL sections :846-850 (the one global dependency: here a host-side sum
// of the participants' 2 x 64 x 64 histograms), and CombineSections :804-816 (here: every participant's
// sections land in their byte range of one buffer, participant 0 writes header + TOC + globals in front).
//
// The participants meet in a control block that lives either on the heap (threads of one process) or in a
// POSIX shared-memory segment (one process per GPU).  All synchronisation is a handful of monotonic
// atomic counters in that block; nothing here touches the GPUs except through the slab operations.
#include <fcntl.h>
#include <limits.h>
#include <linux/futex.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/jxl_tiny_amd_testing.h"
#include "encoder/enc_bit_writer.h"
#include "entropy_coder.h"
#include "frame_assembler.h"
#include "host_internal.h"

namespace jxlt {
namespace {

constexpr int kMaxWorld = 64;
constexpr uint64_t kMagic = 0x6a786c7473686431ull;  // "jxltshd1"
constexpr size_t kHistWords = 64 * 64;

// Phases every participant arrives at / values participant 0 publishes (per frame).
enum Arrival { kDcHist = 0, kAcHist, kSizes, kPlaced, kNumArrivals };
enum Publication { kDcTable = 0, kAcTable, kLayout, kNumPublications };

// Lives at offset 0 of the region.  Plain data + lock-free atomics only (shared between processes).
struct Control {
  uint64_t magic;
  uint32_t world;
  uint32_t reserved;
  uint64_t max_sections;     // capacity of the two section tables behind the control block
  uint64_t output_offset;    // where the codestream is assembled, from the start of the region
  uint64_t output_capacity;
  uint64_t region_bytes;
  std::atomic<uint64_t> arrived[kNumArrivals];       // += 1 per participant and frame
  std::atomic<uint64_t> published[kNumPublications];  // = frame number
  std::atomic<int32_t> failed;                        // sticky first error
  std::atomic<uint32_t> sleepers;                     // participants asleep in a futex wait on one of the counters
  std::atomic<uint32_t> attached;                     // processes that have mapped the segment (its name goes once all have)
  uint64_t dc_begin, ac_begin;                        // kLayout: where the DC-group / AC-group sections start in the output
                                                      // (every section's offset from there: the sec_off table)
  uint32_t dc_table[kHistWords], ac_table[kHistWords];
  uint32_t ac_global_size;                            // kAcTable: the serialised ACGlobal section (its builder is
  uint8_t ac_global[16384];                           // not the participant that assembles the frame)
  uint32_t hist[kMaxWorld][2][kHistWords];            // [participant][0 = AC, 1 = DC]
};
static_assert(std::atomic<uint64_t>::is_always_lock_free, "the control block needs lock-free atomics");

size_t ControlBytes() { return (sizeof(Control) + 4095) & ~size_t(4095); }
// (per section: bits u32, bytes u32, offset u64 -- the offset from the start of its kind's sections, kLayout)
size_t TablesBytes(size_t max_sections) { return (max_sections * (2 * sizeof(uint32_t) + sizeof(uint64_t)) + 8 + 4095) & ~size_t(4095); }

}  // namespace
}  // namespace jxlt

struct jxlt_shard_group {
  uint8_t* base = nullptr;   // region: Control | sec_bits[max] | sec_bytes[max] | output
  size_t bytes = 0;
  jxlt::Control* ctl = nullptr;
  int rank = 0;              // -1: in-process group (participants pass their rank explicitly)
  int world = 1;
  bool shm = false;
  bool registered = false;   // output area page-locked for this process's devices
  bool unlinked = false;     // (rank 0) the segment's name is gone already
  std::string name;
  std::string error;         // first failure (several participants of one process may report)
  std::mutex error_mu;
  void SetError(const std::string& what) {
    std::lock_guard<std::mutex> lock(error_mu);
    if (error.empty() || ctl == nullptr || ctl->failed.load() == 0) error = what;
  }
  uint64_t frame[jxlt::kMaxWorld] = {};  // frames begun, per participant of this process
  // Host-side stage times of the last frame of the participant this handle belongs to (rank >= 0; an in-process
  // group keeps participant 0's), milliseconds from the call's start: device pipeline enqueued; DC histogram here; AC
  // histogram here; both code tables here; own section sizes here; layout here; section hand-over issued; all placed.
  double timeline_ms[8] = {};
  uint32_t* sec_bits() const { return reinterpret_cast<uint32_t*>(base + jxlt::ControlBytes()); }
  uint32_t* sec_bytes() const { return sec_bits() + ctl->max_sections; }
  uint64_t* sec_off() const { return reinterpret_cast<uint64_t*>(sec_bits() + 2 * ((ctl->max_sections + 1) & ~uint64_t(1))); }
  uint8_t* output() const { return base + ctl->output_offset; }
};

namespace jxlt {
namespace {

// The frame's DC groups (2048 x 2048 pixels: the reference's unit of independent work, enc_frame.cc:839-844) dealt
// out to `world` participants as RECTANGLES of whole DC groups -- a slab must be a rectangle of pixels to be a frame
// of its own to the kernels, and a rectangle's sections are a few runs of the codestream (one per row of DC groups /
// of groups).  The grid of xdc x ydc DC groups is cut into B bands of rows; band b is cut into n_b column ranges,
// sum n_b = the participants that get work (at most one per DC group); band heights follow the n_b.  Of all B the
// one with the smallest largest rectangle wins, ties go to the larger B (fewer column cuts: fewer runs).  Examples:
// 8 x 8 (16384^2) over 8: eight rows of DC groups, as in rounds 1-3; 8 x 1 (16384 x 2048) over 8: one DC group
// each; 4 x 4 (8192^2) over 8: four bands of two rectangles of 2 x 1; 8 x 8 over 5: two bands (three rectangles
// of 3|3|2 x 5, two of 4 x 3) -- 15 DC groups at most instead of 16 with rows alone.
// rows_only: bands only, one participant per band (a PFM payload is cut along its rows: they are contiguous in the file).
struct DcRect {
  size_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;  // in DC groups; empty: no work for this participant
  bool empty() const { return x1 <= x0 || y1 <= y0; }
};
void ShardPartition(size_t xdc, size_t ydc, int world, bool rows_only, DcRect* out) {
  for (int r = 0; r < world; ++r) out[r] = DcRect();
  const size_t w = std::min<size_t>(static_cast<size_t>(world), rows_only ? ydc : xdc * ydc);  // participants with work
  if (w == 0) return;
  size_t best_b = 0, best_cost = ~size_t(0);
  std::vector<size_t> edges, best_edges;
  const size_t b_lo = rows_only ? w : 1, b_hi = std::min(ydc, w);
  for (size_t nb = b_lo; nb <= b_hi; ++nb) {
    const size_t base = w / nb, extra = w % nb;  // band k has base + (k < extra) participants
    if (base + (extra ? 1 : 0) > xdc) continue;
    // band boundaries: heights proportional to the bands' participants, every band at least one row
    edges.assign(nb + 1, 0);
    size_t cum = 0, cost = 0;
    for (size_t k = 0; k < nb; ++k) {
      const size_t n = base + (k < extra ? 1 : 0);
      cum += n;
      size_t e = k + 1 == nb ? ydc : (ydc * cum + w / 2) / w;
      e = std::max(e, edges[k] + 1);
      e = std::min(e, ydc - (nb - 1 - k));
      edges[k + 1] = e;
      cost = std::max(cost, ((xdc + n - 1) / n) * (e - edges[k]));
    }
    if (cost < best_cost || (cost == best_cost && nb > best_b)) {
      best_cost = cost;
      best_b = nb;
      best_edges = edges;
    }
  }
  if (best_b == 0) return;
  const size_t base = w / best_b, extra = w % best_b;
  int r = 0;
  for (size_t k = 0; k < best_b; ++k) {
    const size_t n = base + (k < extra ? 1 : 0);
    for (size_t j = 0; j < n; ++j, ++r) {
      out[r].x0 = xdc * j / n;
      out[r].x1 = xdc * (j + 1) / n;
      out[r].y0 = best_edges[k];
      out[r].y1 = best_edges[k + 1];
    }
  }
}

// Participant `rank`'s rectangle of an xsize x ysize frame, in pixels.
struct PixelRect {
  size_t x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  bool empty() const { return x1 <= x0 || y1 <= y0; }
  size_t width() const { return x1 - x0; }
  size_t height() const { return y1 - y0; }
};
PixelRect ShardRect(size_t xsize, size_t ysize, int world, int rank, bool rows_only = false) {
  const size_t xdc = (xsize + 2047) / 2048, ydc = (ysize + 2047) / 2048;
  DcRect all[kMaxWorld];
  ShardPartition(xdc, ydc, world, rows_only, all);
  PixelRect p;
  if (all[rank].empty()) return p;
  p.x0 = all[rank].x0 * 2048;
  p.y0 = all[rank].y0 * 2048;
  p.x1 = std::min(xsize, all[rank].x1 * 2048);
  p.y1 = std::min(ysize, all[rank].y1 * 2048);
  return p;
}

int NonEmptySlabs(size_t xsize, size_t ysize, int world, bool rows_only = false) {
  int n = 0;
  for (int r = 0; r < world; ++r) n += !ShardRect(xsize, ysize, world, r, rows_only).empty();
  return n;
}

void InitControl(Control* c, int world, size_t max_sections, size_t output_capacity, size_t region_bytes) {
  memset(static_cast<void*>(c), 0, sizeof(Control));
  for (auto& a : c->arrived) new (&a) std::atomic<uint64_t>(0);
  for (auto& a : c->published) new (&a) std::atomic<uint64_t>(0);
  new (&c->failed) std::atomic<int32_t>(0);
  new (&c->sleepers) std::atomic<uint32_t>(0);
  new (&c->attached) std::atomic<uint32_t>(0);
  c->world = static_cast<uint32_t>(world);
  c->max_sections = max_sections;
  c->output_offset = ControlBytes() + TablesBytes(max_sections);
  c->output_capacity = output_capacity;
  c->region_bytes = region_bytes;
  std::atomic_thread_fence(std::memory_order_seq_cst);
  c->magic = kMagic;
}

// The counters are waited for with a bounded spin (the usual case: the other participants are a fraction of a
// millisecond behind) and then a futex sleep on the counter's low word (the segment is shared between
// processes: no FUTEX_PRIVATE).  Whoever changes a counter wakes the sleepers, if there are any.
long Futex(void* addr, int op, uint32_t val, const struct timespec* timeout) {
  return syscall(SYS_futex, addr, op, val, timeout, nullptr, 0);
}
uint32_t* LowWord(std::atomic<uint64_t>* a) { return reinterpret_cast<uint32_t*>(a); }  // (little endian)
void Wake(Control* c, std::atomic<uint64_t>* a) {
  if (c->sleepers.load(std::memory_order_seq_cst) != 0) Futex(LowWord(a), FUTEX_WAKE, INT_MAX, nullptr);
}
void Arrive(Control* c, int what) {
  c->arrived[what].fetch_add(1, std::memory_order_seq_cst);
  Wake(c, &c->arrived[what]);
}
void Publish(Control* c, int what, uint64_t frame) {
  c->published[what].store(frame, std::memory_order_seq_cst);
  Wake(c, &c->published[what]);
}

// Waits until *a >= target; gives up when a participant has failed or after two minutes (a peer died).
int WaitAtLeast(jxlt_shard_group* g, std::atomic<uint64_t>* a, uint64_t target) {
  Control* c = g->ctl;
  const auto t0 = std::chrono::steady_clock::now();
  auto failed = [&]() -> int {
    const int32_t f = c->failed.load(std::memory_order_acquire);
    if (f != 0) {
      std::lock_guard<std::mutex> lock(g->error_mu);
      if (g->error.empty()) g->error = "another participant of the sharded frame failed";
    }
    return f;
  };
  // Spinning covers every wait INSIDE a frame (the participants are fractions of a millisecond apart, and a futex
  // wake-up costs 50-100 us of latency, four times per frame); a participant whose peers are far behind -- other
  // ranks still in a CPU-side leg, a dead peer -- goes to sleep after 2 ms instead of burning its core.
  for (uint64_t spins = 0;; ++spins) {
    if (a->load(std::memory_order_acquire) >= target) return JXLT_OK;
    if ((spins & 63) == 63) {
      if (const int f = failed()) return f;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  for (;;) {
    c->sleepers.fetch_add(1, std::memory_order_seq_cst);
    const uint64_t v = a->load(std::memory_order_seq_cst);
    if (v < target && c->failed.load(std::memory_order_acquire) == 0) {
      const struct timespec ts = {0, 5 * 1000 * 1000};  // (a lost wake-up costs 5 ms, not the frame)
      Futex(LowWord(a), FUTEX_WAIT, static_cast<uint32_t>(v), &ts);
    }
    c->sleepers.fetch_sub(1, std::memory_order_seq_cst);
    if (a->load(std::memory_order_acquire) >= target) return JXLT_OK;
    if (const int f = failed()) return f;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
      g->SetError("timed out waiting for the other participants of the sharded frame");
      int32_t expected = 0;
      c->failed.compare_exchange_strong(expected, JXLT_ERR_INTERNAL);
      for (auto& w : c->arrived) Futex(LowWord(&w), FUTEX_WAKE, INT_MAX, nullptr);
      for (auto& w : c->published) Futex(LowWord(&w), FUTEX_WAKE, INT_MAX, nullptr);
      return JXLT_ERR_INTERNAL;
    }
  }
}

int Fail(jxlt_shard_group* g, int rc, const char* what) {
  g->SetError(what);
  Control* c = g->ctl;
  int32_t expected = 0;
  c->failed.compare_exchange_strong(expected, rc);
  // (whoever sleeps on a counter that will never move now)
  for (auto& w : c->arrived) Futex(LowWord(&w), FUTEX_WAKE, INT_MAX, nullptr);
  for (auto& w : c->published) Futex(LowWord(&w), FUTEX_WAKE, INT_MAX, nullptr);
  return rc;
}

// The protocol, run by every participant with its own slab operations.
int EncodeShard(jxlt_shard_group* g, int rank, const jxlt_slab_ops* ops, size_t xsize, size_t ysize,
                float distance, const uint8_t** bytes, size_t* size, bool rows_only = false) {
  Control* c = g->ctl;
  const int world = g->world;
  if (bytes) *bytes = nullptr;
  if (size) *size = 0;
  if (c->failed.load() != 0) {
    g->SetError("the shard group is in a failed state (an earlier frame failed)");
    return c->failed.load();
  }
  const uint64_t frame = ++g->frame[rank];
  const uint64_t all = frame * static_cast<uint64_t>(world);
  const PixelRect rect = ShardRect(xsize, ysize, world, rank, rows_only);
  const bool empty = rect.empty();
  const size_t xdc = (xsize + 2047) / 2048, xgroups = (xsize + 255) / 256;
  const size_t ndc_frame = xdc * ((ysize + 2047) / 2048), ngroups_frame = xgroups * ((ysize + 255) / 256);
  // The slab's sections in the frame's raster order: its DC groups / AC groups are a rectangle of the frame's grid,
  // row `j` of it (sw_* sections wide) starts at frame index *_first + j * (row length of the frame).  The slab's own
  // section order (a frame of its own to the device) is the same rectangle in raster order.
  const size_t dc_first = empty ? 0 : xdc * (rect.y0 / 2048) + rect.x0 / 2048;
  const size_t ac_first = empty ? 0 : xgroups * (rect.y0 / 256) + rect.x0 / 256;
  const size_t sw_dc = empty ? 0 : (rect.width() + 2047) / 2048, sh_dc = empty ? 0 : (rect.height() + 2047) / 2048;
  const size_t sw_ac = empty ? 0 : (rect.width() + 255) / 256, sh_ac = empty ? 0 : (rect.height() + 255) / 256;
  const size_t ndc = sw_dc * sh_dc, nac = sw_ac * sh_ac;
  const auto dc_frame_index = [&](size_t i) { return dc_first + (i / sw_dc) * xdc + i % sw_dc; };
  const auto ac_frame_index = [&](size_t i) { return ac_first + (i / sw_ac) * xgroups + i % sw_ac; };
  if (ndc_frame + ngroups_frame > c->max_sections)
    return Fail(g, JXLT_ERR_OUT_OF_MEMORY, "frame has more sections than the shard group was opened for");
  if (ndc_frame + ngroups_frame == 2)
    return Fail(g, JXLT_ERR_UNSUPPORTED, "single-group frames are not sharded");

  const auto t_start = std::chrono::steady_clock::now();
  const bool keeps_timeline = g->rank >= 0 || rank == 0;
  const auto stamp = [&](int i) {
    if (keeps_timeline) g->timeline_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  };
  const DistanceParams distp = ComputeDistanceParams(distance);
  jxlt_params params;
  params.distance = distp.distance;
  params.scale = distp.scale;
  params.inv_scale = distp.inv_scale;
  params.scale_dc = distp.scale_dc;
  params.x_qm_scale = distp.x_qm_scale;
  params.flags = 0;
  int rc;
#define SLAB(call, what)                                   \
  if (!empty && (rc = (call)) != JXLT_OK) return Fail(g, rc, what)

  SLAB(ops->enqueue(ops->self, &params), "device pipeline failed");
  stamp(0);
  // participants 0 and 1 build the codes: their helper threads (entropy_coder.h) stop sleeping now and spin
  // for the histograms -- a wake-up would cost as much as half of a code construction
  // (only when this thread's last code construction did share its work with them: most histograms are clustered
  // faster alone -- entropy_coder.cc -- and helpers that are woken for nothing spin through the frame)
  static thread_local bool code_shared_last_time = true;
  if ((rank == 0 || rank == 1) && code_shared_last_time) WarmCodeConstruction(0.0, 8.0);

  // ---- both histograms leave for the meeting point, then the two codes are built IN PARALLEL by two
  // participants: 0 builds the DC code, 1 the AC code (each in its own process / thread, with whatever helper
  // threads its process has); the tables -- and the serialised ACGlobal section -- come back through the control
  // block.  (The DC histogram is complete before the AC tokenisation runs; fetching both before waiting for
  // either table keeps a participant's own tokenisation off the other code's critical path.)
  const uint32_t* h = nullptr;
  SLAB(ops->dc_histogram(ops->self, &h), "DC histogram fetch failed");
  if (empty) memset(c->hist[rank][1], 0, sizeof(c->hist[rank][1]));
  else memcpy(c->hist[rank][1], h, sizeof(c->hist[rank][1]));
  stamp(1);
  Arrive(c, kDcHist);
  SLAB(ops->ac_histogram(ops->self, &h), "AC histogram fetch failed");
  if (empty) memset(c->hist[rank][0], 0, sizeof(c->hist[rank][0]));
  else memcpy(c->hist[rank][0], h, sizeof(c->hist[rank][0]));
  stamp(2);
  Arrive(c, kAcHist);
  const int ac_builder = world > 1 ? 1 : 0;
  EntropyCode dc_code, ac_code;
  std::vector<uint32_t> sum(kHistWords);
  FrameGlobals globals;
  auto build_ac = [&]() -> int {
    const int rcw = WaitAtLeast(g, &c->arrived[kAcHist], all);
    if (rcw != JXLT_OK) return rcw;
    std::fill(sum.begin(), sum.end(), 0u);
    for (int r = 0; r < world; ++r)
      for (size_t i = 0; i < kHistWords; ++i) sum[i] += c->hist[r][0][i];
    (void)TakeClusteringShared();
    BuildAcCode(sum.data(), &ac_code);
    code_shared_last_time = TakeClusteringShared();
    FillCodeTable(ac_code, c->ac_table);
    const std::vector<uint8_t> acg = BuildAcGlobal(xsize, ysize, ac_code);
    if (acg.size() > sizeof(c->ac_global)) return Fail(g, JXLT_ERR_INTERNAL, "ACGlobal section larger than expected");
    memcpy(c->ac_global, acg.data(), acg.size());
    c->ac_global_size = static_cast<uint32_t>(acg.size());
    Publish(c, kAcTable, frame);
    return JXLT_OK;
  };
  if (rank == 0) {
    if ((rc = WaitAtLeast(g, &c->arrived[kDcHist], all)) != JXLT_OK) return rc;
    std::fill(sum.begin(), sum.end(), 0u);
    for (int r = 0; r < world; ++r)
      for (size_t i = 0; i < kHistWords; ++i) sum[i] += c->hist[r][1][i];
    (void)TakeClusteringShared();
    BuildDcCode(sum.data(), &dc_code);
    code_shared_last_time = TakeClusteringShared();
    FillCodeTable(dc_code, c->dc_table);
    Publish(c, kDcTable, frame);
    globals.dc_global = BuildDcGlobal(xsize, ysize, distp, dc_code);
  }
  if (rank == ac_builder && (rc = build_ac()) != JXLT_OK) return rc;
  if ((rc = WaitAtLeast(g, &c->published[kDcTable], frame)) != JXLT_OK) return rc;
  SLAB(ops->begin_dc_pack(ops->self, c->dc_table), "DC section measuring failed");
  if ((rc = WaitAtLeast(g, &c->published[kAcTable], frame)) != JXLT_OK) return rc;
  stamp(3);
  if (rank == 0) globals.ac_global.assign(c->ac_global, c->ac_global + c->ac_global_size);

  // ---- exact section sizes of every slab -> layout of the one output buffer
  jxlt_packed_sections dcm = {nullptr, nullptr, nullptr, 0}, acm = {nullptr, nullptr, nullptr, 0};
  SLAB(ops->measure(ops->self, c->ac_table, &dcm, &acm), "section measuring failed");
  if (!empty) {
    if (dcm.num_sections != ndc || acm.num_sections != nac)
      return Fail(g, JXLT_ERR_INTERNAL, "slab geometry does not match the frame (wrong number of rows set?)");
    uint32_t* bits = g->sec_bits();
    uint32_t* sizes = g->sec_bytes();
    for (size_t i = 0; i < ndc; ++i) {
      bits[dc_frame_index(i)] = dcm.section_bits[i];
      sizes[dc_frame_index(i)] = static_cast<uint32_t>(dcm.section_offset[i + 1] - dcm.section_offset[i]);
    }
    for (size_t i = 0; i < nac; ++i) {
      bits[ndc_frame + ac_frame_index(i)] = acm.section_bits[i];
      sizes[ndc_frame + ac_frame_index(i)] = static_cast<uint32_t>(acm.section_offset[i + 1] - acm.section_offset[i]);
    }
  }
  stamp(4);
  Arrive(c, kSizes);
  std::vector<uint64_t> dc_off, ac_off;
  std::vector<uint8_t> file_header;
  size_t dc_begin = 0, ac_global_at = 0, total_end = 0;
  if (rank == 0) {
    if ((rc = WaitAtLeast(g, &c->arrived[kSizes], all)) != JXLT_OK) return rc;
    jxl::BitWriter fh;
    if (!WriteFileHeader(xsize, ysize, &fh)) return Fail(g, JXLT_ERR_INVALID_ARGUMENT, "invalid frame size");
    file_header = fh.TakeBytes();
    const uint32_t* sizes = g->sec_bytes();
    dc_off.assign(ndc_frame + 1, 0);
    ac_off.assign(ngroups_frame + 1, 0);
    for (size_t i = 0; i < ndc_frame; ++i) dc_off[i + 1] = dc_off[i] + sizes[i];
    for (size_t i = 0; i < ngroups_frame; ++i) ac_off[i + 1] = ac_off[i] + sizes[ndc_frame + i];
    // head (file header + frame header + TOC + DCGlobal) is right-aligned in front of the DC sections: its
    // size is bounded before it exists, so the devices start copying at once
    dc_begin = (file_header.size() + HeadSizeBound(xsize, ysize, globals) + 255) & ~size_t(255);
    ac_global_at = dc_begin + dc_off[ndc_frame];
    const size_t ac_begin = ac_global_at + globals.ac_global.size();
    total_end = ac_begin + ac_off[ngroups_frame];
    if (total_end + 16 > c->output_capacity)
      return Fail(g, JXLT_ERR_OUT_OF_MEMORY, "codestream does not fit the shard group's output area");
    // every participant places its own sections: their offsets from the start of their kind, for all to read
    uint64_t* off = g->sec_off();
    for (size_t i = 0; i < ndc_frame; ++i) off[i] = dc_off[i];
    for (size_t i = 0; i < ngroups_frame; ++i) off[ndc_frame + i] = ac_off[i];
    c->dc_begin = dc_begin;
    c->ac_begin = ac_begin;
    Publish(c, kLayout, frame);
  } else if ((rc = WaitAtLeast(g, &c->published[kLayout], frame)) != JXLT_OK) {
    return rc;
  }

  stamp(5);
  // ---- every participant's device writes its sections in place: a run of the codestream per row of the slab's
  // DC groups / groups (rows that follow each other in the frame's order -- a slab as wide as the frame -- are one run)
  uint8_t* out = g->output();
  std::vector<jxlt_section_run> dc_runs, ac_runs;
  if (!empty) {
    const uint64_t* off = g->sec_off();
    const auto runs_of = [&](size_t sw, size_t sh, size_t first, size_t frame_row, uint64_t begin, const uint64_t* o,
                             std::vector<jxlt_section_run>* runs) {
      for (size_t j = 0; j < sh; ++j) {
        const size_t f = first + j * frame_row;  // frame index of the row's first section
        if (!runs->empty() && sw == frame_row) {
          runs->back().num_sections += static_cast<uint32_t>(sw);  // (continues the run before it)
          continue;
        }
        jxlt_section_run run;
        run.first_section = static_cast<uint32_t>(j * sw);
        run.num_sections = static_cast<uint32_t>(sw);
        run.dst_offset = begin + o[f];
        runs->push_back(run);
      }
    };
    runs_of(sw_dc, sh_dc, dc_first, xdc, c->dc_begin, off, &dc_runs);
    runs_of(sw_ac, sh_ac, ac_first, xgroups, c->ac_begin, off + ndc_frame, &ac_runs);
  }
  SLAB(ops->write(ops->self, out, dc_runs.data(), dc_runs.size(), ac_runs.data(), ac_runs.size()), "section placement failed");
  stamp(6);
  size_t frame_begin = 0;
  if (rank == 0) {
    const PackedSections dc = {nullptr, dc_off.data(), g->sec_bits(), ndc_frame};
    const PackedSections ac = {nullptr, ac_off.data(), g->sec_bits() + ndc_frame, ngroups_frame};
    std::vector<uint8_t> head;
    if (!BuildFrameHead(xsize, ysize, distp, globals, dc, ac, &head)) {
      if (!empty) ops->finish(ops->self);  // (the copies queued above still target the output area)
      return Fail(g, JXLT_ERR_INTERNAL, "frame head construction failed");
    }
    frame_begin = dc_begin - head.size() - file_header.size();
    memcpy(out + frame_begin, file_header.data(), file_header.size());
    memcpy(out + frame_begin + file_header.size(), head.data(), head.size());
    memcpy(out + ac_global_at, globals.ac_global.data(), globals.ac_global.size());
  }
  SLAB(ops->finish(ops->self), "device synchronisation failed");
#undef SLAB
  Arrive(c, kPlaced);
  if (rank == 0) {
    if ((rc = WaitAtLeast(g, &c->arrived[kPlaced], all)) != JXLT_OK) return rc;
    stamp(7);
    if (bytes) *bytes = out + frame_begin;
    if (size) *size = total_end - frame_begin;
    // every process has mapped the segment by now (it has taken part in this frame): the name can go, so that a
    // rank that dies later leaves nothing behind in /dev/shm
    if (g->shm && !g->unlinked && c->attached.load(std::memory_order_acquire) >= static_cast<uint32_t>(world)) {
      shm_unlink(g->name.c_str());
      g->unlinked = true;
    }
  }
  return JXLT_OK;
}

// ---- slab operations bound to a device context
int CtxEnqueue(void* self, const jxlt_params* p) {
  jxlt_context* ctx = static_cast<jxlt_context*>(self);
  ApplyStrategyDistanceEmulation(ctx, p->distance);
  return jxlt_encode_enqueue(ctx, p);
}
int CtxDcHist(void* self, const uint32_t** h) { return jxlt_fetch_dc_histogram(static_cast<jxlt_context*>(self), h); }
int CtxBeginDc(void* self, const uint32_t* t) { return jxlt_pack_begin(static_cast<jxlt_context*>(self), 0, t); }
int CtxAcHist(void* self, const uint32_t** h) {
  return jxlt_fetch_histograms(static_cast<jxlt_context*>(self), h, nullptr);
}
int CtxMeasure(void* self, const uint32_t* t, jxlt_packed_sections* dc, jxlt_packed_sections* ac) {
  jxlt_context* ctx = static_cast<jxlt_context*>(self);
  int rc = jxlt_pack_begin(ctx, 1, t);
  if (rc == JXLT_OK) rc = jxlt_pack_sizes(ctx, 0, dc);
  if (rc == JXLT_OK) rc = jxlt_pack_sizes(ctx, 1, ac);
  return rc;
}
int CtxWrite(void* self, uint8_t* out, const jxlt_section_run* dc_runs, size_t n_dc, const jxlt_section_run* ac_runs,
             size_t n_ac) {
  jxlt_context* ctx = static_cast<jxlt_context*>(self);
  const int rc = jxlt_pack_deliver(ctx, 0, out, dc_runs, n_dc, 0);
  return rc != JXLT_OK ? rc : jxlt_pack_deliver(ctx, 1, out, ac_runs, n_ac, 0);
}
int CtxFinish(void* self) { return jxlt_synchronize(static_cast<jxlt_context*>(self)); }

jxlt_slab_ops OpsOf(jxlt_context* ctx) {
  return {ctx, CtxEnqueue, CtxDcHist, CtxBeginDc, CtxAcHist, CtxMeasure, CtxWrite, CtxFinish};
}

size_t SectionsOf(size_t xsize, size_t ysize) {
  return ((xsize + 2047) / 2048) * ((ysize + 2047) / 2048) + ((xsize + 255) / 256) * ((ysize + 255) / 256);
}

}  // namespace
}  // namespace jxlt

// ---------------------------------------------------------------------------------------------
// In-process form: one context + one host thread per device.
struct jxlt_multi_encoder {
  std::vector<int> devices;
  std::vector<jxlt_context*> ctx;
  jxlt_shard_group group;          // heap region, re-made when a frame needs more room
  std::string error;
  // persistent workers (participants 1 .. n-1; the caller's thread is participant 0)
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  uint64_t job_id = 0;
  int pending = 0;
  bool quit = false;
  // the job
  enum Source { kNone, kHostPlanes, kHostPfm, kResident } source = kNone;
  const float* planes[3] = {nullptr, nullptr, nullptr};
  size_t pitch_bytes = 0;
  const uint8_t* pfm = nullptr;
  int pfm_big_endian = 0;
  size_t xsize = 0, ysize = 0;
  float distance = 1.0f;
  std::vector<int> status;
  const uint8_t* out_bytes = nullptr;
  size_t out_size = 0;
  // frames set slab by slab (jxlt_multi_encoder_set_device_slab)
  std::vector<size_t> slab_rows, slab_cols;
};

namespace jxlt {
namespace {

int EnsureLocalRegion(jxlt_multi_encoder* enc, size_t xsize, size_t ysize) {
  jxlt_shard_group& g = enc->group;
  const size_t sections = SectionsOf(xsize, ysize);
  // worst case of the device packer: 28 bits per record, 9.14 B/pixel of records -> bounded by the raw
  // frame; in practice a few percent of it.  Start with 1 byte per pixel + slack and grow on demand.
  size_t want_out = g.ctl ? static_cast<size_t>(g.ctl->output_capacity) : 0;
  const size_t floor_out = xsize * ysize + (size_t(1) << 20);
  if (want_out < floor_out) want_out = floor_out;
  if (g.ctl && g.ctl->max_sections >= sections && g.ctl->output_capacity >= want_out) return JXLT_OK;
  if (g.base) jxlt_pinned_free(g.base);
  g.base = nullptr;
  g.ctl = nullptr;
  const size_t max_sections = sections + sections / 4 + 64;
  const size_t bytes = ControlBytes() + TablesBytes(max_sections) + want_out;
  g.base = static_cast<uint8_t*>(jxlt_pinned_alloc(bytes));  // page-locked, visible to every device
  if (!g.base) {
    enc->error = "cannot allocate the page-locked output region (no usable HIP device?)";
    return JXLT_ERR_NO_DEVICE;
  }
  g.bytes = bytes;
  g.ctl = reinterpret_cast<Control*>(g.base);
  InitControl(g.ctl, g.world, max_sections, want_out, bytes);
  for (auto& f : g.frame) f = 0;
  return JXLT_OK;
}

// What participant `rank` does for the current job.
int RunParticipant(jxlt_multi_encoder* enc, int rank, const uint8_t** bytes, size_t* size) {
  jxlt_context* ctx = enc->ctx[rank];
  // (a PFM payload is cut along its rows only: they are contiguous in the file, its columns are not)
  const bool rows_only = enc->source == jxlt_multi_encoder::kHostPfm;
  const PixelRect rect = ShardRect(enc->xsize, enc->ysize, enc->group.world, rank, rows_only);
  int rc = JXLT_OK;
  if (!rect.empty()) {
    if (enc->source == jxlt_multi_encoder::kHostPlanes) {
      // the slab is a rectangle of the caller's planes: same pitch, moved base
      const float* slab[3];
      for (int c = 0; c < 3; ++c)
        slab[c] = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(enc->planes[c]) + rect.y0 * enc->pitch_bytes) + rect.x0;
      // page-locked memory: the upload is pipelined under the slab's kernels; anything else is staged
      rc = jxlt_image_attach_host(ctx, slab, enc->pitch_bytes, rect.width(), rect.height());
      if (rc != JXLT_OK) rc = jxlt_image_upload(ctx, slab, enc->pitch_bytes, rect.width(), rect.height());
    } else if (enc->source == jxlt_multi_encoder::kHostPfm) {
      // bottom-up payload: rows [y0, y1) from the top are the payload rows [ysize - y1, ysize - y0)
      const uint8_t* slab = enc->pfm + (enc->ysize - rect.y1) * enc->xsize * 3 * sizeof(float);
      rc = jxlt_image_attach_host_pfm(ctx, slab, enc->xsize, rect.height(), enc->pfm_big_endian);
      if (rc != JXLT_OK) rc = jxlt_image_upload_pfm(ctx, slab, enc->xsize, rect.height(), enc->pfm_big_endian);
    }
    if (rc != JXLT_OK) {
      enc->group.SetError(std::string("slab upload failed: ") + jxlt_last_error(ctx));
      int32_t expected = 0;
      enc->group.ctl->failed.compare_exchange_strong(expected, rc);
      return rc;
    }
  }
  const jxlt_slab_ops ops = OpsOf(ctx);
  return EncodeShard(&enc->group, rank, &ops, enc->xsize, enc->ysize, enc->distance, bytes, size, rows_only);
}

void WorkerLoop(jxlt_multi_encoder* enc, int rank) {
  uint64_t seen = 0;
  for (;;) {
    {
      std::unique_lock<std::mutex> lock(enc->mu);
      enc->cv_go.wait(lock, [&] { return enc->quit || enc->job_id != seen; });
      if (enc->quit) return;
      seen = enc->job_id;
    }
    const int rc = RunParticipant(enc, rank, nullptr, nullptr);
    {
      std::lock_guard<std::mutex> lock(enc->mu);
      enc->status[rank] = rc;
      if (--enc->pending == 0) enc->cv_done.notify_all();
    }
  }
}

int RunJob(jxlt_multi_encoder* enc, const uint8_t** bytes, size_t* size) {
  const int world = enc->group.world;
  *bytes = nullptr;
  *size = 0;
  float d = enc->distance;
  if (!NormalizeDistance(&d)) {
    enc->error = "invalid distance";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  enc->distance = d;
  if (enc->xsize == 0 || enc->ysize == 0) {
    enc->error = "empty frame";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  if (NonEmptySlabs(enc->xsize, enc->ysize, world, enc->source == jxlt_multi_encoder::kHostPfm) <= 1) {
    // one DC group (or one row of DC groups of a PFM payload): nothing to shard -- the ordinary path on participant
    // 0's device, which owns the only slab (ShardPartition)
    if (enc->source == jxlt_multi_encoder::kResident) {
      // (the caller has set the whole frame as slab 0)
    }
    jxlt_context* ctx = enc->ctx[0];
    int rc = JXLT_OK;
    if (enc->source == jxlt_multi_encoder::kHostPlanes)
      rc = jxlt_image_upload(ctx, enc->planes, enc->pitch_bytes, enc->xsize, enc->ysize);
    else if (enc->source == jxlt_multi_encoder::kHostPfm)
      rc = jxlt_image_upload_pfm(ctx, enc->pfm, enc->xsize, enc->ysize, enc->pfm_big_endian);
    if (rc == JXLT_OK) rc = jxlt_encode_resident_view(ctx, enc->distance, 0, bytes, size);
    if (rc != JXLT_OK) enc->error = std::string("single-device encode failed: ") + jxlt_last_error(ctx);
    return rc;
  }
  int rc = EnsureLocalRegion(enc, enc->xsize, enc->ysize);
  if (rc != JXLT_OK) return rc;
  for (int attempt = 0;; ++attempt) {
    enc->group.ctl->failed.store(0);
    {
      std::lock_guard<std::mutex> lock(enc->mu);
      enc->status.assign(world, JXLT_OK);
      enc->pending = world - 1;
      ++enc->job_id;
    }
    enc->cv_go.notify_all();
    const int rc0 = RunParticipant(enc, 0, bytes, size);
    {
      std::unique_lock<std::mutex> lock(enc->mu);
      enc->cv_done.wait(lock, [&] { return enc->pending == 0; });
    }
    rc = rc0;
    for (int r = 1; r < world && rc == JXLT_OK; ++r) rc = enc->status[r];
    if (rc == JXLT_ERR_OUT_OF_MEMORY && attempt == 0 &&
        enc->group.error.find("does not fit") != std::string::npos) {
      // incompressible content: give the output area the packer's worst case and redo the frame
      // (every participant is past its last wait: the failure flag released them)
      Control* c = enc->group.ctl;
      const size_t worst = 10 * enc->xsize * enc->ysize + (size_t(1) << 20);
      const size_t max_sections = static_cast<size_t>(c->max_sections);
      jxlt_pinned_free(enc->group.base);
      enc->group.base = nullptr;
      enc->group.ctl = nullptr;
      const size_t bytes_needed = ControlBytes() + TablesBytes(max_sections) + worst;
      enc->group.base = static_cast<uint8_t*>(jxlt_pinned_alloc(bytes_needed));
      if (!enc->group.base) {
        enc->error = "cannot allocate the page-locked output region";
        return JXLT_ERR_OUT_OF_MEMORY;
      }
      enc->group.bytes = bytes_needed;
      enc->group.ctl = reinterpret_cast<Control*>(enc->group.base);
      InitControl(enc->group.ctl, world, max_sections, worst, bytes_needed);
      for (auto& f : enc->group.frame) f = 0;
      continue;
    }
    break;
  }
  if (rc != JXLT_OK) {
    enc->error = enc->group.error.empty() ? "sharded encode failed" : enc->group.error;
    // a failed frame leaves the counters of the control block out of step: start over next time
    InitControl(enc->group.ctl, world, static_cast<size_t>(enc->group.ctl->max_sections),
                static_cast<size_t>(enc->group.ctl->output_capacity), enc->group.bytes);
    for (auto& f : enc->group.frame) f = 0;
  }
  return rc;
}

}  // namespace
}  // namespace jxlt

extern "C" {

int jxlt_shard_rect(size_t xsize, size_t ysize, int world, int rank, size_t* x0, size_t* y0, size_t* x1, size_t* y1) {
  if (!x0 || !y0 || !x1 || !y1 || xsize == 0 || ysize == 0 || world < 1 || world > jxlt::kMaxWorld || rank < 0 ||
      rank >= world)
    return JXLT_ERR_INVALID_ARGUMENT;
  const jxlt::PixelRect r = jxlt::ShardRect(xsize, ysize, world, rank);
  *x0 = r.x0;
  *y0 = r.y0;
  *x1 = r.x1;
  *y1 = r.y1;
  return JXLT_OK;
}

int jxlt_multi_encoder_create(const int* device_ordinals, int num_devices, jxlt_multi_encoder** out) {
  if (!out) return JXLT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!device_ordinals || num_devices < 1 || num_devices > jxlt::kMaxWorld) return JXLT_ERR_INVALID_ARGUMENT;
  jxlt_multi_encoder* enc = new jxlt_multi_encoder;
  enc->group.rank = -1;
  enc->group.world = num_devices;
  for (int i = 0; i < num_devices; ++i) {
    jxlt_context* ctx = nullptr;
    const int rc = jxlt_context_create(device_ordinals[i], &ctx);
    if (rc != JXLT_OK) {  // no device: there is no CPU fallback
      for (jxlt_context* c : enc->ctx) jxlt_context_destroy(c);
      delete enc;
      return rc;
    }
    enc->devices.push_back(device_ordinals[i]);
    enc->ctx.push_back(ctx);
  }
  enc->slab_rows.assign(num_devices, 0);
  enc->slab_cols.assign(num_devices, 0);
  for (int r = 1; r < num_devices; ++r) enc->workers.emplace_back(jxlt::WorkerLoop, enc, r);
  *out = enc;
  return JXLT_OK;
}

void jxlt_multi_encoder_destroy(jxlt_multi_encoder* enc) {
  if (!enc) return;
  {
    std::lock_guard<std::mutex> lock(enc->mu);
    enc->quit = true;
  }
  enc->cv_go.notify_all();
  for (std::thread& t : enc->workers) t.join();
  for (jxlt_context* c : enc->ctx) jxlt_context_destroy(c);
  if (enc->group.base) jxlt_pinned_free(enc->group.base);
  delete enc;
}

const char* jxlt_multi_encoder_last_error(const jxlt_multi_encoder* enc) { return enc ? enc->error.c_str() : ""; }

int jxlt_multi_encoder_encode(jxlt_multi_encoder* enc, const float* const planes[3], size_t pitch_bytes,
                              size_t xsize, size_t ysize, float distance, const uint8_t** bytes, size_t* size) {
  if (!enc || !planes || !planes[0] || !planes[1] || !planes[2] || !bytes || !size ||
      pitch_bytes < xsize * sizeof(float) || pitch_bytes % sizeof(float))
    return JXLT_ERR_INVALID_ARGUMENT;
  enc->source = jxlt_multi_encoder::kHostPlanes;
  for (int c = 0; c < 3; ++c) enc->planes[c] = planes[c];
  enc->pitch_bytes = pitch_bytes;
  enc->xsize = xsize;
  enc->ysize = ysize;
  enc->distance = distance;
  return jxlt::RunJob(enc, bytes, size);
}

int jxlt_multi_encoder_encode_pfm(jxlt_multi_encoder* enc, const void* host_payload, size_t xsize, size_t ysize,
                                  int big_endian, float distance, const uint8_t** bytes, size_t* size) {
  if (!enc || !host_payload || !bytes || !size) return JXLT_ERR_INVALID_ARGUMENT;
  enc->source = jxlt_multi_encoder::kHostPfm;
  enc->pfm = static_cast<const uint8_t*>(host_payload);
  enc->pfm_big_endian = big_endian;
  enc->xsize = xsize;
  enc->ysize = ysize;
  enc->distance = distance;
  return jxlt::RunJob(enc, bytes, size);
}

int jxlt_multi_encoder_set_device_slab(jxlt_multi_encoder* enc, int slab, const void* const device_planes[3],
                                       size_t pitch_bytes, size_t xsize, size_t rows) {
  if (!enc || slab < 0 || slab >= enc->group.world) return JXLT_ERR_INVALID_ARGUMENT;
  const int rc = jxlt_image_set_device(enc->ctx[slab], device_planes, pitch_bytes, xsize, rows);
  if (rc != JXLT_OK) {
    enc->error = jxlt_last_error(enc->ctx[slab]);
    return rc;
  }
  enc->slab_rows[slab] = rows;
  enc->slab_cols[slab] = xsize;
  return JXLT_OK;
}

int jxlt_multi_encoder_encode_resident(jxlt_multi_encoder* enc, size_t xsize, size_t ysize, float distance,
                                       const uint8_t** bytes, size_t* size) {
  if (!enc || !bytes || !size) return JXLT_ERR_INVALID_ARGUMENT;
  for (int r = 0; r < enc->group.world; ++r) {
    const jxlt::PixelRect rect = jxlt::ShardRect(xsize, ysize, enc->group.world, r);
    if (!rect.empty() && (enc->slab_rows[r] != rect.height() || enc->slab_cols[r] != rect.width())) {
      enc->error = "slab " + std::to_string(r) + " was not set with the rectangle jxlt_shard_rect gives for this frame";
      return JXLT_ERR_INVALID_ARGUMENT;
    }
  }
  enc->source = jxlt_multi_encoder::kResident;
  enc->xsize = xsize;
  enc->ysize = ysize;
  enc->distance = distance;
  return jxlt::RunJob(enc, bytes, size);
}

// ---------------------------------------------------------------------------------------------
// One process per GPU: the region is a POSIX shared-memory segment.
int jxlt_shard_group_open(const char* shm_name, int rank, int world, size_t output_capacity, size_t max_sections,
                          jxlt_shard_group** out) {
  if (!out) return JXLT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!shm_name || shm_name[0] != '/' || world < 1 || world > jxlt::kMaxWorld || rank < 0 || rank >= world ||
      output_capacity == 0 || max_sections == 0)
    return JXLT_ERR_INVALID_ARGUMENT;
  size_t bytes = jxlt::ControlBytes() + jxlt::TablesBytes(max_sections) + ((output_capacity + 4095) & ~size_t(4095));
  int fd;
  if (rank == 0) {
    shm_unlink(shm_name);  // a stale segment of a run that died
    fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, static_cast<off_t>(bytes)) != 0) {
      if (fd >= 0) close(fd);
      return JXLT_ERR_OUT_OF_MEMORY;
    }
  } else {
    // the segment is what rank 0 made it (its capacity arguments rule; this rank's are not consulted)
    fd = shm_open(shm_name, O_RDWR, 0600);
    if (fd < 0) return JXLT_ERR_INVALID_ARGUMENT;
    struct stat st;
    if (fstat(fd, &st) != 0 || static_cast<size_t>(st.st_size) < jxlt::ControlBytes()) {
      close(fd);
      return JXLT_ERR_INVALID_ARGUMENT;
    }
    bytes = static_cast<size_t>(st.st_size);
  }
  void* map = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (map == MAP_FAILED) {
    if (rank == 0) shm_unlink(shm_name);
    return JXLT_ERR_OUT_OF_MEMORY;
  }
  jxlt_shard_group* g = new jxlt_shard_group;
  g->base = static_cast<uint8_t*>(map);
  g->bytes = bytes;
  g->ctl = reinterpret_cast<jxlt::Control*>(g->base);
  g->rank = rank;
  g->world = world;
  g->shm = true;
  g->name = shm_name;
  if (rank == 0) {
    jxlt::InitControl(g->ctl, world, max_sections, (output_capacity + 4095) & ~size_t(4095), bytes);
  } else if (g->ctl->magic != jxlt::kMagic || g->ctl->world != static_cast<uint32_t>(world) ||
             g->ctl->region_bytes != bytes) {
    munmap(map, bytes);
    delete g;
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  g->ctl->attached.fetch_add(1, std::memory_order_acq_rel);
  *out = g;
  return JXLT_OK;
}

void jxlt_shard_group_close(jxlt_shard_group* g) {
  if (!g) return;
  if (g->registered) jxlt_pinned_unregister(g->output());
  if (g->shm) {
    munmap(g->base, g->bytes);
    if (g->rank == 0 && !g->unlinked) shm_unlink(g->name.c_str());
  }
  delete g;
}

const char* jxlt_shard_group_last_error(const jxlt_shard_group* g) { return g ? g->error.c_str() : ""; }
int jxlt_shard_group_last_timeline(const jxlt_shard_group* g, double* ms8) {
  if (!g || !ms8) return JXLT_ERR_INVALID_ARGUMENT;
  memcpy(ms8, g->timeline_ms, sizeof(g->timeline_ms));
  return JXLT_OK;
}

int jxlt_shard_encode_ops(jxlt_shard_group* g, const jxlt_slab_ops* ops, size_t xsize, size_t ysize, float distance,
                          const uint8_t** bytes, size_t* size) {
  if (!g || !ops || g->rank < 0 || !ops->enqueue || !ops->dc_histogram || !ops->begin_dc_pack || !ops->ac_histogram ||
      !ops->measure || !ops->write || !ops->finish)
    return JXLT_ERR_INVALID_ARGUMENT;
  if (!jxlt::NormalizeDistance(&distance) || xsize == 0 || ysize == 0) return JXLT_ERR_INVALID_ARGUMENT;
  return jxlt::EncodeShard(g, g->rank, ops, xsize, ysize, distance, bytes, size);
}

int jxlt_shard_encode(jxlt_shard_group* g, jxlt_context* ctx, size_t xsize, size_t ysize, float distance,
                      const uint8_t** bytes, size_t* size) {
  if (!g || !ctx || g->rank < 0) return JXLT_ERR_INVALID_ARGUMENT;
  if (bytes) *bytes = nullptr;
  if (size) *size = 0;
  if (!jxlt::NormalizeDistance(&distance) || xsize == 0 || ysize == 0) return JXLT_ERR_INVALID_ARGUMENT;
  if (jxlt::NonEmptySlabs(xsize, ysize, g->world) <= 1) {
    // nothing to shard: rank 0 owns the only slab (jxlt_shard_rect) and encodes it the ordinary way; the
    // other ranks have nothing to do and the control block is not involved
    if (g->rank != 0) return JXLT_OK;
    const int rc = jxlt_encode_resident_view(ctx, distance, 0, bytes, size);
    if (rc != JXLT_OK) g->SetError(std::string("single-device encode failed: ") + jxlt_last_error(ctx));
    return rc;
  }
  // The host side of the frame (hand-overs, code construction) next to the GPU -- for the duration of this call only:
  // the caller's own affinity mask is put back before it returns (a library call must not re-pin the
  // application's thread for good).  Threads the library creates from inside the call -- the code construction's
  // worker and helper threads -- inherit the bound mask and keep it: they are the library's own.
  struct ScopedNearDevice {
    cpu_set_t saved;
    bool restore = false;
    explicit ScopedNearDevice(int device) {
      static const bool no_affinity = getenv("JXLT_NO_AFFINITY") != nullptr;
      if (no_affinity || sched_getaffinity(0, sizeof(saved), &saved) != 0) return;
      restore = jxlt_bind_thread_near_device(device) == JXLT_OK;
    }
    ~ScopedNearDevice() {
      if (restore) (void)sched_setaffinity(0, sizeof(saved), &saved);
    }
  } near_device(jxlt_context_device(ctx));
  if (!g->registered) {
    // the devices copy their sections straight into the segment: page-lock this process's mapping of it
    if (jxlt_pinned_register(g->output(), static_cast<size_t>(g->ctl->output_capacity)) != JXLT_OK)
      return jxlt::Fail(g, JXLT_ERR_NO_DEVICE, "cannot page-lock the shared output area");
    g->registered = true;
  }
  const jxlt_slab_ops ops = jxlt::OpsOf(ctx);
  return jxlt::EncodeShard(g, g->rank, &ops, xsize, ysize, distance, bytes, size);
}

// ---------------------------------------------------------------------------------------------
// Frames in flight over the group: `depth` lanes, each a shard group of its own (segment <name>.<lane>), a device
// context and a host thread.  Frame k runs on lane k % depth, so while frame k is in its code construction and
// section packing -- the part of a sharded frame that does not shrink with the number of GPUs -- the kernels of
// frame k + 1 already run on every GPU.
struct jxlt_shard_pipeline {
  struct Lane {
    jxlt_shard_group* group = nullptr;
    jxlt_context* ctx = nullptr;
    jxlt_slab_ops ops = {};   // testing form (ops.enqueue != nullptr): instead of ctx
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, busy = false, quit = false;
    uint64_t ticket = 0;      // of the job / of the result
    const void* planes[3] = {nullptr, nullptr, nullptr};
    size_t pitch_bytes = 0, xsize = 0, ysize = 0, rows = 0;
    float distance = 1.0f;
    int rc = JXLT_OK;
    const uint8_t* bytes = nullptr;
    size_t size = 0;
    std::string error;
  };
  std::vector<Lane*> lanes;
  int rank = 0, world = 1;
  uint64_t next_ticket = 0;
  std::string error;
};

namespace jxlt {
namespace {
void LaneLoop(jxlt_shard_pipeline::Lane* lane) {
  // (a lane's thread is the library's own: it lives next to its GPU for good)
  if (lane->ctx && getenv("JXLT_NO_AFFINITY") == nullptr) (void)jxlt_bind_thread_near_device(jxlt_context_device(lane->ctx));
  for (;;) {
    {
      std::unique_lock<std::mutex> lock(lane->mu);
      lane->cv.wait(lock, [&] { return lane->quit || lane->has_job; });
      // (a frame that has been submitted is encoded before the lane goes: the other ranks have started it, and
      // would wait for this rank's part until their time-out -- ADVICE r3)
      if (!lane->has_job) return;
      lane->has_job = false;
    }
    int rc = JXLT_OK;
    const uint8_t* bytes = nullptr;
    size_t size = 0;
    if (lane->ops.enqueue != nullptr) {
      rc = jxlt_shard_encode_ops(lane->group, &lane->ops, lane->xsize, lane->ysize, lane->distance, &bytes, &size);
    } else {
      if (lane->rows != 0) {
        // the rank's slab: the rectangle jxlt_shard_rect gives it, starting at the planes' first sample
        const PixelRect rect = ShardRect(lane->xsize, lane->ysize, lane->group->world, lane->group->rank);
        if (rect.empty() || rect.height() != lane->rows) {
          rc = JXLT_ERR_INVALID_ARGUMENT;
          lane->error = "slab_rows is not the height of the rectangle jxlt_shard_rect gives this rank";
        } else {
          rc = jxlt_image_set_device(lane->ctx, lane->planes, lane->pitch_bytes, rect.width(), rect.height());
          if (rc != JXLT_OK) lane->error = std::string("slab set-up failed: ") + jxlt_last_error(lane->ctx);
        }
      }
      if (rc != JXLT_OK) {
        // (the other ranks must not wait for this one's part of the frame)
        Fail(lane->group, rc, lane->error.c_str());
      } else {
        rc = jxlt_shard_encode(lane->group, lane->ctx, lane->xsize, lane->ysize, lane->distance, &bytes, &size);
      }
    }
    if (rc != JXLT_OK && lane->error.empty()) lane->error = jxlt_shard_group_last_error(lane->group);
    {
      std::lock_guard<std::mutex> lock(lane->mu);
      lane->rc = rc;
      lane->bytes = bytes;
      lane->size = size;
      lane->busy = false;
    }
    lane->cv.notify_all();
  }
}

int OpenPipeline(const char* shm_name, int rank, int world, int device_ordinal, const jxlt_slab_ops* lane_ops, int depth,
                 size_t output_capacity, size_t max_sections, jxlt_shard_pipeline** out) {
  if (!out) return JXLT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!shm_name || depth < 1 || depth > 8) return JXLT_ERR_INVALID_ARGUMENT;
  jxlt_shard_pipeline* p = new jxlt_shard_pipeline;
  p->rank = rank;
  p->world = world;
  int rc = JXLT_OK;
  for (int l = 0; l < depth && rc == JXLT_OK; ++l) {
    jxlt_shard_pipeline::Lane* lane = new jxlt_shard_pipeline::Lane;
    p->lanes.push_back(lane);
    const std::string name = std::string(shm_name) + "." + std::to_string(l);
    rc = jxlt_shard_group_open(name.c_str(), rank, world, output_capacity, max_sections, &lane->group);
    if (rc == JXLT_OK) {
      if (lane_ops) lane->ops = lane_ops[l];
      else rc = jxlt_context_create(device_ordinal, &lane->ctx);
    }
  }
  if (rc != JXLT_OK) {
    for (jxlt_shard_pipeline::Lane* lane : p->lanes) {
      if (lane->ctx) jxlt_context_destroy(lane->ctx);
      jxlt_shard_group_close(lane->group);
      delete lane;
    }
    delete p;
    return rc;
  }
  for (jxlt_shard_pipeline::Lane* lane : p->lanes) lane->thread = std::thread(LaneLoop, lane);
  *out = p;
  return JXLT_OK;
}

int Submit(jxlt_shard_pipeline* p, const void* const planes[3], size_t pitch_bytes, size_t xsize, size_t ysize,
           size_t rows, float distance, uint64_t* ticket) {
  const uint64_t t = p->next_ticket++;
  jxlt_shard_pipeline::Lane* lane = p->lanes[t % p->lanes.size()];
  {
    std::unique_lock<std::mutex> lock(lane->mu);
    lane->cv.wait(lock, [&] { return !lane->busy; });  // (the lane's previous frame, depth frames back)
    for (int c = 0; c < 3; ++c) lane->planes[c] = planes ? planes[c] : nullptr;
    lane->pitch_bytes = pitch_bytes;
    lane->xsize = xsize;
    lane->ysize = ysize;
    lane->rows = rows;
    lane->distance = distance;
    lane->ticket = t;
    lane->error.clear();
    lane->busy = true;
    lane->has_job = true;
  }
  lane->cv.notify_all();
  if (ticket) *ticket = t;
  return JXLT_OK;
}
}  // namespace
}  // namespace jxlt

int jxlt_shard_pipeline_open(const char* shm_name, int rank, int world, int device_ordinal, int depth,
                             size_t output_capacity, size_t max_sections, jxlt_shard_pipeline** out) {
  return jxlt::OpenPipeline(shm_name, rank, world, device_ordinal, nullptr, depth, output_capacity, max_sections, out);
}

int jxlt_shard_pipeline_open_ops(const char* shm_name, int rank, int world, const jxlt_slab_ops* lane_ops, int depth,
                                 size_t output_capacity, size_t max_sections, jxlt_shard_pipeline** out) {
  if (!lane_ops) return JXLT_ERR_INVALID_ARGUMENT;
  return jxlt::OpenPipeline(shm_name, rank, world, -1, lane_ops, depth, output_capacity, max_sections, out);
}

void jxlt_shard_pipeline_close(jxlt_shard_pipeline* p) {
  if (!p) return;
  for (jxlt_shard_pipeline::Lane* lane : p->lanes) {
    {
      std::lock_guard<std::mutex> lock(lane->mu);
      lane->quit = true;
    }
    lane->cv.notify_all();
    if (lane->thread.joinable()) lane->thread.join();
    if (lane->ctx) jxlt_context_destroy(lane->ctx);
    jxlt_shard_group_close(lane->group);
    delete lane;
  }
  delete p;
}

const char* jxlt_shard_pipeline_last_error(const jxlt_shard_pipeline* p) { return p ? p->error.c_str() : ""; }

int jxlt_shard_pipeline_submit_device(jxlt_shard_pipeline* p, const void* const device_planes[3], size_t pitch_bytes,
                                      size_t xsize, size_t ysize, size_t slab_rows, float distance, uint64_t* ticket) {
  if (!p || xsize == 0 || ysize == 0 || (slab_rows != 0 && (!device_planes || !device_planes[0])))
    return JXLT_ERR_INVALID_ARGUMENT;
  if (p->lanes[0]->ctx == nullptr) return JXLT_ERR_INVALID_ARGUMENT;  // (opened over slab operations)
  return jxlt::Submit(p, device_planes, pitch_bytes, xsize, ysize, slab_rows, distance, ticket);
}

int jxlt_shard_pipeline_submit_ops(jxlt_shard_pipeline* p, size_t xsize, size_t ysize, float distance, uint64_t* ticket) {
  if (!p || xsize == 0 || ysize == 0 || p->lanes[0]->ops.enqueue == nullptr) return JXLT_ERR_INVALID_ARGUMENT;
  return jxlt::Submit(p, nullptr, 0, xsize, ysize, 0, distance, ticket);
}

int jxlt_shard_pipeline_wait(jxlt_shard_pipeline* p, uint64_t ticket, const uint8_t** bytes, size_t* size) {
  if (!p || ticket >= p->next_ticket) return JXLT_ERR_INVALID_ARGUMENT;
  if (bytes) *bytes = nullptr;
  if (size) *size = 0;
  jxlt_shard_pipeline::Lane* lane = p->lanes[ticket % p->lanes.size()];
  std::unique_lock<std::mutex> lock(lane->mu);
  if (lane->ticket != ticket) {
    p->error = "the frame's result is gone: its lane has been given a later frame";
    return JXLT_ERR_INVALID_ARGUMENT;
  }
  lane->cv.wait(lock, [&] { return !lane->busy; });
  if (lane->rc != JXLT_OK) p->error = lane->error;
  if (bytes) *bytes = lane->bytes;
  if (size) *size = lane->size;
  return lane->rc;
}

}  // extern "C"
