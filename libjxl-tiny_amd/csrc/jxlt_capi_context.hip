// jxlt_capi_context.hip -- libjxltiny_hip.so (include/jxl_tiny_amd.h): contexts, streams, device and page-locked
// memory, frames in / onto the device, the output buffer.  There is no CPU fallback: without a usable HIP device
// every entry point returns JXLT_ERR_NO_DEVICE.
#include "jxlt_context.h"

using namespace jxlt_dev;
using namespace jxlt_host;

namespace {

thread_local std::string g_create_error;

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
// The arithmetic the kernels rely on beyond IEEE operations: FP32 DENORMALS ARE KEPT.  tile12_kernel takes the byte
// offset of a quantised magnitude's square root (and of its zeros' cost) as the bit pattern of q x 2^-147
// (jxlt_tile_kernel.h): with denormals flushed every offset would be 0, no overflow would be flagged and the transform
// search would go wrong without a sign.  The Makefile pins -fno-gpu-flush-denormals-to-zero; this
// kernel checks the product on the device itself -- once per device and process, in jxlt_context_create, which fails
// loudly if it does not hold: bits(3.0f x 2^-147) = 12.
__global__ void denormal_probe_kernel(float q, uint32_t* out) { *out = __float_as_uint(q * 0x1p-147f); }
uint32_t DenormalProbe(jxlt_context* ctx) {
  uint32_t* slot = &ctx->mail.p->denormal_probe;
  *slot = 0xFFFFFFFFu;
  hipLaunchKernelGGL(denormal_probe_kernel, dim3(1), dim3(1), 0, ctx->stream, 3.0f, slot);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return 0xFFFFFFFFu;
  return *(volatile uint32_t*)slot;
}
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
  // (an ORDINARY stream: created with the device's highest priority -- round 4's first form -- it bought nothing, the
  // DC-group sections' kernels do not get in beside token_kernel either way, and it made every kernel of the main
  // stream 18 % slower in a process whose first HIP streams are this context's)
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->dc_pack_stream, hipStreamNonBlocking);
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
  static std::atomic<bool> warmed[64], denormals_checked[64];
  if (device_ordinal < 64 && !denormals_checked[device_ordinal].exchange(true)) {
    const uint32_t bits = DenormalProbe(ctx);
    if (bits != 12u) {
      denormals_checked[device_ordinal] = false;
      char msg[160];
      snprintf(msg, sizeof(msg), "the device code does not keep FP32 denormals (3.0f * 2^-147 has the bits 0x%08x, not 12): "
               "build with -fno-gpu-flush-denormals-to-zero", bits);
      g_create_error = msg;
      jxlt_context_destroy(ctx);
      return JXLT_ERR_INTERNAL;
    }
  }
  if (copy_warmup && device_ordinal >= 0 && device_ordinal < 64 && !warmed[device_ordinal].exchange(true)) CopyWarmup(ctx, copy_warmup);
  ctx->counted = true;
  DeviceBlockCache::Get().ContextCreated(ctx->device);
  *out = ctx;
  return JXLT_OK;
}

int jxlt_debug_denormal_probe(jxlt_context* ctx, uint32_t* bits) {
  if (!ctx || !bits) return JXLT_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  *bits = DenormalProbe(ctx);
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


size_t jxlt_release_cached_memory(int device_ordinal) { return DeviceBlockCache::Get().Release(device_ordinal); }


int jxlt_output_buffer(jxlt_context* ctx, size_t bytes, uint8_t** out) {
  if (!ctx || !out) return JXLT_ERR_INVALID_ARGUMENT;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->h_output.cap < bytes) {
    if (ctx->h_output.p) {
      // The buffer grows WITH its contents, whoever wrote them: sections may be on their way into it
      // (jxlt_pack_deliver: waited for first), and the caller may have written bytes by CPU -- a prefix, ACGlobal --
      // that a later, larger request must not lose (include/jxl_tiny_amd.h says so; until round 4 the copy was made
      // only when a hand-over was in flight).
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



}  // extern "C"
