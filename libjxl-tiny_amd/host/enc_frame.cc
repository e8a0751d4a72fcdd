// EncodeFrame: device pixel pipeline (through the C ABI of libjxltiny_hip.so)
// followed by host bitstream assembly.  Counterpart of
// /root/reference/encoder/enc_frame.cc:818-860 with the per-DC-group loop
// (:839-844) replaced by one device pass over all groups.
#include "encoder/enc_frame.h"

#include <stdio.h>

#include <thread>
#include <vector>

#include "../../include/jxl_tiny_amd.h"
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
bool EncodeFrameOnContext(jxlt_context* ctx, float distance, int num_threads, jxl::BitWriter* writer) {
  const DistanceParams distp = ComputeDistanceParams(distance);
  jxlt_params params;
  params.distance = distp.distance;
  params.scale = distp.scale;
  params.inv_scale = distp.inv_scale;
  params.scale_dc = distp.scale_dc;
  params.x_qm_scale = distp.x_qm_scale;
  params.flags = 0;
  if (jxlt_encode_enqueue(ctx, &params) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: device encode failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  jxlt_frame_result res;
  const uint32_t* hist = nullptr;
  if (jxlt_fetch_side_info(ctx, &res, &hist) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: fetch failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  const size_t num_dc_groups = ((res.xsize + 2047) / 2048) * ((res.ysize + 2047) / 2048);
  if (res.num_groups + num_dc_groups == 2) {
    if (jxlt_fetch_result(ctx, &res) != JXLT_OK) return false;
    const uint8_t* ptr = res.tokens;
    const size_t len = static_cast<size_t>(res.group_token_offset[1]);
    return AssembleFrame(ViewOf(res, &ptr, &len), distp, writer, num_threads);
  }
  const FrameView view = ViewOf(res, nullptr, nullptr);
  EntropyCode ac_code, dc_code;
  std::vector<jxl::BitWriter> dc_sections;
  BuildAcCode(hist, &ac_code);
  std::vector<uint32_t> table(64 * 64);
  FillCodeTable(ac_code, table.data());
  // DC groups on host threads while the device packs the AC sections.
  std::thread dc_thread([&]() { BuildDcSections(view, num_threads, &dc_code, &dc_sections); });
  jxlt_packed_sections packed;
  const int rc = jxlt_pack_ac_sections(ctx, table.data(), &packed);
  dc_thread.join();
  if (rc != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: section packing failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  PackedSections ac = {packed.bytes, packed.section_offset, packed.section_bits, packed.num_sections};
  return FinishFrame(view, distp, dc_code, &dc_sections, ac_code, ac, writer);
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

jxlt_context* AcquireContext() {
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

}  // namespace

void SetEncoderDevice(int device_ordinal) { g_device = device_ordinal; }

Status EncodeFrame(const float distance, const Image3F& linear, ThreadPool* pool,
                   BitWriter* writer) {
  if (linear.xsize() == 0 || linear.ysize() == 0 || !(distance > 0)) return false;
  jxlt_context* ctx = AcquireContext();
  if (!ctx) return false;  // no CPU fallback by design
  const float* planes[3] = {linear.ConstPlaneRow(0, 0), linear.ConstPlaneRow(1, 0),
                            linear.ConstPlaneRow(2, 0)};
  if (jxlt_image_upload(ctx, planes, linear.bytes_per_row(), linear.xsize(), linear.ysize()) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: upload failed: %s\n", jxlt_last_error(ctx));
    return false;
  }
  return jxlt::EncodeFrameOnContext(ctx, distance, pool ? pool->NumThreads() : 0, writer);
}

}  // namespace jxl
