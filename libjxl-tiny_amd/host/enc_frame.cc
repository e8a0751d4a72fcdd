// EncodeFrame: device pixel pipeline (through the C ABI of libjxltiny_hip.so)
// followed by host bitstream assembly.  Counterpart of
// /root/reference/encoder/enc_frame.cc:818-860 with the per-DC-group loop
// (:839-844) replaced by one device pass over all groups.
#include "encoder/enc_frame.h"

#include <stdio.h>

#include "../../include/jxl_tiny_amd.h"
#include "frame_assembler.h"

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
  const jxlt::DistanceParams distp = jxlt::ComputeDistanceParams(distance);

  jxlt_context* ctx = AcquireContext();
  if (!ctx) return false;  // no CPU fallback by design
  const float* planes[3] = {linear.ConstPlaneRow(0, 0), linear.ConstPlaneRow(1, 0),
                            linear.ConstPlaneRow(2, 0)};
  jxlt_params params;
  params.distance = distp.distance;
  params.scale = distp.scale;
  params.inv_scale = distp.inv_scale;
  params.scale_dc = distp.scale_dc;
  params.x_qm_scale = distp.x_qm_scale;
  params.flags = 0;
  jxlt_frame_result res;
  if (jxlt_image_upload(ctx, planes, linear.bytes_per_row(), linear.xsize(), linear.ysize()) != JXLT_OK ||
      jxlt_encode_enqueue(ctx, &params) != JXLT_OK || jxlt_fetch_result(ctx, &res) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: device encode failed: %s\n", jxlt_last_error(ctx));
    return false;
  }

  std::vector<const uint8_t*> group_ptr(res.num_groups);
  std::vector<size_t> group_len(res.num_groups);
  for (size_t g = 0; g < res.num_groups; ++g) {
    group_ptr[g] = res.tokens + res.group_token_offset[g];
    group_len[g] = static_cast<size_t>(res.group_token_offset[g + 1] - res.group_token_offset[g]);
  }
  jxlt::FrameView view;
  view.xsize = res.xsize;
  view.ysize = res.ysize;
  for (int c = 0; c < 3; ++c) view.quant_dc[c] = res.quant_dc[c];
  view.raw_quant_field = res.raw_quant_field;
  view.ac_strategy = res.ac_strategy;
  view.ytox_map = res.ytox_map;
  view.ytob_map = res.ytob_map;
  view.group_tokens = group_ptr.data();
  view.group_token_bytes = group_len.data();
  return jxlt::AssembleFrame(view, distp, writer, pool ? pool->NumThreads() : 0);
}

}  // namespace jxl
