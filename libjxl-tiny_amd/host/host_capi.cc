// C ABI of libjxltiny_host.so (see include/jxl_tiny_amd.h).
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../../include/jxl_tiny_amd_testing.h"
#include "encoder/enc_file.h"
#include "encoder/enc_frame.h"
#include "frame_assembler.h"
#include "host_internal.h"

namespace {

int ToMalloc(const std::vector<uint8_t>& bytes, uint8_t** out_bytes, size_t* out_size) {
  uint8_t* p = static_cast<uint8_t*>(malloc(bytes.size() ? bytes.size() : 1));
  if (!p) return JXLT_ERR_OUT_OF_MEMORY;
  memcpy(p, bytes.data(), bytes.size());
  *out_bytes = p;
  *out_size = bytes.size();
  return JXLT_OK;
}

jxlt::DistanceParams FromC(const jxlt_distance_params& d) {
  jxlt::DistanceParams p;
  p.distance = d.distance;
  p.global_scale = d.global_scale;
  p.quant_dc = d.quant_dc;
  p.scale = d.scale;
  p.inv_scale = d.inv_scale;
  p.scale_dc = d.scale_dc;
  p.x_qm_scale = d.x_qm_scale;
  p.epf_iters = d.epf_iters;
  return p;
}

}  // namespace

extern "C" {

void jxlt_compute_distance_params(float distance, jxlt_distance_params* out) {
  const jxlt::DistanceParams p = jxlt::ComputeDistanceParams(distance);
  out->distance = p.distance;
  out->global_scale = p.global_scale;
  out->quant_dc = p.quant_dc;
  out->scale = p.scale;
  out->inv_scale = p.inv_scale;
  out->scale_dc = p.scale_dc;
  out->x_qm_scale = p.x_qm_scale;
  out->epf_iters = p.epf_iters;
}

int jxlt_assemble_frame_groups(const jxlt_frame_result* frame, const uint8_t* const* group_tokens,
                               const size_t* group_token_bytes, const jxlt_distance_params* distp,
                               int num_threads, uint8_t** out_bytes, size_t* out_size) {
  if (!frame || !group_tokens || !group_token_bytes || !distp || !out_bytes || !out_size)
    return JXLT_ERR_INVALID_ARGUMENT;
  jxlt::FrameView view;
  view.xsize = frame->xsize;
  view.ysize = frame->ysize;
  for (int c = 0; c < 3; ++c) view.quant_dc[c] = frame->quant_dc[c];
  view.raw_quant_field = frame->raw_quant_field;
  view.ac_strategy = frame->ac_strategy;
  view.ytox_map = frame->ytox_map;
  view.ytob_map = frame->ytob_map;
  view.group_tokens = group_tokens;
  view.group_token_bytes = group_token_bytes;
  jxl::BitWriter writer;
  if (!jxlt::AssembleFrame(view, FromC(*distp), &writer, num_threads)) return JXLT_ERR_INTERNAL;
  return ToMalloc(writer.TakeBytes(), out_bytes, out_size);
}

int jxlt_assemble_frame(const jxlt_frame_result* frame, const jxlt_distance_params* distp,
                        int num_threads, uint8_t** out_bytes, size_t* out_size) {
  if (!frame || !frame->tokens || !frame->group_token_offset) return JXLT_ERR_INVALID_ARGUMENT;
  std::vector<const uint8_t*> ptr(frame->num_groups);
  std::vector<size_t> len(frame->num_groups);
  for (size_t g = 0; g < frame->num_groups; ++g) {
    ptr[g] = frame->tokens + frame->group_token_offset[g];
    len[g] = static_cast<size_t>(frame->group_token_offset[g + 1] - frame->group_token_offset[g]);
  }
  return jxlt_assemble_frame_groups(frame, ptr.data(), len.data(), distp, num_threads, out_bytes,
                                    out_size);
}

int jxlt_encode_pfm_file(const char* filename, float distance, int device_ordinal, uint8_t** out_bytes,
                         size_t* out_size) {
  if (!filename || !out_bytes || !out_size) return JXLT_ERR_INVALID_ARGUMENT;
  jxl::SetEncoderDevice(device_ordinal);
  std::vector<uint8_t> out;
  if (!jxl::EncodePFMFile(filename, distance, &out)) return JXLT_ERR_INTERNAL;
  uint8_t* buf = static_cast<uint8_t*>(malloc(out.size() ? out.size() : 1));
  if (!buf) return JXLT_ERR_OUT_OF_MEMORY;
  memcpy(buf, out.data(), out.size());
  *out_bytes = buf;
  *out_size = out.size();
  return JXLT_OK;
}

void jxlt_emulate_reference_static_constants(int on) { jxl::EmulateReferenceStaticConstants(on != 0); }
void jxlt_emulate_reference_single_symbol_codes(int on) { jxl::EmulateReferenceSingleSymbolCodes(on != 0); }

int jxlt_write_file_header(size_t xsize, size_t ysize, uint8_t** out_bytes, size_t* out_size) {
  jxl::BitWriter writer;
  if (!jxlt::WriteFileHeader(xsize, ysize, &writer)) return JXLT_ERR_INVALID_ARGUMENT;
  return ToMalloc(writer.TakeBytes(), out_bytes, out_size);
}

int jxlt_encode_file_planar(const float* const planes[3], size_t pitch_bytes, size_t xsize,
                            size_t ysize, float distance, int device_ordinal, uint8_t** out_bytes,
                            size_t* out_size) {
  if (!planes || !out_bytes || !out_size || xsize == 0 || ysize == 0 ||
      pitch_bytes < xsize * sizeof(float) || pitch_bytes % sizeof(float))
    return JXLT_ERR_INVALID_ARGUMENT;
  // Same as jxl::EncodeFile (enc_file.cc:55-105) but the planes go to the device straight from
  // the caller's memory (no intermediate Image3F copy).
  jxl::SetEncoderDevice(device_ordinal);
  jxlt_context* ctx = jxlt::AcquireThreadContext();
  if (!ctx) return JXLT_ERR_NO_DEVICE;
  {
    float d = distance;
    if (!jxlt::NormalizeDistance(&d)) return JXLT_ERR_INVALID_ARGUMENT;
  }
  const int rc = jxlt_image_upload(ctx, planes, pitch_bytes, xsize, ysize);
  if (rc != JXLT_OK) return rc;
  return jxlt_encode_resident(ctx, distance, 0, out_bytes, out_size);
}

int jxlt_encode_file_planar_devices(const float* const planes[3], size_t pitch_bytes, size_t xsize, size_t ysize,
                                    float distance, const int* device_ordinals, int num_devices, uint8_t** out_bytes,
                                    size_t* out_size) {
  if (!planes || !out_bytes || !out_size || xsize == 0 || ysize == 0 || pitch_bytes < xsize * sizeof(float) ||
      pitch_bytes % sizeof(float) || (num_devices > 0 && !device_ordinals))
    return JXLT_ERR_INVALID_ARGUMENT;
  {
    float d = distance;
    if (!jxlt::NormalizeDistance(&d)) return JXLT_ERR_INVALID_ARGUMENT;
  }
  jxl::SetEncoderDevices(device_ordinals, num_devices);  // (the calling thread's list)
  std::vector<uint8_t> whole;
  bool used = false;
  int code = JXLT_ERR_INTERNAL;
  if (!jxlt::EncodeOnDeviceList(planes, pitch_bytes, nullptr, 0, xsize, ysize, distance, &whole, &used, &code)) return code;
  if (used) return ToMalloc(whole, out_bytes, out_size);
  return jxlt_encode_file_planar(planes, pitch_bytes, xsize, ysize, distance, num_devices > 0 ? device_ordinals[0] : 0,
                                 out_bytes, out_size);
}

namespace {
// What a failed encode on `ctx` returns: JXLT_ERR_UNSUPPORTED when the device refused the frame's values (a quantised
// coefficient beyond the token format's 16 bits, a DC value beyond int16 -- the status sticks to the context until
// the next encode), JXLT_ERR_INTERNAL otherwise.
int FailureCode(jxlt_context* ctx) {
  jxlt_encode_stats_t st;
  return jxlt_encode_stats(ctx, &st) == JXLT_ERR_UNSUPPORTED ? JXLT_ERR_UNSUPPORTED : JXLT_ERR_INTERNAL;
}
// Shared by the two resident entry points: file header + frame into the buffer `alloc` returns.
int EncodeResident(jxlt_context* ctx, float distance, int num_threads,
                   const std::function<uint8_t*(size_t)>& alloc, size_t* out_size) {
  if (!jxlt::NormalizeDistance(&distance)) return JXLT_ERR_INVALID_ARGUMENT;
  size_t xsize = 0, ysize = 0;
  if (jxlt_image_size(ctx, &xsize, &ysize) != JXLT_OK) return JXLT_ERR_INVALID_ARGUMENT;
  jxl::BitWriter writer;
  if (!jxlt::WriteFileHeader(xsize, ysize, &writer)) return JXLT_ERR_INVALID_ARGUMENT;
  const std::vector<uint8_t> file_header = writer.TakeBytes();
  const std::function<uint8_t*(size_t)> placer = [&](size_t frame_bytes) -> uint8_t* {
    uint8_t* buf = alloc(file_header.size() + frame_bytes);
    if (!buf) return nullptr;
    memcpy(buf, file_header.data(), file_header.size());
    *out_size = file_header.size() + frame_bytes;
    return buf + file_header.size();
  };
  return jxlt::EncodeFrameOnContext(ctx, distance, num_threads, nullptr, &placer) ? JXLT_OK : FailureCode(ctx);
}
}  // namespace

int jxlt_encode_resident(jxlt_context* ctx, float distance, int num_threads, uint8_t** out_bytes,
                         size_t* out_size) {
  if (!ctx || !out_bytes || !out_size) return JXLT_ERR_INVALID_ARGUMENT;
  uint8_t* buf = nullptr;
  const int rc = EncodeResident(ctx, distance, num_threads, [&](size_t n) {
    buf = static_cast<uint8_t*>(malloc(n ? n : 1));
    return buf;
  }, out_size);
  if (rc != JXLT_OK) {
    // copies into `buf` may already be queued on the context's copy stream: let them land before the
    // memory goes back to the heap
    if (buf) (void)jxlt_synchronize(ctx);
    free(buf);
    return rc;
  }
  *out_bytes = buf;
  return JXLT_OK;
}

int jxlt_encode_resident_view(jxlt_context* ctx, float distance, int num_threads, const uint8_t** bytes,
                              size_t* size) {
  if (!ctx || !bytes || !size) return JXLT_ERR_INVALID_ARGUMENT;
  if (!jxlt::NormalizeDistance(&distance)) return JXLT_ERR_INVALID_ARGUMENT;
  size_t xsize = 0, ysize = 0;
  if (jxlt_image_size(ctx, &xsize, &ysize) != JXLT_OK) return JXLT_ERR_INVALID_ARGUMENT;
  jxl::BitWriter writer;
  if (!jxlt::WriteFileHeader(xsize, ysize, &writer)) return JXLT_ERR_INVALID_ARGUMENT;
  const std::vector<uint8_t> file_header = writer.TakeBytes();
  // The codestream is assembled in the context's page-locked output buffer; the sections arrive
  // there by the copy commands of jxlt_pack_deliver, header + TOC are set in front of them.
  jxlt::ContextOutput out;
  out.prefix = &file_header;
  if (!jxlt::EncodeFrameOnContext(ctx, distance, num_threads, nullptr, nullptr, &out)) return FailureCode(ctx);
  *bytes = out.data;
  *size = out.size;
  return JXLT_OK;
}

int jxlt_last_frame_timeline(jxlt_frame_timeline* out) {
  return out && jxlt::LastFrameTimeline(out) ? JXLT_OK : JXLT_ERR_INVALID_ARGUMENT;
}

int jxlt_build_code_tables(const uint32_t* ac_histograms, const uint32_t* dc_histograms,
                           uint32_t* ac_code_table, uint32_t* dc_code_table) {
  if (!ac_histograms || !dc_histograms || !ac_code_table || !dc_code_table) return JXLT_ERR_INVALID_ARGUMENT;
  jxlt::EntropyCode ac_code, dc_code;
  jxlt::BuildAcCode(ac_histograms, &ac_code);
  jxlt::BuildDcCode(dc_histograms, &dc_code);
  jxlt::FillCodeTable(ac_code, ac_code_table);
  jxlt::FillCodeTable(dc_code, dc_code_table);
  return JXLT_OK;
}

int jxlt_finish_frame(size_t xsize, size_t ysize, float distance, const uint32_t* ac_histograms,
                      const uint32_t* dc_histograms, const jxlt_packed_sections* dc_sections,
                      const jxlt_packed_sections* ac_sections, uint8_t** out_bytes, size_t* out_size) {
  if (!ac_histograms || !dc_histograms || !dc_sections || !ac_sections || !out_bytes || !out_size)
    return JXLT_ERR_INVALID_ARGUMENT;
  if (!jxlt::NormalizeDistance(&distance)) return JXLT_ERR_INVALID_ARGUMENT;
  jxl::BitWriter writer;
  if (!jxlt::WriteFileHeader(xsize, ysize, &writer)) return JXLT_ERR_INVALID_ARGUMENT;
  jxlt::EntropyCode ac_code, dc_code;
  jxlt::BuildAcCode(ac_histograms, &ac_code);
  jxlt::BuildDcCode(dc_histograms, &dc_code);
  const jxlt::PackedSections dc = {dc_sections->bytes, dc_sections->section_offset,
                                   dc_sections->section_bits, dc_sections->num_sections};
  const jxlt::PackedSections ac = {ac_sections->bytes, ac_sections->section_offset,
                                   ac_sections->section_bits, ac_sections->num_sections};
  jxlt::FramePieces pieces;
  if (!jxlt::FinishFrame(xsize, ysize, jxlt::ComputeDistanceParams(distance), dc_code, dc, ac_code, ac, &pieces))
    return JXLT_ERR_INVALID_ARGUMENT;
  const std::vector<uint8_t> file_header = writer.TakeBytes();
  const size_t dc_bytes = static_cast<size_t>(dc.offset[dc.n]), ac_bytes = static_cast<size_t>(ac.offset[ac.n]);
  const size_t total = file_header.size() + pieces.head.size() + dc_bytes + pieces.ac_global.size() + ac_bytes;
  uint8_t* buf = static_cast<uint8_t*>(malloc(total ? total : 1));
  if (!buf) return JXLT_ERR_OUT_OF_MEMORY;
  size_t pos = 0;
  auto put = [&](const uint8_t* p, size_t n) {
    memcpy(buf + pos, p, n);
    pos += n;
  };
  put(file_header.data(), file_header.size());
  put(pieces.head.data(), pieces.head.size());
  put(dc.bytes, dc_bytes);
  put(pieces.ac_global.data(), pieces.ac_global.size());
  put(ac.bytes, ac_bytes);
  *out_bytes = buf;
  *out_size = total;
  return JXLT_OK;
}

int jxlt_debug_dc_records(const jxlt_frame_result* frame, size_t dc_group_index, uint8_t** out_bytes,
                          size_t* out_size) {
  if (!frame || !out_bytes || !out_size) return JXLT_ERR_INVALID_ARGUMENT;
  jxlt::FrameView view;
  view.xsize = frame->xsize;
  view.ysize = frame->ysize;
  for (int c = 0; c < 3; ++c) view.quant_dc[c] = frame->quant_dc[c];
  view.raw_quant_field = frame->raw_quant_field;
  view.ac_strategy = frame->ac_strategy;
  view.ytox_map = frame->ytox_map;
  view.ytob_map = frame->ytob_map;
  view.group_tokens = nullptr;
  view.group_token_bytes = nullptr;
  return ToMalloc(jxlt::DcGroupRecords(view, dc_group_index), out_bytes, out_size);
}

void jxlt_free(void* p) { free(p); }

}  // extern "C"
