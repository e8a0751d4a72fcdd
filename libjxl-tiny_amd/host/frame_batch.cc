// Frame batches (BASELINE config #5): several device contexts on one GPU, one host thread each,
// fed from a shared queue of frames -- the upload of one frame overlaps the kernels of another
// and the download of a third.  See include/jxl_tiny_amd.h (jxlt_batch_encoder_*).
// Every frame goes through exactly the path of jxl::EncodeFile (enc_file.cc:55-105 in the
// reference): file header, then EncodeFrameOnContext.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/jxl_tiny_amd.h"
#include "encoder/enc_bit_writer.h"
#include "host_internal.h"

struct jxlt_batch_encoder {
  int device = 0;
  std::vector<jxlt_context*> lanes;
};

namespace {

int EncodeOne(jxlt_context* ctx, const jxlt_batch_frame& f, float distance, uint8_t** out_bytes,
              size_t* out_size) {
  const bool planar = f.planes[0] && f.planes[1] && f.planes[2];
  if (f.xsize == 0 || f.ysize == 0 || (planar == (f.pfm_payload != nullptr))) return JXLT_ERR_INVALID_ARGUMENT;
  if (planar && (f.pitch_bytes < f.xsize * sizeof(float) || f.pitch_bytes % sizeof(float)))
    return JXLT_ERR_INVALID_ARGUMENT;
  jxl::BitWriter writer;
  if (!jxlt::WriteFileHeader(f.xsize, f.ysize, &writer)) return JXLT_ERR_INVALID_ARGUMENT;
  const std::vector<uint8_t> file_header = writer.TakeBytes();
  int rc;
  if (f.in_device_memory) {
    const void* const dev[3] = {f.planes[0], f.planes[1], f.planes[2]};
    rc = planar ? jxlt_image_set_device(ctx, dev, f.pitch_bytes, f.xsize, f.ysize)
                : jxlt_image_set_device_pfm(ctx, f.pfm_payload, f.xsize, f.ysize, f.pfm_big_endian);
  } else {
    rc = planar ? jxlt_image_upload(ctx, f.planes, f.pitch_bytes, f.xsize, f.ysize)
                : jxlt_image_upload_pfm(ctx, f.pfm_payload, f.xsize, f.ysize, f.pfm_big_endian);
  }
  if (rc != JXLT_OK) return rc;
  jxlt::ContextOutput out;
  out.prefix = &file_header;
  if (!jxlt::EncodeFrameOnContext(ctx, distance, 0, nullptr, nullptr, &out)) return JXLT_ERR_INTERNAL;
  uint8_t* copy = static_cast<uint8_t*>(malloc(out.size ? out.size : 1));
  if (!copy) return JXLT_ERR_OUT_OF_MEMORY;
  memcpy(copy, out.data, out.size);
  *out_bytes = copy;
  *out_size = out.size;
  return JXLT_OK;
}

}  // namespace

extern "C" {

int jxlt_batch_encoder_create(int device_ordinal, int lanes, jxlt_batch_encoder** out) {
  if (!out) return JXLT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (lanes <= 0) lanes = 3;
  if (lanes > 16) lanes = 16;
  jxlt_batch_encoder* enc = new jxlt_batch_encoder;
  enc->device = device_ordinal;
  for (int i = 0; i < lanes; ++i) {
    jxlt_context* ctx = nullptr;
    const int rc = jxlt_context_create(device_ordinal, &ctx);
    if (rc != JXLT_OK) {  // no device: no CPU fallback
      jxlt_batch_encoder_destroy(enc);
      return rc;
    }
    enc->lanes.push_back(ctx);
  }
  *out = enc;
  return JXLT_OK;
}

void jxlt_batch_encoder_destroy(jxlt_batch_encoder* enc) {
  if (!enc) return;
  for (jxlt_context* ctx : enc->lanes) jxlt_context_destroy(ctx);
  delete enc;
}

int jxlt_batch_encoder_run(jxlt_batch_encoder* enc, const jxlt_batch_frame* frames, size_t num_frames,
                           float distance, uint8_t** out_bytes, size_t* out_sizes) {
  if (!enc || (num_frames && (!frames || !out_bytes || !out_sizes))) return JXLT_ERR_INVALID_ARGUMENT;
  if (!jxlt::NormalizeDistance(&distance)) return JXLT_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < num_frames; ++i) {
    out_bytes[i] = nullptr;
    out_sizes[i] = 0;
  }
  std::atomic<size_t> next(0);
  std::vector<int> status(num_frames, JXLT_OK);
  auto lane = [&](jxlt_context* ctx) {
    for (size_t i; (i = next.fetch_add(1)) < num_frames;)
      status[i] = EncodeOne(ctx, frames[i], distance, &out_bytes[i], &out_sizes[i]);
  };
  const size_t used = num_frames < enc->lanes.size() ? num_frames : enc->lanes.size();
  std::vector<std::thread> threads;
  for (size_t l = 1; l < used; ++l) threads.emplace_back(lane, enc->lanes[l]);
  if (used) lane(enc->lanes[0]);
  for (std::thread& t : threads) t.join();
  for (size_t i = 0; i < num_frames; ++i)
    if (status[i] != JXLT_OK) return status[i];
  return JXLT_OK;
}

}  // extern "C"
