// Frame batches (BASELINE config #5): several device contexts on one GPU, one host thread each,
// fed from a shared queue of frames -- the upload of one frame overlaps the kernels of another
// and the download of a third.  See include/jxl_tiny_amd.h (jxlt_batch_encoder_*).
// Every frame goes through exactly the path of jxl::EncodeFile (enc_file.cc:55-105 in the
// reference): file header, then EncodeFrameOnContext.
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/jxl_tiny_amd.h"
#include "encoder/enc_bit_writer.h"
#include "host_internal.h"

struct jxlt_batch_encoder {
  std::vector<jxlt_context*> lanes;
  std::vector<int> lane_device;  // GPU ordinal of every lane
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

int jxlt_batch_encoder_create_multi(const int* device_ordinals, int num_devices, int lanes_per_device,
                                    jxlt_batch_encoder** out) {
  if (!out) return JXLT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!device_ordinals || num_devices < 1 || num_devices > 64) return JXLT_ERR_INVALID_ARGUMENT;
  if (lanes_per_device <= 0) lanes_per_device = 3;
  if (lanes_per_device > 16) lanes_per_device = 16;
  jxlt_batch_encoder* enc = new jxlt_batch_encoder;
  // lane order: one lane of every device first, then the second of every device, ... so that a short batch
  // spreads over the GPUs before it doubles up on one
  for (int l = 0; l < lanes_per_device; ++l) {
    for (int d = 0; d < num_devices; ++d) {
      jxlt_context* ctx = nullptr;
      const int rc = jxlt_context_create(device_ordinals[d], &ctx);
      if (rc != JXLT_OK) {  // no device: no CPU fallback
        jxlt_batch_encoder_destroy(enc);
        return rc;
      }
      // (lanes share their GPU and the host's CPUs: their waits poll in short sleeps instead of spinning -- with eight
      // lanes, eight encoding threads + their helpers spun through the control group's CPU quota, round 5)
      if (lanes_per_device > 1) (void)jxlt_context_set_wait_mode(ctx, 1);
      enc->lanes.push_back(ctx);
      enc->lane_device.push_back(device_ordinals[d]);
    }
  }
  *out = enc;
  return JXLT_OK;
}

int jxlt_batch_encoder_create(int device_ordinal, int lanes, jxlt_batch_encoder** out) {
  return jxlt_batch_encoder_create_multi(&device_ordinal, 1, lanes, out);
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
  // One queue: a lane claims the first frame nobody has taken that it can take -- any frame in host memory,
  // or a frame in the memory of the lane's own GPU.
  std::vector<std::atomic<bool>> taken(num_frames);
  for (auto& t : taken) t.store(false);
  std::vector<int> status(num_frames, JXLT_ERR_INVALID_ARGUMENT);  // (stays for device frames no lane could take)
  bool multi_device = false;
  for (int d : enc->lane_device) multi_device |= d != enc->lane_device[0];
  auto lane = [&](size_t l) {
    // (short sleeps need a short timer slack: the default 50 us would make a 8-us sleep a 60-us one)
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    jxlt_context* ctx = enc->lanes[l];
    const int dev = enc->lane_device[l];
    size_t from = 0;
    for (;;) {
      size_t i = from;
      bool advanced_past_all_mine = true;
      for (; i < num_frames; ++i) {
        if (taken[i].load(std::memory_order_relaxed)) {
          if (advanced_past_all_mine) from = i + 1;
          continue;
        }
        const bool mine = !frames[i].in_device_memory || !multi_device || frames[i].device_ordinal == dev;
        if (!mine) {
          advanced_past_all_mine = false;
          continue;
        }
        bool expected = false;
        if (taken[i].compare_exchange_strong(expected, true)) break;
      }
      if (i >= num_frames) return;
      status[i] = EncodeOne(ctx, frames[i], distance, &out_bytes[i], &out_sizes[i]);
    }
  };
  // (a multi-device encoder starts every lane: a frame in device memory needs a lane of ITS device)
  const size_t used = multi_device || num_frames >= enc->lanes.size() ? enc->lanes.size() : num_frames;
  std::vector<std::thread> threads;
  for (size_t l = 1; l < used; ++l) threads.emplace_back(lane, l);
  if (used) lane(0);
  for (std::thread& t : threads) t.join();
  for (size_t i = 0; i < num_frames; ++i)
    if (status[i] != JXLT_OK) return status[i];
  return JXLT_OK;
}

}  // extern "C"
