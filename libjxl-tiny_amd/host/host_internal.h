// Internal declarations shared by the host sources.
#ifndef JXLT_HOST_INTERNAL_H_
#define JXLT_HOST_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <vector>

#include "../../include/jxl_tiny_amd.h"
#include "encoder/enc_bit_writer.h"

struct jxlt_context;

namespace jxlt {
// Device context of the calling thread for the device chosen by jxl::SetEncoderDevice (or null).
jxlt_context* AcquireThreadContext();
// Where EncodeFrameOnContext leaves the frame: appended to `writer`; or written to the address
// `placer(frame_bytes)` returns; or (in_context) assembled, behind the bytes of `prefix`, in the
// context's page-locked output buffer, where the device places the AC sections itself.
struct ContextOutput {
  const std::vector<uint8_t>* prefix = nullptr;  // e.g. the file header
  const uint8_t* data = nullptr;                 // result: prefix + frame, valid until the next encode
  size_t size = 0;
};
bool EncodeFrameOnContext(jxlt_context* ctx, float distance, int num_threads, jxl::BitWriter* writer,
                          const std::function<uint8_t*(size_t)>* placer, ContextOutput* in_context = nullptr);
bool WriteFileHeader(size_t xsize, size_t ysize, jxl::BitWriter* writer);
// One frame over the calling thread's device list (jxl::SetEncoderDevices / JXLT_DEVICES); see enc_frame.cc.
bool EncodeOnDeviceList(const float* const planes[3], size_t pitch_bytes, const void* pfm_payload, int big_endian,
                        size_t xsize, size_t ysize, float distance, std::vector<uint8_t>* codestream, bool* used,
                        int* failure_code = nullptr);
bool ParsePFMHeader(const uint8_t* data, size_t size, size_t* xsize, size_t* ysize, bool* big_endian,
                    size_t* payload_offset);
bool NormalizeDistance(float* distance);
void SetStaticConstantEmulation(bool on);  // jxl::EmulateReferenceStaticConstants
// While that emulation is on: latches the process's first distance and sets it on `ctx` (a context's own
// jxlt_set_strategy_distance setting is left alone otherwise).
void ApplyStrategyDistanceEmulation(jxlt_context* ctx, float distance);
bool LastFrameTimeline(jxlt_frame_timeline* out);  // (enc_frame.cc)
}  // namespace jxlt

#endif  // JXLT_HOST_INTERNAL_H_
