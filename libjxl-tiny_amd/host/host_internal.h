// Internal declarations shared by the host sources.
#ifndef JXLT_HOST_INTERNAL_H_
#define JXLT_HOST_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>

#include <functional>

#include "encoder/enc_bit_writer.h"

struct jxlt_context;

namespace jxlt {
// Device context of the calling thread for the device chosen by jxl::SetEncoderDevice (or null).
jxlt_context* AcquireThreadContext();
bool EncodeFrameOnContext(jxlt_context* ctx, float distance, int num_threads, jxl::BitWriter* writer,
                          const std::function<uint8_t*(size_t)>* placer);
bool WriteFileHeader(size_t xsize, size_t ysize, jxl::BitWriter* writer);
bool NormalizeDistance(float* distance);
}  // namespace jxlt

#endif  // JXLT_HOST_INTERNAL_H_
