// Drop-in for /root/reference/encoder/read_pfm.h:14.
#ifndef JXLT_HOST_ENCODER_READ_PFM_H_
#define JXLT_HOST_ENCODER_READ_PFM_H_

#include "encoder/image.h"

namespace jxl {

// Reads a colour PFM ("PF", interleaved RGB f32, bottom-to-top rows, sign of
// the scale = endianness) into a planar image.
bool ReadPFM(const char* filename, Image3F* image);

}  // namespace jxl

#endif  // JXLT_HOST_ENCODER_READ_PFM_H_
