// Drop-in for /root/reference/encoder/enc_file.h:20-21.
#ifndef JXLT_HOST_ENCODER_ENC_FILE_H_
#define JXLT_HOST_ENCODER_ENC_FILE_H_

#include <stdint.h>

#include <vector>

#include "encoder/image.h"

namespace jxl {

// Compresses `input` (linear sRGB, nominal range [0,1], values outside allowed)
// to a JPEG XL codestream at the given butteraugli distance.
bool EncodeFile(const Image3F& input, float distance, std::vector<uint8_t>* output);

}  // namespace jxl

#endif  // JXLT_HOST_ENCODER_ENC_FILE_H_
