// Drop-in for /root/reference/encoder/enc_file.h:20-21.
#ifndef JXLT_HOST_ENCODER_ENC_FILE_H_
#define JXLT_HOST_ENCODER_ENC_FILE_H_

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "encoder/image.h"

namespace jxl {

// Compresses `input` (linear sRGB, nominal range [0,1], values outside allowed)
// to a JPEG XL codestream at the given butteraugli distance.
bool EncodeFile(const Image3F& input, float distance, std::vector<uint8_t>* output);

// Addition (not in the reference): ReadPFM + EncodeFile in one call, without ever building
// the planar Image3F on the host.  The file is read into page-locked memory, its sample
// payload goes to the GPU as it is (interleaved RGB, bottom row first, either byte order) and
// the kernels read it in place.  Same output bytes as ReadPFM + EncodeFile.
bool EncodePFMFile(const char* filename, float distance, std::vector<uint8_t>* output,
                   size_t* xsize = nullptr, size_t* ysize = nullptr);

}  // namespace jxl

#endif  // JXLT_HOST_ENCODER_ENC_FILE_H_
