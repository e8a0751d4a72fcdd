// EncodeFile: codestream signature, size header and fixed image metadata, then
// one frame.  Bit layout follows /root/reference/encoder/enc_file.cc:26-105.
#include "encoder/enc_file.h"

#include <stdio.h>

#include "encoder/enc_bit_writer.h"
#include "encoder/enc_frame.h"
#include "host_internal.h"

namespace jxlt {

namespace {
void WriteSize(uint32_t size, jxl::BitWriter* writer) {  // enc_file.cc:28-38
  size -= 1;
  static const uint32_t kBits[4] = {9, 13, 18, 30};
  for (uint32_t i = 0; i < 4; ++i) {
    if (size < (1u << kBits[i])) {
      writer->Write(2, i);
      writer->Write(kBits[i], size);
      return;
    }
  }
}
}  // namespace

bool WriteFileHeader(size_t xsize, size_t ysize, jxl::BitWriter* writer) {
  if (xsize == 0 || ysize == 0) return false;
  if (xsize > 0x3FFFFFFFull || ysize > 0x3FFFFFFFull) return false;  // "Image too large"
  writer->Write(8, 0xFF);
  writer->Write(8, 0x0A);  // codestream marker
  writer->Write(1, 0);     // small
  WriteSize(static_cast<uint32_t>(ysize), writer);
  writer->Write(3, 0);  // ratio
  WriteSize(static_cast<uint32_t>(xsize), writer);
  writer->Write(1, 0);  // not all default image metadata
  writer->Write(1, 0);  // no extra fields in image metadata
  writer->Write(1, 1);  // floating point samples
  writer->Write(2, 0);  // 32 bits per sample
  writer->Write(4, 7);  // 8 exponent bits per sample
  writer->Write(1, 0);  // modular 16 bit sufficient
  writer->Write(2, 0);  // no extra channels
  writer->Write(1, 1);  // xyb encoded
  writer->Write(1, 0);  // not all default color encoding
  writer->Write(1, 0);  // no icc
  writer->Write(2, 0);  // RGB color space
  writer->Write(2, 1);  // D65 white point
  writer->Write(2, 1);  // SRGB primaries
  writer->Write(1, 0);  // no gamma
  writer->Write(2, 2);  // transfer function selector bits (2 .. 17)
  writer->Write(4, 6);  // linear transfer function (enum value 8)
  writer->Write(2, 1);  // relative rendering intent
  writer->Write(2, 0);  // no extensions
  writer->Write(1, 1);  // all default transform data
  writer->ZeroPadToByte();
  return true;
}

bool NormalizeDistance(float* distance) {  // enc_file.cc:57-65
  if (*distance < 0.0) {
    fprintf(stderr, "Invalid butteraugli distance (%f)\n", *distance);
    return false;
  } else if (*distance == 0.0) {
    fprintf(stderr, "Lossless compression is not supported.\n");
    return false;
  } else if (*distance <= 0.03) {
    *distance = 0.03;
  }
  return true;
}

}  // namespace jxlt

namespace jxl {

bool EncodeFile(const Image3F& input, float distance, std::vector<uint8_t>* output) {
  if (!jxlt::NormalizeDistance(&distance)) return false;
  if (input.xsize() == 0 || input.ysize() == 0) return false;  // "Empty image"
  BitWriter writer;
  if (!jxlt::WriteFileHeader(input.xsize(), input.ysize(), &writer)) return false;
  ThreadPool pool;
  if (!EncodeFrame(distance, input, &pool, &writer)) return false;
  *output = writer.TakeBytes();
  return true;
}

}  // namespace jxl
