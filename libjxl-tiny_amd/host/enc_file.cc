// EncodeFile: codestream signature, size header and fixed image metadata, then
// one frame.  Bit layout follows /root/reference/encoder/enc_file.cc:26-105.
#include "encoder/enc_file.h"

#include <fcntl.h>
#include <stdio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "encoder/enc_bit_writer.h"
#include "encoder/enc_frame.h"
#include "host_internal.h"
#include "../../include/jxl_tiny_amd.h"

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
  // ImageMetadata of a float, XYB-encoded, linear-sRGB image (the only kind this encoder writes;
  // enc_file.cc:75-95): a fixed field list, kept as data -- {bits, value} in bitstream order.
  struct Field {
    uint8_t bits;
    uint8_t value;
  };
  static const Field kImageMetadata[] = {
      {1, 0},  // all_default = false
      {1, 0},  // extra_fields = false
      // bit depth
      {1, 1},  // floating_point_sample
      {2, 0},  // bits_per_sample selector: 32
      {4, 7},  // exponent_bits_per_sample - 1: 8
      {1, 0},  // modular_16_bit_buffer_sufficient = false
      {2, 0},  // num_extra_channels selector: 0
      {1, 1},  // xyb_encoded
      // colour encoding
      {1, 0},  // all_default = false
      {1, 0},  // want_icc = false
      {2, 0},  // colour space: RGB
      {2, 1},  // white point: D65
      {2, 1},  // primaries: sRGB
      {1, 0},  // have_gamma = false
      {2, 2},  // transfer function: selector for enum values 2 .. 17 ...
      {4, 6},  // ... value 8 = linear
      {2, 1},  // rendering intent: relative
      {2, 0},  // extensions: none
      {1, 1},  // default transform data
  };
  for (const Field& f : kImageMetadata) writer->Write(f.bits, f.value);
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

bool EncodePFMFile(const char* filename, float distance, std::vector<uint8_t>* output, size_t* xsize_out,
                   size_t* ysize_out) {
  if (!jxlt::NormalizeDistance(&distance)) return false;
  // The file is mapped, not read: its pages go from the page cache into the device library's
  // page-locked staging buffers (several threads) and from there over PCIe, overlapped.
  const int fd = open(filename, O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st) != 0 || st.st_size < 2) {
    if (fd >= 0) close(fd);
    fprintf(stderr, "Could not read %s\n", filename);
    return false;
  }
  const size_t size = static_cast<size_t>(st.st_size);
  void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) {
    fprintf(stderr, "Could not read %s\n", filename);
    return false;
  }
  (void)madvise(map, size, MADV_SEQUENTIAL);
  const uint8_t* data = static_cast<const uint8_t*>(map);
  size_t xsize = 0, ysize = 0, payload_offset = 0;
  bool big_endian = false;
  bool ok = jxlt::ParsePFMHeader(data, size, &xsize, &ysize, &big_endian, &payload_offset);
  if (ok && xsize_out) *xsize_out = xsize;
  if (ok && ysize_out) *ysize_out = ysize;
  // (the device comes behind the file: a readable image on a machine without a GPU is reported as an encoding
  // failure, like a reference whose EncodeFile fails, not as an unreadable file)
  jxlt_context* ctx = ok ? jxlt::AcquireThreadContext() : nullptr;
  if (ok && !ctx) {
    fprintf(stderr, "jxl_tiny_amd: no usable HIP device (there is no CPU fallback)\n");
    munmap(map, size);
    return false;
  }
  if (ok) {  // several GPUs configured: the payload's row slabs go to one GPU each
    bool used = false;
    const bool done = jxlt::EncodeOnDeviceList(nullptr, 0, data + payload_offset, big_endian ? 1 : 0, xsize, ysize,
                                               distance, output, &used);
    if (used || !done) {
      munmap(map, size);
      return done;
    }
  }
  BitWriter writer;
  ok = ok && jxlt::WriteFileHeader(xsize, ysize, &writer);
  if (ok && jxlt_image_upload_pfm(ctx, data + payload_offset, xsize, ysize, big_endian ? 1 : 0) != JXLT_OK) {
    fprintf(stderr, "jxl_tiny_amd: upload failed: %s\n", jxlt_last_error(ctx));
    ok = false;
  }
  munmap(map, size);
  if (!ok) return false;
  const std::vector<uint8_t> file_header = writer.TakeBytes();
  jxlt::ContextOutput out;
  out.prefix = &file_header;
  if (!jxlt::EncodeFrameOnContext(ctx, distance, 0, nullptr, nullptr, &out)) return false;
  output->assign(out.data, out.data + out.size);
  return true;
}

}  // namespace jxl
