// cjxl_tiny command line: <file in.pfm> [<file out.jxl>] [-d distance]
// Same interface as /root/reference/encoder/cjxl_main.cc:40-101, plus
// --device N to pick the GPU and --host-ingest to de-interleave the PFM on the host
// (ReadPFM + EncodeFile, as the reference does) instead of on the device.
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "encoder/enc_file.h"
#include "encoder/enc_frame.h"
#include "encoder/read_pfm.h"

namespace {

void Usage(const char* arg0) {
  fprintf(stderr,
          "Usage: %s <file in> [<file out>] [-d distance] [--device N] [--host-ingest]\n\n"
          "  NOTE: <file in> is a .pfm file in linear SRGB colorspace\n",
          arg0);
}

bool Save(const char* filename, const std::vector<uint8_t>& bytes) {
  FILE* f = fopen(filename, "wb");
  if (!f) {
    fprintf(stderr, "Could not open %s for writing\nError: %s", filename, strerror(errno));
    return false;
  }
  const bool wrote = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
  if (!wrote) fprintf(stderr, "Could not write to file\nError: %s", strerror(errno));
  if (fclose(f) != 0) {
    fprintf(stderr, "Could not close file\nError: %s", strerror(errno));
    return false;
  }
  return wrote;
}

}  // namespace

int main(int argc, char** argv) {
  const char* file_in = nullptr;
  const char* file_out = nullptr;
  float distance = 1.0f;
  bool host_ingest = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp("--host-ingest", argv[i])) {
      host_ingest = true;
      continue;
    }
    if (!strcmp("-h", argv[i]) || !strcmp("--help", argv[i])) {
      Usage(argv[0]);
      return EXIT_SUCCESS;
    }
    if (!strcmp("--device", argv[i])) {
      if (++i == argc) {
        fprintf(stderr, "--device requires an argument\n");
        return EXIT_FAILURE;
      }
      jxl::SetEncoderDevice(atoi(argv[i]));
      continue;
    }
    if (argv[i][0] == '-' && argv[i][1] == 'd') {
      const char* arg = argv[i][2] != '\0' ? &argv[i][2] : (++i < argc ? argv[i] : nullptr);
      if (!arg) {
        fprintf(stderr, "-d requires an argument\n");
        return EXIT_FAILURE;
      }
      char* end;
      distance = static_cast<float>(strtod(arg, &end));
      if (*end != '\0') {
        fprintf(stderr, "Unable to interpret as float: %s\n", arg);
        return EXIT_FAILURE;
      }
      continue;
    }
    if (!file_in) file_in = argv[i];
    else if (!file_out) file_out = argv[i];
  }
  if (!file_in) {
    fprintf(stderr, "Missing input file.\n");
    return EXIT_FAILURE;
  }
  std::vector<uint8_t> output;
  if (host_ingest) {
    jxl::Image3F image;
    if (!jxl::ReadPFM(file_in, &image)) {
      fprintf(stderr, "Error reading PFM input file.\n");
      return EXIT_FAILURE;
    }
    fprintf(stderr, "Read %zux%zu pixels input image.\n", image.xsize(), image.ysize());
    if (!jxl::EncodeFile(image, distance, &output)) {
      fprintf(stderr, "Encoding failed.\n");
      return EXIT_FAILURE;
    }
  } else {
    size_t xsize = 0, ysize = 0;
    const bool ok = jxl::EncodePFMFile(file_in, distance, &output, &xsize, &ysize);
    if (xsize) fprintf(stderr, "Read %zux%zu pixels input image.\n", xsize, ysize);
    if (!ok) {
      fprintf(stderr, xsize ? "Encoding failed.\n" : "Error reading PFM input file.\n");
      return EXIT_FAILURE;
    }
  }
  fprintf(stderr, "Compressed to %zu bytes.\n", output.size());
  if (file_out && !Save(file_out, output)) {
    fprintf(stderr, "Failed to write to output file %s\n", file_out);
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
