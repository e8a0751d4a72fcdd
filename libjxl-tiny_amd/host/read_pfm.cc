// PFM reader.  Behaviour follows /root/reference/encoder/read_pfm.cc:24-213:
// header "PF" <ws> width <blank> height <ws> scale <ws>, scale must be +-1.0
// (negative = little endian), rows stored bottom-to-top, RGB interleaved.
// Unlike the reference the payload size is validated.
#include "encoder/read_pfm.h"

#include <stdio.h>
#include <string.h>

#include <vector>

namespace jxl {
namespace {

struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  bool AtEnd() const { return p >= end; }
};

bool IsSpace(uint8_t c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }

bool EatOneWhitespace(Cursor* c) {
  if (c->AtEnd() || !IsSpace(*c->p)) return false;
  ++c->p;
  return true;
}

bool EatBlank(Cursor* c) {  // exactly one ' ' or '\n' (read_pfm.cc:117-129)
  if (c->AtEnd() || (*c->p != ' ' && *c->p != '\n')) return false;
  ++c->p;
  return true;
}

bool ParseSize(Cursor* c, size_t* out) {
  if (c->AtEnd() || *c->p < '0' || *c->p > '9') return false;
  size_t v = 0;
  while (!c->AtEnd() && *c->p >= '0' && *c->p <= '9') {
    v = v * 10 + static_cast<size_t>(*c->p - '0');
    ++c->p;
  }
  *out = v;
  return true;
}

bool ParseScale(Cursor* c, double* out) {
  if (c->AtEnd()) return false;
  bool neg = false;
  if (*c->p == '-' || *c->p == '+') {
    neg = *c->p == '-';
    ++c->p;
    if (c->AtEnd()) return false;
  } else if (*c->p < '0' || *c->p > '9') {
    return false;
  }
  double v = 0.0;
  while (!c->AtEnd() && *c->p >= '0' && *c->p <= '9') {
    v = v * 10 + (*c->p - '0');
    ++c->p;
  }
  if (!c->AtEnd() && *c->p == '.') {
    ++c->p;
    double place = 0.1;
    while (!c->AtEnd() && *c->p >= '0' && *c->p <= '9') {
      v += (*c->p - '0') * place;
      place *= 0.1;
      ++c->p;
    }
  }
  *out = neg ? -v : v;
  return true;
}

bool Slurp(const char* filename, std::vector<uint8_t>* out) {
  FILE* f = fopen(filename, "rb");
  if (!f) return false;
  bool ok = fseek(f, 0, SEEK_END) == 0;
  long size = ok ? ftell(f) : -1;
  ok = ok && size >= 0 && fseek(f, 0, SEEK_SET) == 0;
  if (ok) {
    out->resize(static_cast<size_t>(size));
    ok = fread(out->data(), 1, out->size(), f) == out->size();
  }
  return (fclose(f) == 0) && ok;
}

}  // namespace

}  // namespace jxl

namespace jxlt {
// Header of a colour PFM (read_pfm.cc:27-147): "PF", sizes, scale +-1 (sign = byte order).
bool ParsePFMHeader(const uint8_t* data, size_t size, size_t* xsize_out, size_t* ysize_out, bool* big_endian,
                    size_t* payload_offset) {
  using namespace jxl;
  if (size < 2 || data[0] != 'P' || data[1] != 'F') {
    fprintf(stderr, "PFM: bad magic.\n");
    return false;
  }
  Cursor c = {data + 2, data + size};
  size_t xsize = 0, ysize = 0;
  double scale = 0;
  if (!EatOneWhitespace(&c) || !ParseSize(&c, &xsize) || !EatBlank(&c) || !ParseSize(&c, &ysize) ||
      !EatOneWhitespace(&c) || !ParseScale(&c, &scale) || !EatOneWhitespace(&c)) {
    fprintf(stderr, "PFM: malformed header.\n");
    return false;
  }
  if (scale != 1.0 && scale != -1.0) {
    fprintf(stderr, "PFM: bad scale factor value.\n");
    return false;
  }
  if (xsize == 0 || ysize == 0 || xsize > 0x3FFFFFFFull || ysize > 0x3FFFFFFFull) return false;
  const size_t need = xsize * ysize * 3 * sizeof(float);
  if (static_cast<size_t>(c.end - c.p) < need) {
    fprintf(stderr, "PFM: truncated pixel data.\n");
    return false;
  }
  *xsize_out = xsize;
  *ysize_out = ysize;
  *big_endian = scale > 0.0;
  *payload_offset = static_cast<size_t>(c.p - data);
  return true;
}
}  // namespace jxlt

namespace jxl {

bool ReadPFM(const char* filename, Image3F* image) {
  std::vector<uint8_t> data;
  if (!Slurp(filename, &data)) {
    fprintf(stderr, "Could not read %s\n", filename);
    return false;
  }
  size_t xsize = 0, ysize = 0, payload_offset = 0;
  bool big_endian = false;
  if (!jxlt::ParsePFMHeader(data.data(), data.size(), &xsize, &ysize, &big_endian, &payload_offset)) return false;
  Image3F img(xsize, ysize);
  if (!img.valid()) return false;
  const uint8_t* payload = data.data() + payload_offset;
  const size_t row_bytes = xsize * 3 * sizeof(float);
  for (size_t y = 0; y < ysize; ++y) {
    const uint8_t* row_in = payload + (ysize - 1 - y) * row_bytes;  // bottom-to-top
    float* rows[3] = {img.PlaneRow(0, y), img.PlaneRow(1, y), img.PlaneRow(2, y)};
    for (size_t x = 0; x < xsize; ++x) {
      for (size_t ch = 0; ch < 3; ++ch) {
        uint32_t u;
        memcpy(&u, row_in + (x * 3 + ch) * 4, 4);
        if (big_endian) u = __builtin_bswap32(u);
        memcpy(&rows[ch][x], &u, 4);
      }
    }
  }
  *image = std::move(img);
  return true;
}

}  // namespace jxl
