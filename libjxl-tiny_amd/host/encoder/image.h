// Planar image containers for the drop-in API.  Minimal counterpart of the
// reference's Plane<T>/Image3<T> (/root/reference/encoder/image.h:143-403):
// rows are 64-byte aligned and separated by bytes_per_row(); an Image3 is three
// separately allocated planes of identical geometry.  Only what callers of
// ReadPFM/EncodeFile/EncodeFrame touch is provided.
#ifndef JXLT_HOST_ENCODER_IMAGE_H_
#define JXLT_HOST_ENCODER_IMAGE_H_

// The reference's image.h:12-15 brings <inttypes.h> and <string.h> to everything that includes it
// (its cjxl_main.cc calls strcmp / strerror on the strength of that): kept, callers compile unchanged.
#include <inttypes.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <utility>

extern "C" {
void* jxlt_pinned_alloc(size_t bytes);  // libjxltiny_hip.so (include/jxl_tiny_amd.h)
void jxlt_pinned_free(void* p);
}

namespace jxl {

template <typename T>
class Plane {
 public:
  Plane() = default;
  Plane(size_t xsize, size_t ysize) : xsize_(xsize), ysize_(ysize) {
    // Row pitch: whole 128-byte units, at least one.
    bytes_per_row_ = ((xsize * sizeof(T) + 127) / 128) * 128;
    if (bytes_per_row_ == 0) bytes_per_row_ = 128;
    const size_t total = bytes_per_row_ * (ysize ? ysize : 1);
    // Page-locked when a GPU is present (EncodeFile's upload then needs no staging copy).
    void* p = jxlt_pinned_alloc(total);
    pinned_ = p != nullptr;
    if (!p && posix_memalign(&p, 64, total) != 0) p = nullptr;
    bytes_ = static_cast<uint8_t*>(p);
  }
  ~Plane() { Release(); }
  Plane(Plane&& o) noexcept { *this = std::move(o); }
  Plane& operator=(Plane&& o) noexcept {
    if (this != &o) {
      Release();
      xsize_ = o.xsize_;
      ysize_ = o.ysize_;
      bytes_per_row_ = o.bytes_per_row_;
      bytes_ = o.bytes_;
      pinned_ = o.pinned_;
      o.bytes_ = nullptr;
      o.xsize_ = o.ysize_ = o.bytes_per_row_ = 0;
    }
    return *this;
  }
  Plane(const Plane&) = delete;
  Plane& operator=(const Plane&) = delete;

  size_t xsize() const { return xsize_; }
  size_t ysize() const { return ysize_; }
  size_t bytes_per_row() const { return bytes_per_row_; }
  intptr_t PixelsPerRow() const { return static_cast<intptr_t>(bytes_per_row_ / sizeof(T)); }
  T* Row(size_t y) { return reinterpret_cast<T*>(bytes_ + y * bytes_per_row_); }
  const T* Row(size_t y) const { return reinterpret_cast<const T*>(bytes_ + y * bytes_per_row_); }
  const T* ConstRow(size_t y) const { return Row(y); }
  bool valid() const { return bytes_ != nullptr; }

 private:
  void Release() {
    if (bytes_) {
      if (pinned_) jxlt_pinned_free(bytes_);
      else free(bytes_);
    }
    bytes_ = nullptr;
  }
  size_t xsize_ = 0, ysize_ = 0, bytes_per_row_ = 0;
  uint8_t* bytes_ = nullptr;
  bool pinned_ = false;
};

using ImageF = Plane<float>;

template <typename T>
class Image3 {
 public:
  Image3() = default;
  Image3(size_t xsize, size_t ysize)
      : planes_{jxl::Plane<T>(xsize, ysize), jxl::Plane<T>(xsize, ysize),
                jxl::Plane<T>(xsize, ysize)} {}
  Image3(Image3&&) = default;
  Image3& operator=(Image3&&) = default;

  size_t xsize() const { return planes_[0].xsize(); }
  size_t ysize() const { return planes_[0].ysize(); }
  size_t bytes_per_row() const { return planes_[0].bytes_per_row(); }
  intptr_t PixelsPerRow() const { return planes_[0].PixelsPerRow(); }
  T* PlaneRow(size_t c, size_t y) { return planes_[c].Row(y); }
  const T* PlaneRow(size_t c, size_t y) const { return planes_[c].Row(y); }
  const T* ConstPlaneRow(size_t c, size_t y) const { return planes_[c].Row(y); }
  const jxl::Plane<T>& plane(size_t c) const { return planes_[c]; }
  bool valid() const { return planes_[0].valid() && planes_[1].valid() && planes_[2].valid(); }

 private:
  jxl::Plane<T> planes_[3];
};

using Image3F = Image3<float>;

}  // namespace jxl

#endif  // JXLT_HOST_ENCODER_IMAGE_H_
