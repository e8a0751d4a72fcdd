// Drop-in for the part of /root/reference/encoder/base/status.h:258-296 that callers of
// EncodeFrame see: jxl::Status (a bool-like result that can carry a code), jxl::StatusCode and
// JXL_RETURN_IF_ERROR.  Callers written against the reference -- `Status s = EncodeFrame(...)`,
// `if (!s)`, `s.code()`, `s.IsFatalError()`, `JXL_RETURN_IF_ERROR(EncodeFrame(...))` -- compile
// unchanged.  The reference's debug-message / abort machinery is not part of the surface.
#ifndef JXLT_HOST_ENCODER_BASE_STATUS_H_
#define JXLT_HOST_ENCODER_BASE_STATUS_H_

#include <stdint.h>

namespace jxl {

enum class StatusCode : int32_t {
  kNotEnoughBytes = -1,  // non-fatal (negative)
  kOk = 0,
  kGenericError = 1,  // fatal (positive)
};

class [[nodiscard]] Status {
 public:
  constexpr Status(bool ok) : code_(ok ? StatusCode::kOk : StatusCode::kGenericError) {}  // NOLINT: implicit by design
  constexpr Status(StatusCode code) : code_(code) {}                                      // NOLINT
  constexpr operator bool() const { return code_ == StatusCode::kOk; }                    // NOLINT
  constexpr StatusCode code() const { return code_; }
  constexpr bool IsFatalError() const { return static_cast<int32_t>(code_) > 0; }

 private:
  StatusCode code_;
};

}  // namespace jxl

#ifndef JXL_RETURN_IF_ERROR
#define JXL_RETURN_IF_ERROR(status)                     \
  do {                                                  \
    ::jxl::Status jxl_return_if_error_status = (status); \
    if (!jxl_return_if_error_status) return jxl_return_if_error_status; \
  } while (0)
#endif

#endif  // JXLT_HOST_ENCODER_BASE_STATUS_H_
