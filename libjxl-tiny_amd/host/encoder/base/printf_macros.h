// Drop-in for /root/reference/encoder/base/printf_macros.h:8-36: the printf length modifiers for
// size_t / ssize_t that callers of the reference (its own cjxl_main.cc:10,86,94) spell PRIuS / PRIdS.
// The product targets Linux + ROCm only, so the C99 `z` modifier is the one definition needed; a
// platform header that already defines either name wins.
#ifndef JXLT_HOST_ENCODER_BASE_PRINTF_MACROS_H_
#define JXLT_HOST_ENCODER_BASE_PRINTF_MACROS_H_

#ifndef PRIuS
#define PRIuS "zu"
#endif

#ifndef PRIdS
#define PRIdS "zd"
#endif

#endif  // JXLT_HOST_ENCODER_BASE_PRINTF_MACROS_H_
