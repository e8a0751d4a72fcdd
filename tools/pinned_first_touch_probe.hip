// Probe: what does the device's FIRST copy into a part of a page-locked host buffer cost, and what makes it cheap?
//   hipHostMalloc 96 MB; 11 MB device -> host copies into [0, 11 MB) twice, then into [40 MB, 51 MB) twice -- timed on the
//   host from issue to completion -- for a buffer (a) left alone, (b) written once by the CPU (memset), (c) read once by
//   the CPU, (d) written once by the DEVICE (hipMemsetAsync over the whole buffer) before the copies.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/pinned_first_touch_probe tools/pinned_first_touch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <utility>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t cap = (size_t)96 << 20, bytes = (size_t)11 << 20;
  uint8_t* d; CK(hipMalloc((void**)&d, bytes)); CK(hipMemset(d, 0x5A, bytes));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const char* names[4] = {"left alone", "memset by the CPU", "read by the CPU", "hipMemsetAsync by the device"};
  for (int mode = 0; mode < 4; mode++) {
    uint8_t* h; CK(hipHostMalloc((void**)&h, cap, hipHostMallocDefault));
    double t_prep = now_us();
    if (mode == 1) memset(h, 0, cap);
    if (mode == 2) { volatile uint8_t sink = 0; for (size_t i = 0; i < cap; i += 4096) sink += h[i]; (void)sink; }
    if (mode == 3) { CK(hipMemsetAsync(h, 0, cap, s)); CK(hipStreamSynchronize(s)); }
    t_prep = now_us() - t_prep;
    printf("%-30s (preparation %.1f ms):", names[mode], t_prep / 1e3);
    for (size_t off : {(size_t)0, (size_t)0, (size_t)40 << 20, (size_t)40 << 20, (size_t)80 << 20}) {
      CK(hipStreamSynchronize(s));
      const double t0 = now_us();
      CK(hipMemcpyAsync(h + off, d, bytes, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      printf("  +%2zu MB: %7.3f ms", off >> 20, (now_us() - t0) / 1e3);
    }
    printf("\n");
    CK(hipHostFree(h));
  }
  // Two copies at the same time on two streams (what a frame does when its DC-group sections and its AC sections leave
  // side by side): the first such pair, the second, then three streams.
  {
    uint8_t* h; CK(hipHostMalloc((void**)&h, cap, hipHostMallocDefault));
    memset(h, 0, cap);
    uint8_t* d2; CK(hipMalloc((void**)&d2, bytes)); CK(hipMemset(d2, 0x33, bytes));
    hipStream_t s2, s3; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now_us();
      CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
      CK(hipMemcpyAsync(h + (40 << 20), d2, bytes, hipMemcpyDeviceToHost, s2));
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
      printf("two copies side by side, pair %d: %7.3f ms\n", rep, (now_us() - t0) / 1e3);
    }
    for (int rep = 0; rep < 2; rep++) {
      CK(hipDeviceSynchronize());
      const double t0 = now_us();
      CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
      CK(hipMemcpyAsync(h + (40 << 20), d2, bytes, hipMemcpyDeviceToHost, s2));
      CK(hipMemcpyAsync(h + (60 << 20), d2, bytes, hipMemcpyDeviceToHost, s3));
      CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s3));
      printf("three copies side by side, set %d: %7.3f ms\n", rep, (now_us() - t0) / 1e3);
    }
    // source / destination byte alignments a frame's copies see (offsets into the blob and into the output buffer are
    // whatever the sections' sizes make them)
    uint8_t* dbig; CK(hipMalloc((void**)&dbig, bytes + 4096)); CK(hipMemset(dbig, 0x44, bytes + 4096));
    for (int pass = 0; pass < 2; pass++)
      for (auto so_do : {std::pair<int,int>{0, 0}, {4, 4}, {1, 1}, {3, 5}, {16, 48}, {2, 0}, {0, 7}, {64, 1}, {255, 129}}) {
        CK(hipDeviceSynchronize());
        const double t0 = now_us();
        CK(hipMemcpyAsync(h + 4096 + so_do.second, dbig + so_do.first, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("pass %d, source +%3d destination +%3d: %7.3f ms\n", pass, so_do.first, so_do.second, (now_us() - t0) / 1e3);
      }
    // small copies, then a large one for the first time on a stream that has only seen small ones
    hipStream_t s4; CK(hipStreamCreateWithFlags(&s4, hipStreamNonBlocking));
    for (size_t n : {(size_t)4096, (size_t)65536, (size_t)1 << 20, (size_t)4 << 20, bytes, bytes}) {
      CK(hipDeviceSynchronize());
      const double t0 = now_us();
      CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s4));
      CK(hipStreamSynchronize(s4));
      printf("fresh stream, %8zu bytes: %7.3f ms\n", n, (now_us() - t0) / 1e3);
    }
  }
  return 0;
}
