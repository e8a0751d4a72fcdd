// jxlt_publish_kernel.h -- publish_kernel: small results from device memory to the host's page-locked memory + a
// sequence word the host polls.  Part of jxlt_device.h (include that one, or -- the product's translation units --
// exactly the kernel headers a unit launches).
#ifndef JXLT_PUBLISH_KERNEL_H_
#define JXLT_PUBLISH_KERNEL_H_

#include "jxlt_device_common.h"

#ifndef JXLT_THREADFENCE_SYSTEM
#define JXLT_THREADFENCE_SYSTEM() __threadfence_system()
#endif

namespace jxlt_dev {

// ---------------------------------------------------------------------------
// Hand-over to the host without the host (round 4).
//
// Until round 3 every result the host waited for came through hipMemcpyAsync + an event: ~20 us of device time for a
// 16 KB download (tools/d2h_probe.hip: the runtime's copy kernel) and ~12 us for the host to notice the event, six
// to eight times per frame; and the section bytes could only leave once the HOST had read their sizes and issued the
// copies.  Now a kernel stores the small results to the host's page-locked memory itself:
//   publish_kernel        histograms, counts, section sizes + a sequence word the host polls
// A 16 KB publish takes ~6 us and its flag is seen ~6 us after the launch.  (The section BYTES travel by copy
// commands: a kernel that stored them to host memory -- pack_deliver_kernel, round 4 -- and an HBM-bound kernel beside
// it slowed each other down by 20-60 %, a DMA copy does not; DESIGN.md 4.5.1, tools/d2h_interfere_probe.hip.  The
// kernel went in round 5.)
// ---------------------------------------------------------------------------

// (four waves: a workgroup that fits whatever is free on a CU that another kernel fills)
constexpr int kPublishThreads = 256;
constexpr int kPublishSegments = 4;
struct PublishArgs {
  const uint32_t* src[kPublishSegments];  // device memory, dword granular
  uint32_t* dst[kPublishSegments];        // page-locked host memory (mapped)
  uint32_t words[kPublishSegments];
  // optional: one 64-bit word of device memory copied behind the segments (a total the host wants with them)
  const unsigned long long* src64;
  unsigned long long* dst64;
  uint32_t* flag;   // host memory: receives `seq` when everything above is visible to the host
  uint32_t* flag2;  // optional: a second word that receives `seq` (one publication that stands for two)
  uint32_t seq;
};
// ONE workgroup (the payloads are a few KB to a few hundred KB): no cross-workgroup completion protocol.
__global__ void __launch_bounds__(kPublishThreads) publish_kernel(const PublishArgs A) {
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < kPublishSegments; k++) {
    const uint32_t n = A.words[k];
    const uint32_t* src = A.src[k];
    uint32_t* dst = A.dst[k];
    // 16 bytes per lane where both sides allow it (segments start 16-byte aligned as a rule), dwords for the rest
    const bool wide = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    const uint32_t n4 = wide ? n >> 2 : 0;
    const uint4* src4 = reinterpret_cast<const uint4*>(src);
    uint4* dst4 = reinterpret_cast<uint4*>(dst);
    for (uint32_t i = tid; i < n4; i += kPublishThreads) dst4[i] = src4[i];
    for (uint32_t i = 4 * n4 + tid; i < n; i += kPublishThreads) dst[i] = src[i];
  }
  if (tid == 0 && A.src64) *A.dst64 = *A.src64;
  JXLT_THREADFENCE_SYSTEM();
  __syncthreads();
  if (tid == 0 && A.flag) {
    *(volatile uint32_t*)A.flag = A.seq;
    if (A.flag2) *(volatile uint32_t*)A.flag2 = A.seq;
  }
}

}  // namespace jxlt_dev

#endif  // JXLT_PUBLISH_KERNEL_H_
