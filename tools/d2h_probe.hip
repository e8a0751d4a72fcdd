// Probe for the section hand-over: how do packed sections best reach page-locked host memory?
//  (a) hipMemcpyAsync device -> host (what rounds 1-3 used; the runtime runs it as a copy kernel of its own),
//  (b) a kernel that stores to the mapped host buffer itself -- its byte ranges can come from device memory, so the
//      host does not have to know a size before the bytes leave,
//  (c) how long the host waits for a word that a kernel stores to mapped host memory (polling) compared with
//      hipEventSynchronize / hipStreamSynchronize behind the same kernel.
// Sizes: 64 KB (a small frame's sections), 1.5 MB (4096^2), 5 MB / 20 MB (DC / AC sections of the 16384^2 bench frame).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/d2h_probe tools/d2h_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const size_t* n_ptr) {
  const size_t n = *n_ptr;  // (the size is read on the device, like a section layout would be)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// Copy + completion flag: the last workgroup to finish stores the flag behind a system-scope fence.
__global__ void copy_flag_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const size_t* n_ptr,
                                 unsigned* counter, volatile unsigned* host_flag, unsigned value) {
  const size_t n = *n_ptr;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(counter, 1u) == gridDim.x - 1) {
      *counter = 0;
      __threadfence_system();
      *host_flag = value;
    }
  }
}
__global__ void flag_kernel(volatile unsigned* host_flag, unsigned value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __threadfence_system();
    *host_flag = value;
  }
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const size_t cap = (size_t)32 << 20;
  uint8_t* h; CK(hipHostMalloc((void**)&h, cap, hipHostMallocPortable | hipHostMallocMapped));
  memset(h, 0, cap);
  unsigned* hflag; CK(hipHostMalloc((void**)&hflag, 4096, hipHostMallocPortable | hipHostMallocMapped));
  *hflag = 0;
  uint8_t* d; CK(hipMalloc((void**)&d, cap));
  CK(hipMemset(d, 0x5A, cap));
  size_t* dn; CK(hipMalloc((void**)&dn, 8));
  unsigned* dcount; CK(hipMalloc((void**)&dcount, 4)); CK(hipMemset(dcount, 0, 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint4* hd; CK(hipHostGetDevicePointer((void**)&hd, h, 0));
  unsigned* hflag_d; CK(hipHostGetDevicePointer((void**)&hflag_d, hflag, 0));
  unsigned seq = 0;
  for (size_t bytes : {(size_t)64 << 10, (size_t)1536 << 10, (size_t)5 << 20, (size_t)20 << 20}) {
    const size_t n16 = bytes / 16;
    CK(hipMemcpy(dn, &n16, 8, hipMemcpyHostToDevice));
    printf("---- %zu KB\n", bytes >> 10);
    for (int rep = 0; rep < 3; rep++) {
      float ms;
      // (a) hipMemcpyAsync
      CK(hipStreamSynchronize(s));
      double t0 = now_us();
      CK(hipEventRecord(e0, s));
      CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      double t1 = now_us();
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("hipMemcpyAsync D2H                      device %8.1f us %6.2f GB/s | host issue->sync %8.1f us\n", ms * 1e3,
             bytes / ms / 1e6, t1 - t0);
      // (b) kernel stores, several grid sizes
      for (int blocks : {64, 256, 1024}) {
        memset(h, 0, 64);
        CK(hipStreamSynchronize(s));
        t0 = now_us();
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)d, hd, (const size_t*)dn);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        t1 = now_us();
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel -> mapped host, %4d WGs          device %8.1f us %6.2f GB/s | host issue->sync %8.1f us %s\n", blocks,
               ms * 1e3, bytes / ms / 1e6, t1 - t0, h[bytes - 1] == 0x5A && h[0] == 0x5A ? "" : "DATA MISSING");
      }
      // (c) copy + flag polled by the host: is the data there when the flag is?
      memset(h, 0, bytes);
      CK(hipStreamSynchronize(s));
      seq++;
      t0 = now_us();
      hipLaunchKernelGGL(copy_flag_kernel, dim3(256), dim3(256), 0, s, (const uint4*)d, hd, (const size_t*)dn, dcount,
                         (volatile unsigned*)hflag_d, seq);
      while (*(volatile unsigned*)hflag != seq) __builtin_ia32_pause();
      t1 = now_us();
      size_t bad = 0;
      for (size_t i = 0; i < bytes; i += 64) bad += h[i] != 0x5A;
      bad += h[bytes - 1] != 0x5A;
      CK(hipStreamSynchronize(s));
      double t2 = now_us();
      printf("kernel -> mapped host + polled flag      host launch->flag %8.1f us %6.2f GB/s (sync %.1f us later) %s\n", t1 - t0,
             bytes / (t1 - t0) / 1e3, t2 - t1, bad ? "DATA NOT VISIBLE AT FLAG" : "data complete at flag");
    }
  }
  // latency of completion notification: an (almost) empty kernel
  for (int rep = 0; rep < 5; rep++) {
    CK(hipStreamSynchronize(s));
    seq++;
    double t0 = now_us();
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, (volatile unsigned*)hflag_d, seq);
    while (*(volatile unsigned*)hflag != seq) __builtin_ia32_pause();
    double t1 = now_us();
    CK(hipStreamSynchronize(s));
    double t2 = now_us();
    seq++;
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, (volatile unsigned*)hflag_d, seq);
    CK(hipStreamSynchronize(s));
    double t3 = now_us();
    seq++;
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, (volatile unsigned*)hflag_d, seq);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    double t4 = now_us();
    seq++;
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, (volatile unsigned*)hflag_d, seq);
    CK(hipEventRecord(e1, s));
    while (hipEventQuery(e1) == hipErrorNotReady) {}
    double t5 = now_us();
    printf("empty kernel: launch->polled flag %6.1f us (+%.1f us to stream sync) | launch->hipStreamSynchronize %6.1f us | "
           "launch->hipEventSynchronize %6.1f us | launch->hipEventQuery spin %6.1f us\n",
           t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
  }
  // small H2D: a 16 KB code table by hipMemcpyAsync from page-locked memory vs a kernel that reads it from mapped memory
  {
    uint32_t* htab; CK(hipHostMalloc((void**)&htab, 16384, hipHostMallocPortable | hipHostMallocMapped));
    for (int i = 0; i < 4096; i++) htab[i] = i;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipStreamSynchronize(s));
      double t0 = now_us();
      CK(hipMemcpyAsync(d, htab, 16384, hipMemcpyHostToDevice, s));
      seq++;
      hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, (volatile unsigned*)hflag_d, seq);
      double t1 = now_us();
      while (*(volatile unsigned*)hflag != seq) __builtin_ia32_pause();
      double t2 = now_us();
      printf("16 KB H2D + kernel behind it: issue %.1f us, until the kernel has run %.1f us\n", t1 - t0, t2 - t0);
    }
  }
  return 0;
}
