// Probe (round 4): how much do a device-to-host hand-over and an HBM-bound kernel slow each other down when they run
// side by side, and does it depend on HOW the bytes are handed over?  (In the frame pipeline the AC measuring pass took
// twice as long beside the first version of pack_deliver_kernel, but not beside the runtime's copy kernel.)
//   streaming kernel: reads 2 GB of device memory (sum), ~0.5 ms alone
//   hand-over of 16 MB to page-locked host memory by
//     (a) hipMemcpyAsync                                   (b) kernel, source and destination 16-byte co-aligned
//     (c) kernel, source 3 bytes off (unaligned 16-byte loads)   (d) as (c) but realigned in registers from aligned loads
//   each with 8 / 64 / 256 workgroups of 256 / 1024 threads.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/d2h_interfere_probe tools/d2h_interfere_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ void stream_read(const v4u* __restrict__ a, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const v4u v = a[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *out = acc;
}
__global__ void copy_aligned(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void copy_unaligned_loads(const unsigned char* __restrict__ src, v4u* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    v4u v;
    __builtin_memcpy(&v, src + (i << 4), 16);
    dst[i] = v;
  }
}
// source `shift` bytes (1..15) beyond a 16-byte boundary: two aligned loads, bytes realigned in registers
__global__ void copy_realigned(const v4u* __restrict__ src_aligned, int shift, v4u* __restrict__ dst, size_t n) {
  const int sw = shift >> 2, sb = shift & 3;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const v4u a = src_aligned[i], b = src_aligned[i + 1];
    unsigned w[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, 0};
    v4u v;
    v.x = __builtin_amdgcn_alignbyte(w[sw + 1], w[sw], sb);
    v.y = __builtin_amdgcn_alignbyte(w[sw + 2], w[sw + 1], sb);
    v.z = __builtin_amdgcn_alignbyte(w[sw + 3], w[sw + 2], sb);
    v.w = __builtin_amdgcn_alignbyte(w[sw + 4], w[sw + 3], sb);
    dst[i] = v;
  }
}

int main() {
  const size_t big = (size_t)2 << 30, bytes = (size_t)16 << 20;
  v4u* a; CK(hipMalloc((void**)&a, big)); CK(hipMemset(a, 1, big));
  unsigned char* d; CK(hipMalloc((void**)&d, bytes + 64)); CK(hipMemset(d, 0x5A, bytes + 64));
  unsigned char* h; CK(hipHostMalloc((void**)&h, bytes + 64, hipHostMallocDefault));
  unsigned* out; CK(hipMalloc((void**)&out, 4));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  const size_t n16 = bytes / 16;
  auto launch_copy = [&](int kind, int wgs, int threads, hipStream_t s) {
    switch (kind) {
      case 0: (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); break;
      case 1: hipLaunchKernelGGL(copy_aligned, dim3(wgs), dim3(threads), 0, s, (const v4u*)d, (v4u*)h, n16); break;
      case 2: hipLaunchKernelGGL(copy_unaligned_loads, dim3(wgs), dim3(threads), 0, s, (const unsigned char*)d + 3, (v4u*)h, n16); break;
      case 3: hipLaunchKernelGGL(copy_realigned, dim3(wgs), dim3(threads), 0, s, (const v4u*)d, 3, (v4u*)h, n16); break;
      case 4: (void)hipMemcpyAsync(h, d + 3, bytes, hipMemcpyDeviceToHost, s); break;
    }
  };
  const char* names[5] = {"hipMemcpyAsync (aligned)", "kernel, co-aligned", "kernel, unaligned 16-B loads", "kernel, realigned in registers",
                          "hipMemcpyAsync (source 3 bytes off)"};
  float stream_alone = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a0, s0));
    hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, s0, (const v4u*)a, big / 16, out);
    CK(hipEventRecord(a1, s0)); CK(hipEventSynchronize(a1)); CK(hipEventElapsedTime(&stream_alone, a0, a1));
  }
  printf("streaming kernel alone: %.3f ms (%.0f GB/s)\n", stream_alone, big / stream_alone / 1e6);
  struct Cfg { int kind, wgs, threads; };
  std::vector<Cfg> cfgs = {{0, 0, 0}, {4, 0, 0}};
  for (int kind = 1; kind <= 3; kind++)
    for (int wgs : {8, 64, 256})
      for (int threads : {256, 1024}) cfgs.push_back({kind, wgs, threads});
  for (const Cfg& c : cfgs) {
    float alone = 0, beside_copy = 0, beside_stream = 0;
    for (int rep = 0; rep < 3; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(b0, s1)); launch_copy(c.kind, c.wgs, c.threads, s1); CK(hipEventRecord(b1, s1));
      CK(hipEventSynchronize(b1)); CK(hipEventElapsedTime(&alone, b0, b1));
      CK(hipDeviceSynchronize());
      // side by side: the streaming kernel twice in a row (1 ms) so that the hand-over lies inside it
      CK(hipEventRecord(a0, s0));
      hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, s0, (const v4u*)a, big / 16, out);
      hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, s0, (const v4u*)a, big / 16, out);
      CK(hipEventRecord(a1, s0));
      CK(hipEventRecord(b0, s1)); launch_copy(c.kind, c.wgs, c.threads, s1); CK(hipEventRecord(b1, s1));
      CK(hipEventSynchronize(a1)); CK(hipEventSynchronize(b1));
      CK(hipEventElapsedTime(&beside_stream, a0, a1)); CK(hipEventElapsedTime(&beside_copy, b0, b1));
    }
    printf("%-36s %4d x %4d | alone %.3f ms %5.1f GB/s | beside the stream %.3f ms %5.1f GB/s | 2 x stream beside it %.3f ms (alone %.3f)\n",
           names[c.kind], c.wgs, c.threads, alone, bytes / alone / 1e6, beside_copy, bytes / beside_copy / 1e6, beside_stream,
           2 * stream_alone);
  }
  return 0;
}
