// Probe: how fast a 403 MB slab (one DC-group row of the 16384^2 bench frame) crosses PCIe from page-locked host
// memory, by (a) one hipMemcpyAsync, (b) the slab split over 2 / 4 streams, (c) a kernel that reads the mapped host
// memory itself.  Decides how jxlt_image_attach_host* should fetch its rows.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/h2d_probe tools/h2d_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t bytes = (size_t)2048 * 16384 * 12, total = bytes * 4;
  float* h; CK(hipHostMalloc((void**)&h, total, hipHostMallocPortable | hipHostMallocMapped));
  memset(h, 1, total);
  float* d; CK(hipMalloc((void**)&d, total));
  hipStream_t s[4]; for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto report = [&](const char* what, float ms, size_t n) { printf("%-44s %8.2f ms  %6.2f GB/s\n", what, ms, n / ms / 1e6); };
  for (int rep = 0; rep < 2; rep++) {
    float ms;
    for (int parts : {1, 2, 4}) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s[0]));
      for (int p = 1; p < parts; p++) CK(hipStreamWaitEvent(s[p], e0, 0));
      for (int slab = 0; slab < 4; slab++)
        for (int p = 0; p < parts; p++) {
          const size_t o = slab * bytes + bytes / parts * p;
          CK(hipMemcpyAsync((char*)d + o, (char*)h + o, bytes / parts, hipMemcpyHostToDevice, s[p]));
        }
      hipEvent_t done[4];
      for (int p = 1; p < parts; p++) { CK(hipEventCreate(&done[p])); CK(hipEventRecord(done[p], s[p])); CK(hipStreamWaitEvent(s[0], done[p], 0)); }
      CK(hipEventRecord(e1, s[0])); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      char name[64]; snprintf(name, sizeof name, "4 slabs, hipMemcpyAsync over %d stream(s)", parts);
      report(name, ms, total);
    }
    for (int blocks : {256, 1024, 4096}) {
      float4* hd; CK(hipHostGetDevicePointer((void**)&hd, h, 0));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s[0]));
      hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, s[0], (const float4*)hd, (float4*)d, total / 16);
      CK(hipEventRecord(e1, s[0])); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      char name[64]; snprintf(name, sizeof name, "kernel reading mapped host memory, %d WGs", blocks);
      report(name, ms, total);
    }
  }
  return 0;
}
