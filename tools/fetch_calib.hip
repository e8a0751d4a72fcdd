// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access widths tile_kernel uses
// (4 B per lane loads, 2 B per lane stores): streams a buffer larger than the 256 MiB
// Infinity Cache once, so the true HBM byte counts are known (MI355X_MICROARCH.md, HBM).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void read_dword_per_lane(const float* in, float* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (size_t k = i; k < n; k += (size_t)gridDim.x * blockDim.x) acc += in[k];
  if (acc == 12345.678f) out[0] = acc;  // never true for the zero-filled buffer
}

__global__ void write_short_per_lane(int16_t* out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t k = i; k < n; k += (size_t)gridDim.x * blockDim.x) out[k] = (int16_t)k;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float* in;
  float* out;
  int16_t* w;
  if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess ||
      hipMalloc(&w, bytes) != hipSuccess)
    return 1;
  hipMemset(in, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(read_dword_per_lane, dim3(4096), dim3(256), 0, 0, in, out, bytes / 4);
    hipLaunchKernelGGL(write_short_per_lane, dim3(4096), dim3(256), 0, 0, w, bytes / 2);
  }
  hipDeviceSynchronize();
  printf("read_dword_per_lane: %zu bytes read; write_short_per_lane: %zu bytes written\n", bytes, bytes);
  return 0;
}
