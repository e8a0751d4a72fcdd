// Probe: is the raw v_sqrt_f32 of gfx950 correctly rounded for integer-valued inputs?
// (build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/sqrt_probe tools/sqrt_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(unsigned long long* stats, uint32_t* first_bad, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float q = (float)i;
  const float a = __builtin_amdgcn_sqrtf(q);
  const float b = sqrtf(q);
  if (a != b) {
    const unsigned long long k = atomicAdd(&stats[0], 1ull);
    if (k < 16) first_bad[k] = i;
    atomicMin(&first_bad[16], i);
    const int d = abs((int)(__float_as_uint(a) - __float_as_uint(b)));
    atomicMax(&stats[1], (unsigned long long)d);
  }
}
int main() {
  unsigned long long* stats; uint32_t* bad;
  hipMallocManaged(&stats, 16); hipMallocManaged(&bad, 128);
  for (uint32_t n : {1u << 12, 1u << 16, 1u << 20, 1u << 24}) {
    stats[0] = stats[1] = 0; bad[16] = 0xFFFFFFFFu;
    hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, stats, bad, n);
    hipDeviceSynchronize();
    printf("n=%u mismatches=%llu max_ulp=%llu first:", n, stats[0], stats[1]);
    for (int k = 0; k < 8 && k < (int)stats[0]; k++) printf(" %u", bad[k]);
    printf("  smallest mismatch: %u\n", bad[16]);
  }
  return 0;
}
