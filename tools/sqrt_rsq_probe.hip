// Probe: is the reciprocal-square-root sequence (v_rsq_f32 + two Goldschmidt-style steps + a residual correction:
// one transcendental, two multiplications, five fused multiply-adds, no compare, no select) the correctly rounded
// square root for every float between 2^-27 and 2^63?  Compared with hipcc's IEEE sqrtf, all 90 x 2^23 patterns, and
// with sqrt_exact_midrange's neighbour test (jxlt_device_common.h), which the kernels use today.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/sqrt_rsq_probe tools/sqrt_rsq_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ float sqrt_rsq(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  float g = x * r;
  float h = 0.5f * r;
  const float e = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, e, g);
  h = __builtin_fmaf(h, e, h);
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
__device__ float sqrt_neighbours(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
  const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
  float r = (r_dn <= 0.0f) ? s_dn : s;
  return (r_up > 0.0f) ? s_up : r;
}
__global__ void probe(unsigned long long* stats, uint32_t* bad, uint32_t first, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t bits = first + (uint32_t)i;
  const float x = __uint_as_float(bits);
  const float want = __builtin_sqrtf(x);
  if (__float_as_uint(sqrt_rsq(x)) != __float_as_uint(want)) {
    const unsigned long long k = atomicAdd(&stats[0], 1ull);
    if (k < 8) bad[k] = bits;
  }
  if (__float_as_uint(sqrt_neighbours(x)) != __float_as_uint(want)) atomicAdd(&stats[1], 1ull);
}
int main() {
  unsigned long long* stats;
  uint32_t* bad;
  hipMallocManaged(&stats, 16);
  hipMallocManaged(&bad, 32);
  stats[0] = stats[1] = 0;
  const uint32_t first = 100u << 23;               // 2^-27
  const uint64_t n = (uint64_t)(190 - 100) << 23;  // .. 2^63
  hipLaunchKernelGGL(probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, stats, bad, first, n);
  hipDeviceSynchronize();
  printf("rsq sequence: mismatches=%llu of %llu; neighbour test: mismatches=%llu; first:", stats[0], (unsigned long long)n, stats[1]);
  for (int k = 0; k < 8 && k < (int)stats[0]; k++) printf(" 0x%08x", bad[k]);
  printf("\n");
  return stats[0] == 0 ? 0 : 1;
}
