// tile_kernel divides with the hardware reciprocal and the refinement steps of the generic IEEE
// expansion, without that expansion's operand scaling and special-case fix-up (jxlt_device.h:
// div_normal) -- valid where operands and quotient are far from the overflow / underflow
// thresholds, which is where every division of the kernel lives (denominators 1e-3 .. 1e6 for
// finite input of ordinary magnitude).  This probe compares it with the compiler's IEEE division on
// 2^32 pseudo-random operand pairs, magnitudes log-uniform in [2^-40, 2^40], both signs of the
// numerator, plus reciprocals (numerator 1).  Prints the number of mismatches (must be 0).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float div_normal(float num, float den) {
  const float r0 = __builtin_amdgcn_rcpf(den);
  const float e0 = __builtin_fmaf(-den, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float q0 = num * r1;
  const float e1 = __builtin_fmaf(-den, q0, num);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-den, q1, num);
  return __builtin_fmaf(e2, r1, q1);
}

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// float with a random mantissa and an exponent in [-40, 40]
__device__ __forceinline__ float pick(uint32_t h) {
  const uint32_t e = 127u - 40u + (h >> 23) % 81u;
  return __uint_as_float((e << 23) | (h & 0x7FFFFFu));
}

__global__ void probe(unsigned long long* bad, unsigned long long* bad_rcp, uint32_t seed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long b = 0, br = 0;
  for (uint32_t k = 0; k < 1024; k++) {
    const uint32_t h1 = mix(i * 1024u + k + seed), h2 = mix(h1 ^ 0x9e3779b9u);
    const float den = pick(h1);
    float num = pick(h2);
    if (h2 & 0x80000000u) num = -num;
    const float want = num / den, got = div_normal(num, den);
    b += __float_as_uint(want) != __float_as_uint(got);
    br += __float_as_uint(1.0f / den) != __float_as_uint(div_normal(1.0f, den));
  }
  if (b) atomicAdd(bad, b);
  if (br) atomicAdd(bad_rcp, br);
}

int main() {
  unsigned long long *d, h[2] = {0, 0};
  hipMalloc(&d, 16);
  hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(16384), dim3(256), 0, 0, d, d + 1, 12345u);  // 2^32 pairs
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("div_normal vs IEEE division: mismatches=%llu of 4294967296; reciprocals: mismatches=%llu\n", h[0], h[1]);
  return (h[0] || h[1]) ? 1 : 0;
}
