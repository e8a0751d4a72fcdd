// Probe: is rcp + the refinement steps of the IEEE division expansion (without div_scale /
// div_fixup) equal to 1.0f / q for every integer-valued q the quantiser can produce?
// (build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o tools/rcp_probe tools/rcp_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ float rcp_trimmed(float q) {
  const float r0 = __builtin_amdgcn_rcpf(q);
  const float e0 = __builtin_fmaf(-q, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float q0 = r1;  // 1.0f * r1
  const float e1 = __builtin_fmaf(-q, q0, 1.0f);
  const float q1 = __builtin_fmaf(e1, r1, q0);
  const float e2 = __builtin_fmaf(-q, q1, 1.0f);
  return __builtin_fmaf(e2, r1, q1);
}
__device__ float rcp_short(float q) {  // one Newton step + one residual correction
  const float r0 = __builtin_amdgcn_rcpf(q);
  const float e0 = __builtin_fmaf(-q, r0, 1.0f);
  const float r1 = __builtin_fmaf(e0, r0, r0);
  const float e1 = __builtin_fmaf(-q, r1, 1.0f);
  return __builtin_fmaf(e1, r1, r1);
}
__device__ float rcp_shorter(float q) {  // residual correction only
  const float r0 = __builtin_amdgcn_rcpf(q);
  const float e0 = __builtin_fmaf(-q, r0, 1.0f);
  return __builtin_fmaf(e0, r0, r0);
}
template <int V>
__global__ void probe(unsigned long long* stats, int32_t* first_bad, int64_t lo, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = (int32_t)(lo + i);
  if (v == 0) return;
  const float q = (float)v;
  const float a = V == 0 ? rcp_trimmed(q) : V == 1 ? rcp_short(q) : rcp_shorter(q);
  const float b = 1.0f / q;
  if (__float_as_uint(a) != __float_as_uint(b)) {
    const unsigned long long k = atomicAdd(&stats[0], 1ull);
    if (k < 16) first_bad[k] = v;
  }
}
int main() {
  unsigned long long* stats; int32_t* bad;
  hipMallocManaged(&stats, 16); hipMallocManaged(&bad, 64);
  // every int16 value, then a wide sweep
  const int64_t ranges[3][2] = {{-32768, 65536}, {-(1 << 24), 1 << 25}, {-2147483647ll, 4294967295ll}};
  for (int variant = 0; variant < 3; variant++)
  for (auto& r : ranges) {
    stats[0] = 0;
    const int64_t n = r[1];
    printf("variant %d ", variant);
    auto kern = variant == 0 ? probe<0> : variant == 1 ? probe<1> : probe<2>;
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, stats, bad, r[0], n);
    hipDeviceSynchronize();
    printf("from %lld count %lld: mismatches=%llu first:", (long long)r[0], (long long)n, stats[0]);
    for (int k = 0; k < 8 && k < (int)stats[0]; k++) printf(" %d", bad[k]);
    printf("\n");
  }
  return 0;
}
