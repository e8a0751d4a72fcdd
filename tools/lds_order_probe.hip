// Does a wave see its own LDS stores in program order without s_waitcnt, for the two transpose
// layouts of tile_kernel?  A: eight dword stores, two 16-byte loads (the one in use).
// B: two 16-byte stores, eight dword loads.  Each with and without an explicit s_waitcnt between
// the stores and the loads.  Prints the number of wrong elements per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VARIANT, bool WAIT>
__global__ void probe(const float* in, float* out, int iters) {
  __shared__ float lds[64 * 72 * 2];
  const int tid = threadIdx.x, l = tid & 7, oct = tid >> 3;
  float* sc = lds + oct * 72;
  float v[8];
  for (int j = 0; j < 8; j++) v[j] = in[(blockIdx.x * 512 + tid) * 8 + j];
  for (int it = 0; it < iters; it++) {
    if (VARIANT == 0) {
      float* w = sc + (l >> 2) * 36 + (l & 3);
#pragma unroll
      for (int j = 0; j < 8; j++) w[j * 4] = v[j];
      asm volatile("" ::: "memory");
      if (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float4 a = *reinterpret_cast<const float4*>(sc + l * 4);
      const float4 b = *reinterpret_cast<const float4*>(sc + 36 + l * 4);
      asm volatile("" ::: "memory");
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      float* w = sc + l * 8 + (l >> 2) * 4;
      float4 lo, hi;
      lo.x = v[0]; lo.y = v[1]; lo.z = v[2]; lo.w = v[3];
      hi.x = v[4]; hi.y = v[5]; hi.z = v[6]; hi.w = v[7];
      *reinterpret_cast<float4*>(w) = lo;
      *reinterpret_cast<float4*>(w + 4) = hi;
      asm volatile("" ::: "memory");
      if (WAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const volatile float* rd = sc + l;
      v[0] = rd[0]; v[1] = rd[8]; v[2] = rd[16]; v[3] = rd[24];
      v[4] = rd[36]; v[5] = rd[44]; v[6] = rd[52]; v[7] = rd[60];
      asm volatile("" ::: "memory");
    }
    // make the next round depend on this one without changing the values
    for (int j = 0; j < 8; j++) v[j] = v[j] + 0.0f;
  }
  for (int j = 0; j < 8; j++) out[(blockIdx.x * 512 + tid) * 8 + j] = v[j];
}

template <int VARIANT, bool WAIT>
int run(const char* name) {
  const int blocks = 2048, n = blocks * 512 * 8, iters = 9;  // odd: result = one transpose
  std::vector<float> h(n), o(n);
  for (int i = 0; i < n; i++) h[i] = (float)(i % 100003) * 0.25f;
  float *din, *dout;
  hipMalloc(&din, n * 4);
  hipMalloc(&dout, n * 4);
  hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((probe<VARIANT, WAIT>), dim3(blocks), dim3(512), 0, 0, din, dout, iters);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int t = 0; t < blocks * 512; t++) {
    const int l = t & 7, base = (t & ~7);
    for (int j = 0; j < 8; j++) bad += o[t * 8 + j] != h[(base + j) * 8 + l];
  }
  printf("%s: %ld wrong of %d\n", name, bad, n);
  hipFree(din);
  hipFree(dout);
  return bad != 0;
}

int main() {
  int rc = 0;
  rc |= run<0, false>("A (dword stores, 16-byte loads), no wait");
  rc |= run<0, true>("A, s_waitcnt between");
  rc |= run<1, false>("B (16-byte stores, dword loads), no wait");
  rc |= run<1, true>("B, s_waitcnt between");
  return rc;
}
