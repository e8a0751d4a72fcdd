// Probe: issue rate of v_pk_fma_f32 / v_pk_mul_f32 vs their scalar forms on gfx950.
// (build: hipcc --offload-arch=gfx950 -O2 -o tools/pk_probe tools/pk_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f2 x[8];
  float y[16];
  for (int i = 0; i < 8; i++) { x[i].x = threadIdx.x * 0.001f + i; x[i].y = threadIdx.x * 0.002f - i; }
  for (int i = 0; i < 16; i++) y[i] = threadIdx.x * 0.003f + i;
  const f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a), "v"(b));
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(av));
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
  for (int i = 0; i < 16; i++) s += y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
float run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 20000;
  const double lanes_ops = 256.0 * 8 * 256 * iters;  // threads x iterations
  const float t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters), t3 = run<3>(out, iters);
  printf("pk_fma : %.3f ms, %.2f Tflop/s (8 pk per iter)\n", t0, lanes_ops * 8 * 4 / t0 / 1e9);
  printf("fma    : %.3f ms, %.2f Tflop/s (16 scalar per iter)\n", t1, lanes_ops * 16 * 2 / t1 / 1e9);
  printf("pk_mul : %.3f ms (8 pk per iter)\n", t2);
  printf("mul    : %.3f ms (16 scalar per iter)\n", t3);
  return 0;
}
