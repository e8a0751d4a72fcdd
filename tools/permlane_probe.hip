// Probe: what v_permlane16_swap_b32 / v_permlane32_swap_b32 do on gfx950 with both operands the same register
// (the chroma-from-luma relay of tile_kernel moves a 16-lane row of accumulators to the next row with them).
// Prints, per 16-lane row of the two results, which source row it holds.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/permlane_probe tools/permlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned v = lane;  // value = lane id
  auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  out[lane] = r16[0];
  out[64 + lane] = r16[1];
  out[128 + lane] = r32[0];
  out[192 + lane] = r32[1];
}
int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap result[0]", "permlane16_swap result[1]", "permlane32_swap result[0]",
                          "permlane32_swap result[1]"};
  for (int r = 0; r < 4; r++) {
    printf("%s: rows hold source rows", names[r]);
    for (int row = 0; row < 4; row++) {
      const unsigned first = h[r * 64 + row * 16];
      bool whole = true;
      for (int i = 0; i < 16; i++) whole &= h[r * 64 + row * 16 + i] == first + i;
      printf(" %u%s", first / 16, whole && first % 16 == 0 ? "" : "?");
    }
    printf("\n");
  }
  return 0;
}
