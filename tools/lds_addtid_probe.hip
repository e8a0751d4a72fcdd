// Does ds_write_addtid_b32 honour an M0 base at or above 64 KB?  (the gfx9 ISA text says the instruction
// takes M0[15:0] as its base; tile12_kernel's fourth pair wave stores its half-transpose rows at LDS byte offset
// 78 592 through JXLT_LDS_STORE_ROW.  If the base wrapped, those stores would land at 13 056, inside the live X plane.)
// A workgroup owns 96 KB of LDS, fills it with a pattern, every wave stores its lane ids through
// `s_mov_b32 m0, base; ds_write_addtid_b32 v, offset:imm` at bases 0 .. 92 KB (above and below 64 KB, with and
// without an immediate offset), then the whole LDS is read back with ordinary loads: the stores must be exactly where
// base + offset + 4 * lane says and nowhere else.  Prints "<n> wrong of <m>" per base; exit code 1 if any is wrong.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kLdsBytes = 96 * 1024;
constexpr int kDwords = kLdsBytes / 4;
constexpr uint32_t kFill = 0xC0FFEE00u;

template <int OFF>
__device__ inline void store_row(uint32_t base, uint32_t val) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%2" : : "s"(base), "v"(val), "n"(OFF) : "memory");
}

__global__ void __launch_bounds__(256) probe(uint32_t* out, uint32_t base, int with_offset) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < kDwords; i += 256) lds[i] = kFill;
  __syncthreads();
  // wave w stores to base + w * 512 (+ 256 through the immediate offset when asked to)
  const uint32_t wave_base = __builtin_amdgcn_readfirstlane(base + wave * 512);
  if (with_offset) store_row<256>(wave_base, 0xAB000000u | (wave << 8) | lane);
  else store_row<0>(wave_base, 0xAB000000u | (wave << 8) | lane);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < kDwords; i += 256) out[i] = lds[i];
}

int main() {
  uint32_t* d = nullptr;
  if (hipMalloc(&d, kLdsBytes) != hipSuccess) return 2;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess) return 2;
  std::vector<uint32_t> h(kDwords);
  const uint32_t bases[] = {0, 13056, 32768, 65536 - 1024, 65536, 65536 + 13056, 78592, 92 * 1024};
  int rc = 0;
  for (int with_offset = 0; with_offset < 2; with_offset++) {
    for (uint32_t base : bases) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(256), kLdsBytes, 0, d, base, with_offset);
      if (hipMemcpy(h.data(), d, kLdsBytes, hipMemcpyDeviceToHost) != hipSuccess) return 2;
      long bad = 0;
      for (int i = 0; i < kDwords; i++) {
        uint32_t want = kFill;
        const long rel = (long)i * 4 - (long)base - (with_offset ? 256 : 0);
        if (rel >= 0 && rel < 4 * 512 && (rel % 512) < 256) {
          const int wave = (int)(rel / 512), lane = (int)((rel % 512) / 4);
          want = 0xAB000000u | (wave << 8) | lane;
        }
        bad += h[i] != want;
      }
      printf("base %6u offset %3d: %ld wrong of %d\n", base, with_offset ? 256 : 0, bad, kDwords);
      if (bad) rc = 1;
    }
  }
  hipFree(d);
  return rc;
}
