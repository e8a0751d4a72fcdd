// Probe: issue cost (shader cycles per wave64 instruction, 8 waves per SIMD, 16 independent
// chains per wave) of the VALU instructions tile_kernel is made of, on this GPU.
// (build: hipcc --offload-arch=gfx950 -O2 -o tools/op_probe tools/op_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS16(STR) \
  asm volatile(STR : "+v"(y[0]) : "v"(a), "v"(b), "v"(y[15])); asm volatile(STR : "+v"(y[1]) : "v"(a), "v"(b), "v"(y[0])); \
  asm volatile(STR : "+v"(y[2]) : "v"(a), "v"(b), "v"(y[1])); asm volatile(STR : "+v"(y[3]) : "v"(a), "v"(b), "v"(y[2])); \
  asm volatile(STR : "+v"(y[4]) : "v"(a), "v"(b), "v"(y[3])); asm volatile(STR : "+v"(y[5]) : "v"(a), "v"(b), "v"(y[4])); \
  asm volatile(STR : "+v"(y[6]) : "v"(a), "v"(b), "v"(y[5])); asm volatile(STR : "+v"(y[7]) : "v"(a), "v"(b), "v"(y[6])); \
  asm volatile(STR : "+v"(y[8]) : "v"(a), "v"(b), "v"(y[7])); asm volatile(STR : "+v"(y[9]) : "v"(a), "v"(b), "v"(y[8])); \
  asm volatile(STR : "+v"(y[10]) : "v"(a), "v"(b), "v"(y[9])); asm volatile(STR : "+v"(y[11]) : "v"(a), "v"(b), "v"(y[10])); \
  asm volatile(STR : "+v"(y[12]) : "v"(a), "v"(b), "v"(y[11])); asm volatile(STR : "+v"(y[13]) : "v"(a), "v"(b), "v"(y[12])); \
  asm volatile(STR : "+v"(y[14]) : "v"(a), "v"(b), "v"(y[13])); asm volatile(STR : "+v"(y[15]) : "v"(a), "v"(b), "v"(y[14]));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  asm volatile("s_mov_b64 s[20:21], 0x33333333" ::: "s20", "s21");
  float y[16];
  for (int i = 0; i < 16; i++) y[i] = threadIdx.x * 0.003f + i + 1.0f;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) { OPS16("v_fma_f32 %0, %0, %1, %2") }
    if (MODE == 1) { OPS16("v_mul_f32 %0, %0, %1") }
    if (MODE == 2) { OPS16("v_add_f32 %0, %0, %1") }
    if (MODE == 3) { OPS16("v_mov_b32 %0, %3") }
    if (MODE == 4) { OPS16("v_cndmask_b32 %0, %0, %1, vcc") }
    if (MODE == 5) { OPS16("v_mov_b32_dpp %0, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
    if (MODE == 6) { OPS16("v_cndmask_b32_dpp %0, %3, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
    if (MODE == 7) { OPS16("v_add_u32 %0, %0, %1") }
    if (MODE == 8) { OPS16("v_rndne_f32 %0, %0") }
    if (MODE == 9) { OPS16("v_sqrt_f32 %0, %0") }
    if (MODE == 10) { OPS16("v_rcp_f32 %0, %0") }
    if (MODE == 11) { OPS16("v_cmp_gt_f32 vcc, %0, %1") }
    if (MODE == 12) { OPS16("v_mov_b32_dpp %0, %3 row_shr:4 row_mask:0xf bank_mask:0xa") }
    if (MODE == 13) { OPS16("v_mul_lo_u32 %0, %0, %1") }
    if (MODE == 14) { OPS16("v_max_f32 %0, %0, %1") }
    if (MODE == 15) { OPS16("v_and_b32 %0, %0, %1") }
    if (MODE == 16) { OPS16("v_cndmask_b32_e64 %0, %0, %1, s[20:21]") }
    if (MODE == 17) { asm volatile("s_mov_b64 vcc, 0x55555555"); OPS16("v_cndmask_b32 %0, %0, %1, vcc") }
    if (MODE == 18) { OPS16("v_cndmask_b32_e64 %0, %1, %3, s[20:21]") }
    if (MODE == 19) { OPS16("v_min_f32 %0, %0, %1") }
    if (MODE == 20) { OPS16("v_sub_f32 %0, %0, %1") }
    if (MODE == 21) { OPS16("v_cvt_f32_i32 %0, %0") }
    if (MODE == 22) { OPS16("v_lshlrev_b32 %0, 1, %0") }
    if (MODE == 23) { OPS16("v_fmac_f32 %0, %1, %2") }
    if (MODE == 24) { OPS16("v_add_f32 %0, |%0|, %1") }
    if (MODE == 25) { OPS16("v_bfe_u32 %0, %0, 3, 5") }
    if (MODE == 31) {
      // transposes' pattern: one mask, four fused select+DPP, four times per iteration
      for (int g = 0; g < 4; g++) {
        asm volatile("v_cmp_gt_f32 vcc, %4, %5\n\t"
                     "v_cndmask_b32_dpp %0, %3, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %1, %0, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %2, %1, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %3, %2, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                     : "+v"(y[4 * g]), "+v"(y[4 * g + 1]), "+v"(y[4 * g + 2]), "+v"(y[4 * g + 3]) : "v"(a), "v"(b) : "vcc");
      }
    }
    if (MODE == 32) {
      for (int g = 0; g < 4; g++) {
        asm volatile("s_mov_b64 vcc, s[20:21]\n\t"
                     "v_cndmask_b32_dpp %0, %3, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %1, %0, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %2, %1, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_cndmask_b32_dpp %3, %2, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                     : "+v"(y[4 * g]), "+v"(y[4 * g + 1]), "+v"(y[4 * g + 2]), "+v"(y[4 * g + 3]) : "v"(a), "v"(b) : "vcc");
      }
    }
    if (MODE == 26) { OPS16("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc") }
    if (MODE == 27) { OPS16("v_cmp_gt_f32 s[22:23], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[22:23]") }
    if (MODE == 28) { OPS16("v_cndmask_b32_e64 %0, %0, %1, vcc") }
    if (MODE == 29) { OPS16("v_addc_co_u32 %0, vcc, %0, %1, vcc") }
    if (MODE == 30) { OPS16("v_cndmask_b32_dpp %0, %3, %1, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1") }
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(float* out, const char* name) {
  const int iters = 8000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // 8 waves per SIMD (2048 workgroups x 4 waves / 1024 SIMDs), 16 instructions per iteration
  const double instr_per_simd = 8.0 * iters * 16;
  printf("%-42s %7.3f ms  %5.2f cycles/instr at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0>(out, "v_fma_f32"); run<1>(out, "v_mul_f32"); run<2>(out, "v_add_f32"); run<3>(out, "v_mov_b32");
  run<4>(out, "v_cndmask_b32"); run<5>(out, "v_mov_b32_dpp quad_perm"); run<6>(out, "v_cndmask_b32_dpp quad_perm");
  run<12>(out, "v_mov_b32_dpp row_shr:4 bank_mask"); run<7>(out, "v_add_u32"); run<15>(out, "v_and_b32");
  run<14>(out, "v_max_f32"); run<8>(out, "v_rndne_f32"); run<11>(out, "v_cmp_gt_f32"); run<13>(out, "v_mul_lo_u32");
  run<9>(out, "v_sqrt_f32"); run<10>(out, "v_rcp_f32");
  run<16>(out, "v_cndmask_b32_e64 sgpr mask"); run<17>(out, "v_cndmask_b32 vcc (s_mov before)"); run<18>(out, "v_cndmask_b32_e64 other srcs");
  run<19>(out, "v_min_f32"); run<20>(out, "v_sub_f32"); run<21>(out, "v_cvt_f32_i32"); run<22>(out, "v_lshlrev_b32");
  run<23>(out, "v_fmac_f32"); run<24>(out, "v_add_f32 with abs"); run<25>(out, "v_bfe_u32");
  run<26>(out, "pair: v_cmp vcc + v_cndmask vcc (x2 instr)"); run<27>(out, "pair: v_cmp sgpr + v_cndmask_e64 (x2 instr)");
  run<31>(out, "4x(v_cmp vcc + 4 cndmask_dpp) per 16 slots"); run<32>(out, "4x(s_mov vcc + 4 cndmask_dpp) per 16 slots");
  run<28>(out, "v_cndmask_b32_e64 with vcc operand"); run<29>(out, "v_addc_co_u32 vcc"); run<30>(out, "v_cndmask_b32_dpp row_shr");
  return 0;
}
