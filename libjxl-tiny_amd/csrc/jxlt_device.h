// jxlt_device.h -- gfx950 device code of the JPEG XL tiny encoder hot path.
//
// Pure device code (kernels + __device__ helpers); the launches live in the
// jxlt_capi_*.hip units, each of which includes exactly the kernel headers it launches.  tests/ compile this same file against a fiber-based HIP
// execution model on the CPU (tests/hipsim) to check it bit-for-bit against the
// oracle without a GPU; the product only ever builds it with hipcc.
//
// MUST be compiled with -ffp-contract=off: the arithmetic model is "every
// operation one IEEE binary32 operation, fused only where __builtin_fmaf is
// written" (SURVEY.md Appendix B, 8-lane canonical model).  The 8 SIMD lanes of
// the reference map onto 8 adjacent GPU lanes ("octets"); SumOfLanes becomes a
// 3-step xor butterfly inside the octet, which reproduces the halving tree.
//
// Files (this one includes them all):
//   jxlt_device_common.h  kernel argument blocks, arithmetic primitives, register-held DCTs, AQ helpers
//   jxlt_tile_kernel.h    tile12_kernel: one 768-thread workgroup (twelve waves) per 64x64 tile: edge-replicated
//                         load + XYB -> LDS, adaptive quant field, chroma-from-luma,
//                         DCT8/16x8/8x16 strategy search, scan-order quantisation, DC, nzeros;
//                         writes side-band grids + scan-ordered quantised coefficients
//                         (ref: enc_frame.cc:597-683 + enc_group.cc:304-443)
//   jxlt_token_kernel.h   token_kernel: one workgroup per 256x256 group, a lane per coefficient token: context
//                         modelling and raw 3-byte token records in stream order (ref: enc_group.cc:444-494)
//   jxlt_dc_kernels.h     the DC groups' token records (ref: enc_frame.cc:287-424, 536-570)
//   jxlt_pack_kernels.h   group_scan_kernel (exclusive scan of 32-bit counts); entropy-coded sections at their final
//                         bit positions (ref: enc_frame.cc:784-800)
//   jxlt_publish_kernel.h small results to the host's page-locked memory + a sequence word
#ifndef JXLT_DEVICE_H_
#define JXLT_DEVICE_H_

#include "jxlt_device_common.h"
#include "jxlt_tile_kernel.h"
#include "jxlt_token_kernel.h"
#include "jxlt_pack_kernels.h"
#include "jxlt_dc_kernels.h"
#include "jxlt_publish_kernel.h"

#endif  // JXLT_DEVICE_H_
