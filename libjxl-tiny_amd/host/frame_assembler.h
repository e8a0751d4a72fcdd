// Host bitstream back-end: turns the hot path's outputs (per-group raw AC token
// records + the DC-group side-band grids) into the JPEG XL frame bitstream.
//
// Replaces, on the host, the part of the reference's EncodeFrame that follows
// the per-group pixel pipeline:
//   /root/reference/encoder/enc_frame.cc:287-424  DC / AC-metadata tokenisers
//   /root/reference/encoder/enc_frame.cc:426-595  headers, globals, TOC
//   /root/reference/encoder/enc_frame.cc:765-816  OptimizeSections, CombineSections
#ifndef JXLT_HOST_FRAME_ASSEMBLER_H_
#define JXLT_HOST_FRAME_ASSEMBLER_H_

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "encoder/enc_bit_writer.h"
#include "entropy_coder.h"

namespace jxlt {

// enc_frame.cc:104-156
struct DistanceParams {
  float distance;
  int global_scale;
  int quant_dc;
  float scale;
  float inv_scale;
  float scale_dc;
  uint32_t x_qm_scale;
  uint32_t epf_iters;
};
DistanceParams ComputeDistanceParams(float distance);

// Image-absolute grids: blocks are 8x8 px (row pitch xsize_blocks), tiles 64x64
// px (row pitch xsize_tiles); one raw token buffer per 256x256 group in raster
// group order ([u8 pre-clustered ctx][u16 LE value] records).
struct FrameView {
  size_t xsize, ysize;
  const int16_t* quant_dc[3];
  const uint8_t* raw_quant_field;
  const uint8_t* ac_strategy;  // (type << 1) | is_first
  const int8_t* ytox_map;
  const int8_t* ytob_map;
  const uint8_t* const* group_tokens;
  const size_t* group_token_bytes;
};

// ---- production path: AC sections entropy-coded on the device -----------------
// Prefix codes from the device's [64][64] symbol histograms (the second half of
// OptimizeSections, enc_frame.cc:783): AC uses the static 1980 -> 64 pre-clustering,
// DC the identity map over its 45 contexts.
void BuildAcCode(const uint32_t* histograms, EntropyCode* ac_code);
void BuildDcCode(const uint32_t* histograms, EntropyCode* dc_code);
// table[ctx * 64 + sym] = (depth << 16) | bits for (pre-clustered) context ctx.
void FillCodeTable(const EntropyCode& code, uint32_t* table);
struct PackedSections {
  const uint8_t* bytes;
  const uint64_t* offset;  // [n + 1]
  const uint32_t* bits;    // [n]
  size_t n;
};
// Frame header + TOC + DCGlobal + DC groups + ACGlobal + AC groups, with both
// kinds of group sections already entropy-coded (byte-aligned) by the device.
// The frame is the concatenation head | dc.bytes | ac_global | ac.bytes; the two
// large blobs are not copied here.
// Not valid for single-group frames (their sections are bit-concatenated,
// enc_frame.cc:805-811): use AssembleFrame there.
struct FramePieces {
  std::vector<uint8_t> head;       // frame header, TOC, DCGlobal
  std::vector<uint8_t> ac_global;  // ACGlobal
};
bool FinishFrame(size_t xsize, size_t ysize, const DistanceParams& distp, const EntropyCode& dc_code,
                 const PackedSections& dc, const EntropyCode& ac_code, const PackedSections& ac,
                 FramePieces* out);
// The same in two steps, for callers that place the AC blob before its section sizes reach
// the host: the two global sections depend on the codes only; the head (frame header + TOC +
// DCGlobal) needs every section size.  HeadSizeBound() bounds head.size() from above.
struct FrameGlobals {
  std::vector<uint8_t> dc_global, ac_global;
};
void BuildFrameGlobals(size_t xsize, size_t ysize, const DistanceParams& distp, const EntropyCode& dc_code,
                       const EntropyCode& ac_code, FrameGlobals* out);
// The two halves (each depends on its own code only; byte aligned).
std::vector<uint8_t> BuildDcGlobal(size_t xsize, size_t ysize, const DistanceParams& distp, const EntropyCode& dc_code);
std::vector<uint8_t> BuildAcGlobal(size_t xsize, size_t ysize, const EntropyCode& ac_code);
size_t HeadSizeBound(size_t xsize, size_t ysize, const FrameGlobals& globals);
bool BuildFrameHead(size_t xsize, size_t ysize, const DistanceParams& distp, const FrameGlobals& globals,
                    const PackedSections& dc, const PackedSections& ac, std::vector<uint8_t>* head);

// Raw 3-byte records of DC group `index` as the host tokeniser produces them (tests).
std::vector<uint8_t> DcGroupRecords(const FrameView& frame, size_t index);

// Appends frame header + TOC + all sections to `writer` (must be byte aligned).
// num_threads <= 0 selects std::thread::hardware_concurrency().
bool AssembleFrame(const FrameView& frame, const DistanceParams& distp, jxl::BitWriter* writer,
                   int num_threads);

}  // namespace jxlt

#endif  // JXLT_HOST_FRAME_ASSEMBLER_H_
