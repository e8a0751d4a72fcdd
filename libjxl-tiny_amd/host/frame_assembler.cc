// See frame_assembler.h.
#include "frame_assembler.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

#include "../csrc/jxlt_tables.h"
#include "entropy_coder.h"

namespace jxlt {
namespace {

inline size_t DivCeil(size_t a, size_t b) { return (a + b - 1) / b; }
template <typename T>
inline T Clamp1(T v, T lo, T hi) { return v < lo ? lo : v > hi ? hi : v; }
inline uint32_t PackSigned(int32_t v) {  // common.h:54-58
  return (static_cast<uint32_t>(v) << 1) ^ ((static_cast<uint32_t>(~v) >> 31) - 1);
}
inline size_t CeilLog2Nonzero(uint64_t x) {  // base/bits.h:128-132
  const size_t fl = 63 ^ static_cast<size_t>(__builtin_clzll(x));
  return (x & (x - 1)) == 0 ? fl : fl + 1;
}

constexpr size_t kNumDCContexts = 45;     // enc_frame.cc:285
constexpr size_t kNumACPreClusters = 64;  // static_entropy_codes.h:161
constexpr size_t kNumACContexts = 1980;   // ac_context.h:76-77
constexpr size_t kNumTreeContexts = 6;    // enc_frame.cc:179

// DC quantiser for a distance (enc_frame.cc:95-102): 1.12 / d_eff with d_eff = 2.9 (d / 2.9)^0.57 -- the DC
// quantisation loosens more slowly than the distance above 2.9 -- kept inside [d / 2, d]; at most 50.
float QuantDC(float distance) {
  const float knee = 2.9f;
  const float softened = knee * std::pow(distance / knee, 0.57f);
  return std::min(1.12f / Clamp1(softened, 0.5f * distance, distance), 50.f);
}

template <typename F>
void ParallelFor(size_t n, int num_threads, const F& f) {
  if (num_threads <= 0) num_threads = static_cast<int>(std::thread::hardware_concurrency());
  num_threads = static_cast<int>(std::min<size_t>(std::max(1, num_threads), n));
  if (num_threads <= 1) {
    for (size_t i = 0; i < n; ++i) f(i, 0);
    return;
  }
  std::atomic<size_t> next(0);
  std::vector<std::thread> pool;
  for (int t = 0; t < num_threads; ++t) {
    pool.emplace_back([&, t]() {
      for (size_t i; (i = next.fetch_add(1)) < n;) f(i, t);
    });
  }
  for (auto& th : pool) th.join();
}

// Raw (un-entropy-coded) section: 3-byte records, context >= 128 = raw bits.
struct RawSection {
  const uint8_t* data = nullptr;
  size_t size = 0;
  std::vector<uint8_t> owned;
  void Put(uint8_t ctx, uint16_t value) {
    owned.push_back(ctx);
    owned.push_back(static_cast<uint8_t>(value & 0xFF));
    owned.push_back(static_cast<uint8_t>(value >> 8));
  }
  void Seal() {
    data = owned.data();
    size = owned.size();
  }
};

int32_t ClampedGradient(int32_t n, int32_t w, int32_t l) {  // enc_frame.cc:158-176
  const int32_t m = std::min(n, w), M = std::max(n, w);
  const int32_t grad = static_cast<int32_t>(static_cast<uint32_t>(n) + static_cast<uint32_t>(w) -
                                            static_cast<uint32_t>(l));
  const int32_t grad_clamp_M = (l < m) ? M : grad;
  return (l > M) ? m : grad_clamp_M;
}

// A rectangular window into an image-absolute grid.
template <typename T>
struct GridView {
  const T* base;
  size_t pitch, xs, ys;
  T at(size_t x, size_t y) const { return base[y * pitch + x]; }
};

// enc_frame.cc:287-316 (WriteDCTokens)
void DCTokens(const GridView<int16_t> dc[3], RawSection* out) {
  static const int kOrder[3] = {1, 0, 2};
  for (int c : kOrder) {
    const GridView<int16_t>& q = dc[c];
    for (size_t y = 0; y < q.ys; y++) {
      for (size_t x = 0; x < q.xs; x++) {
        int64_t left = (x ? q.at(x - 1, y) : y ? q.at(x, y - 1) : 0);
        int64_t top = (y ? q.at(x, y - 1) : left);
        int64_t topleft = (x && y ? q.at(x - 1, y - 1) : left);
        int32_t guess = ClampedGradient(static_cast<int32_t>(top), static_cast<int32_t>(left),
                                        static_cast<int32_t>(topleft));
        uint32_t gradprop =
            static_cast<uint32_t>(Clamp1<int64_t>(512 + top + left - topleft, 0, 1023));
        int32_t residual = q.at(x, y) - guess;
        out->Put(JXLT_kGradientContextLut[gradprop], static_cast<uint16_t>(PackSigned(residual)));
      }
    }
  }
}

inline bool IsFirst(uint8_t acs) { return acs & 1; }
inline int32_t StrategyCode(uint8_t acs) {  // ac_strategy.h:59-62
  static const uint8_t kLut[3] = {0, 6, 7};
  return kLut[acs >> 1];
}

// enc_frame.cc:329-424 (WriteACMetadataTokens)
void ACMetadataTokens(const GridView<int8_t>& ytox, const GridView<int8_t>& ytob,
                      const GridView<uint8_t>& acs, const GridView<uint8_t>& qf, RawSection* out) {
  for (size_t c = 0; c < 2; ++c) {
    const GridView<int8_t>& m = (c == 0 ? ytox : ytob);
    for (size_t y = 0; y < m.ys; y++) {
      for (size_t x = 0; x < m.xs; x++) {
        int64_t left = (x ? m.at(x - 1, y) : y ? m.at(x, y - 1) : 0);
        int64_t top = (y ? m.at(x, y - 1) : left);
        int64_t topleft = (x && y ? m.at(x - 1, y - 1) : left);
        int32_t guess = ClampedGradient(static_cast<int32_t>(top), static_cast<int32_t>(left),
                                        static_cast<int32_t>(topleft));
        int32_t residual = static_cast<int32_t>(m.at(x, y)) - guess;
        out->Put(static_cast<uint8_t>(2u - c), static_cast<uint16_t>(PackSigned(residual)));
      }
    }
  }
  {
    int32_t left = 0;
    for (size_t y = 0; y < acs.ys; y++)
      for (size_t x = 0; x < acs.xs; x++) {
        if (!IsFirst(acs.at(x, y))) continue;
        int32_t cur = StrategyCode(acs.at(x, y));
        uint8_t ctx = (left > 11 ? 7 : left > 5 ? 8 : left > 3 ? 9 : 10);
        out->Put(ctx, static_cast<uint16_t>(PackSigned(cur)));
        left = cur;
      }
  }
  {
    int32_t left = StrategyCode(acs.at(0, 0));  // sic (enc_frame.cc:386)
    for (size_t y = 0; y < acs.ys; y++)
      for (size_t x = 0; x < acs.xs; x++) {
        if (!IsFirst(acs.at(x, y))) continue;
        size_t cur = qf.at(x, y) - 1;
        int32_t residual = static_cast<int32_t>(cur - left);
        uint8_t ctx = (left > 11 ? 3 : left > 5 ? 4 : left > 3 ? 5 : 6);
        out->Put(ctx, static_cast<uint16_t>(PackSigned(residual)));
        left = static_cast<int32_t>(cur);
      }
  }
  for (size_t i = 0; i < acs.xs * acs.ys; ++i) out->Put(0, static_cast<uint16_t>(PackSigned(4)));
}

// enc_frame.cc:536-570 (WriteDCGroup), raw-record form
void DCGroupSection(const FrameView& f, size_t xsize_blocks, size_t xsize_tiles, size_t dc_gx,
                    size_t dc_gy, RawSection* out) {
  const size_t ysize_blocks = DivCeil(f.ysize, 8);
  const size_t bx0 = dc_gx * 256, by0 = dc_gy * 256;
  const size_t nbx = std::min<size_t>(256, xsize_blocks - bx0);
  const size_t nby = std::min<size_t>(256, ysize_blocks - by0);
  GridView<int16_t> dc[3];
  for (int c = 0; c < 3; c++) dc[c] = {f.quant_dc[c] + by0 * xsize_blocks + bx0, xsize_blocks, nbx, nby};
  GridView<uint8_t> acs = {f.ac_strategy + by0 * xsize_blocks + bx0, xsize_blocks, nbx, nby};
  GridView<uint8_t> qf = {f.raw_quant_field + by0 * xsize_blocks + bx0, xsize_blocks, nbx, nby};
  const size_t tx0 = dc_gx * 32, ty0 = dc_gy * 32;
  const size_t ntx = DivCeil(nbx * 8, 64), nty = DivCeil(nby * 8, 64);
  GridView<int8_t> ytox = {f.ytox_map + ty0 * xsize_tiles + tx0, xsize_tiles, ntx, nty};
  GridView<int8_t> ytob = {f.ytob_map + ty0 * xsize_tiles + tx0, xsize_tiles, ntx, nty};

  out->owned.reserve(3 * (nbx * nby * 6 + 64));
  out->Put(kMaxContexts + 6, 12);  // extra_dc_precision(2)=0, global tree etc (4)=3
  DCTokens(dc, out);
  size_t num_ac_blocks = 0;
  for (size_t y = 0; y < nby; y++)
    for (size_t x = 0; x < nbx; x++) num_ac_blocks += IsFirst(acs.at(x, y));
  const size_t nb_bits = CeilLog2Nonzero(nbx * nby);
  if (nb_bits != 0) out->Put(static_cast<uint8_t>(kMaxContexts + nb_bits), static_cast<uint16_t>(num_ac_blocks - 1));
  out->Put(kMaxContexts + 4, 3);
  ACMetadataTokens(ytox, ytob, acs, qf, out);
  out->Seal();
}

// enc_frame.cc:765-802 (OptimizeSections)
void OptimizeSections(const std::vector<RawSection>& raw, size_t num_histograms,
                      const uint8_t* static_map, size_t num_static_contexts, EntropyCode* code,
                      std::vector<jxl::BitWriter>* encoded, int num_threads) {
  int nt = num_threads <= 0 ? static_cast<int>(std::thread::hardware_concurrency()) : num_threads;
  nt = std::max(1, nt);
  std::vector<std::vector<Histogram>> partial(nt, std::vector<Histogram>(num_histograms));
  ParallelFor(raw.size(), nt, [&](size_t i, int t) {
    std::vector<Histogram>& h = partial[t];
    const uint8_t* p = raw[i].data;
    for (size_t j = 0; j < raw[i].size; j += 3) {
      const uint8_t context = p[j];
      if (context >= kMaxContexts) continue;
      const uint32_t value = (static_cast<uint32_t>(p[j + 2]) << 8) + p[j + 1];
      uint32_t tok, nbits, bits;
      HybridUintEncode(value, &tok, &nbits, &bits);
      h[context].Add(tok);
    }
  });
  std::vector<Histogram> histograms(num_histograms);
  for (const auto& ph : partial)
    for (size_t k = 0; k < num_histograms; ++k) histograms[k].AddHistogram(ph[k]);
  OptimizeEntropyCode(&histograms, static_map, num_static_contexts, code);

  encoded->clear();
  encoded->resize(raw.size());
  ParallelFor(raw.size(), nt, [&](size_t i, int) {
    jxl::BitWriter& w = (*encoded)[i];
    const uint8_t* p = raw[i].data;
    w.Reserve(raw[i].size / 3 + 16);
    for (size_t j = 0; j < raw[i].size; j += 3) {
      const uint8_t context = p[j];
      const uint32_t value = (static_cast<uint32_t>(p[j + 2]) << 8) + p[j + 1];
      if (context >= kMaxContexts) w.Write(context - kMaxContexts, value);
      else WriteToken(context, value, *code, &w);
    }
  });
}

// A run of fixed-width header fields.
struct Field {
  uint8_t bits;
  uint32_t value;
};
template <size_t N>
void WriteFields(const Field (&fields)[N], jxl::BitWriter* writer) {
  for (const Field& f : fields) writer->Write(f.bits, f.value);
}

// JPEG XL "U32" with four direct-offset branches: selector s (2 bits) when value < base[s] + 2^bits[s],
// then value - base[s] in bits[s] bits.
void WriteU32(uint32_t value, const uint32_t (&base)[4], const uint8_t (&bits)[4], jxl::BitWriter* writer) {
  for (uint32_t s = 0; s < 4; ++s) {
    if (s == 3 || value < base[s] + (1u << bits[s])) {
      writer->Write(2, s);
      writer->Write(bits[s], value - base[s]);
      return;
    }
  }
}

// Frame header of a single regular VarDCT frame (the field values of enc_frame.cc:426-457): all-default flag
// off; frame type 0; VarDCT; flags = 128 (skip adaptive DC smoothing), coded as selector 2 + (128 - 17);
// no upsampling; the two colour-channel quant-matrix scales; one pass; no crop; replace blending; last
// frame; no name; then the restoration filter: default (gaborish off is NOT default, so only epf_iters == 2
// may use the shortcut), else gaborish off + the edge-preserving filter's iteration count with default
// parameters; no extensions anywhere.
void WriteFrameHeader(uint32_t x_qm_scale, uint32_t epf_iters, jxl::BitWriter* writer) {
  const Field front[] = {{1, 0}, {2, 0}, {1, 0}, {2, 2}, {8, 128 - 17}, {2, 0}, {3, x_qm_scale}, {3, 2},
                         {2, 0}, {1, 0}, {2, 0}, {1, 1}, {2, 0}};
  WriteFields(front, writer);
  if (epf_iters == 2) {
    writer->Write(1, 1);
  } else {
    const Field filter[] = {{1, 0}, {1, 0}, {2, epf_iters}};
    WriteFields(filter, writer);
    if (epf_iters != 0) writer->Write(3, 0);  // sharpness, weights, sigma: three "default" flags
    writer->Write(2, 0);
  }
  writer->Write(2, 0);
}

// Quantiser scales of DCGlobal (values of enc_frame.cc:459-486): global_scale as U32(1 + u(11), 2049 + u(11),
// 4097 + u(12), 8193 + u(16)); quant_dc as U32(16, 1 + u(5), 1 + u(8), 1 + u(16)).
void WriteQuantScales(int global_scale, int quant_dc, jxl::BitWriter* writer) {
  static const uint32_t kScaleBase[4] = {1, 2049, 4097, 8193};
  static const uint8_t kScaleBits[4] = {11, 11, 12, 16};
  WriteU32(static_cast<uint32_t>(global_scale), kScaleBase, kScaleBits, writer);
  if (quant_dc == 16) {
    writer->Write(2, 0);  // the constant branch
  } else {
    const uint32_t sel = quant_dc < 33 ? 1 : quant_dc < 257 ? 2 : 3;
    static const uint8_t kDcBits[4] = {0, 5, 8, 16};
    writer->Write(2, sel);
    writer->Write(kDcBits[sel], static_cast<uint32_t>(quant_dc - 1));
  }
}

void WriteContextTree(size_t num_dc_groups, jxl::BitWriter* writer) {  // enc_frame.cc:488-503
  std::vector<Token> tokens(313);
  for (size_t i = 0; i < 313; ++i)
    tokens[i] = {JXLT_kContextTreeTokens[2 * i], JXLT_kContextTreeTokens[2 * i + 1]};
  tokens[1].value = PackSigned(static_cast<int32_t>(1 + num_dc_groups));
  EntropyCode code;
  OptimizeEntropyCode(tokens, kNumTreeContexts, &code);
  writer->Write(1, 1);  // not an empty tree
  writer->Write(1, 0);  // no lz77
  WriteEntropyCode(code, writer);
  for (const Token& t : tokens) WriteToken(t.context, t.value, code, writer);
}

// DCGlobal (enc_frame.cc:505-522): default DC dequantisation, the quantiser scales, an explicit block context
// map without DC / quant-field thresholds (the compact 39-entry map), default chroma-from-luma base, the global
// modular tree, and the DC code itself (no LZ77 either side).
void WriteDCGlobal(const DistanceParams& distp, size_t num_dc_groups, const EntropyCode& dc_code,
                   jxl::BitWriter* writer) {
  writer->Write(1, 1);
  WriteQuantScales(distp.global_scale, distp.quant_dc, writer);
  const Field block_ctx[] = {{1, 0}, {16, 0}};
  WriteFields(block_ctx, writer);
  WriteStaticContextMap(JXLT_kCompactBlockContextMap, 39, writer);
  writer->Write(1, 1);
  WriteContextTree(num_dc_groups, writer);
  writer->Write(1, 0);
  WriteEntropyCode(dc_code, writer);
}

// ACGlobal (enc_frame.cc:524-534): default quantisation matrices, one histogram set (its index field is
// ceil(log2(num_groups)) bits wide), default coefficient orders (selector 3, 13 zero bits), no LZ77, the AC code.
void WriteACGlobal(size_t num_groups, const EntropyCode& ac_code, jxl::BitWriter* writer) {
  writer->Write(1, 1);
  if (const size_t index_bits = CeilLog2Nonzero(num_groups)) writer->Write(index_bits, 0);
  const Field orders[] = {{2, 3}, {13, 0}, {1, 0}};
  WriteFields(orders, writer);
  WriteEntropyCode(ac_code, writer);
}

// Table of contents (enc_frame.cc:572-595): no permutation, then every section's byte size as
// U32(u(10), 1024 + u(14), 17408 + u(22), 4211712 + u(30)), byte aligned before and after.  Sections of
// 4 MiB and more are refused like the reference does (its assert at :577).
bool WriteTOCSizes(const std::vector<size_t>& section_sizes, jxl::BitWriter* output) {
  static const uint8_t kSizeBits[4] = {10, 14, 22, 30};
  static const uint32_t kSizeBase[4] = {0, 1u << 10, (1u << 10) + (1u << 14), (1u << 10) + (1u << 14) + (1u << 22)};
  output->Write(1, 0);
  output->ZeroPadToByte();
  for (const size_t bytes : section_sizes) {
    if (bytes >= (1u << 22)) return false;
    WriteU32(static_cast<uint32_t>(bytes), kSizeBase, kSizeBits, output);
  }
  output->ZeroPadToByte();
  return true;
}

bool WriteTOC(const std::vector<jxl::BitWriter>& sections, jxl::BitWriter* output) {
  std::vector<size_t> sizes;
  sizes.reserve(sections.size());
  for (const jxl::BitWriter& s : sections) sizes.push_back(DivCeil(s.BitsWritten(), 8));
  return WriteTOCSizes(sizes, output);
}

}  // namespace

// The scalars the rest of the encoder derives from the butteraugli distance (enc_frame.cc:104-156), in float
// arithmetic exactly as there:
//   AC:  the quant field aims at 5, the AC scale at 0.8 / distance -> global_scale = 65536 * 0.8 / (5 d), kept in
//        [1, 32768] and below 1.6 * 4096 * (DC quantiser), as an integer; scale = global_scale / 65536
//   DC:  quant_dc = round(DC quantiser / scale) in [1, 65536]
//   X channel matrix scale 2, +1 above d = 1.25, +1 above d = 9, +1 below d = 0.299
//   edge-preserving filter iterations: one per threshold 0.7 / 1.5 / 4.0 reached
DistanceParams ComputeDistanceParams(float distance) {
  DistanceParams out;
  out.distance = distance;
  const float dc_quantiser = QuantDC(distance);
  const float ideal = Clamp1(65536 * 0.8f / (distance * 5.0f), 1.0f, 32768.0f);
  const int ceiling = static_cast<int>(dc_quantiser * 4096 * 1.6);
  out.global_scale = Clamp1(static_cast<int>(ideal), 1, ceiling);
  out.scale = out.global_scale * (1.0f / 65536);
  out.inv_scale = 1.0f / out.scale;
  out.quant_dc = Clamp1(static_cast<int>(dc_quantiser / out.scale + 0.5f), 1, 65536);
  out.scale_dc = out.quant_dc * out.scale;
  out.x_qm_scale = 2u + (distance > 1.25f) + (distance > 9.0f) + (distance < 0.299f);
  out.epf_iters = 0u + (distance >= 0.7f) + (distance >= 1.5f) + (distance >= 4.0f);
  return out;
}

std::vector<uint8_t> DcGroupRecords(const FrameView& f, size_t index) {
  const size_t xsize_dc_groups = DivCeil(f.xsize, 2048);
  RawSection sec;
  DCGroupSection(f, DivCeil(f.xsize, 8), DivCeil(f.xsize, 64), index % xsize_dc_groups, index / xsize_dc_groups, &sec);
  return sec.owned;
}

void BuildAcCode(const uint32_t* histograms, EntropyCode* ac_code) {
  std::vector<Histogram> h(kNumACPreClusters);
  for (size_t c = 0; c < kNumACPreClusters; ++c)
    for (size_t s = 0; s < kAlphabetSize; ++s) {
      h[c].counts[s] = histograms[c * kAlphabetSize + s];
      h[c].total_count += histograms[c * kAlphabetSize + s];
    }
  OptimizeEntropyCode(&h, JXLT_kACContextMap, kNumACContexts, ac_code);
}

void FillCodeTable(const EntropyCode& code, uint32_t* table) {
  for (size_t c = 0; c < 64; ++c) {
    for (size_t s = 0; s < kAlphabetSize; ++s) {
      uint32_t e = 0;
      if (c < code.context_map.size()) {
        const PrefixCode& pc = code.prefix_codes[code.context_map[c]];
        e = pc.single_symbol ? 0u : (static_cast<uint32_t>(pc.depths[s]) << 16) | pc.bits[s];
      }
      table[c * kAlphabetSize + s] = e;
    }
  }
}

void BuildDcCode(const uint32_t* histograms, EntropyCode* dc_code) {
  std::vector<Histogram> h(kNumDCContexts);
  uint8_t dc_identity[kNumDCContexts];
  for (size_t c = 0; c < kNumDCContexts; ++c) {
    dc_identity[c] = static_cast<uint8_t>(c);
    for (size_t s = 0; s < kAlphabetSize; ++s) {
      h[c].counts[s] = histograms[c * kAlphabetSize + s];
      h[c].total_count += histograms[c * kAlphabetSize + s];
    }
  }
  OptimizeEntropyCode(&h, dc_identity, kNumDCContexts, dc_code);
}

std::vector<uint8_t> BuildDcGlobal(size_t xsize, size_t ysize, const DistanceParams& distp, const EntropyCode& dc_code) {
  const size_t num_dc_groups = DivCeil(xsize, 2048) * DivCeil(ysize, 2048);
  jxl::BitWriter w;
  WriteDCGlobal(distp, num_dc_groups, dc_code, &w);
  w.ZeroPadToByte();
  return w.TakeBytes();
}

std::vector<uint8_t> BuildAcGlobal(size_t xsize, size_t ysize, const EntropyCode& ac_code) {
  const size_t num_groups = DivCeil(xsize, 256) * DivCeil(ysize, 256);
  jxl::BitWriter w;
  WriteACGlobal(num_groups, ac_code, &w);
  w.ZeroPadToByte();
  return w.TakeBytes();
}

void BuildFrameGlobals(size_t xsize, size_t ysize, const DistanceParams& distp, const EntropyCode& dc_code,
                       const EntropyCode& ac_code, FrameGlobals* out) {
  out->dc_global = BuildDcGlobal(xsize, ysize, distp, dc_code);
  out->ac_global = BuildAcGlobal(xsize, ysize, ac_code);
}

size_t HeadSizeBound(size_t xsize, size_t ysize, const FrameGlobals& globals) {
  const size_t num_groups = DivCeil(xsize, 256) * DivCeil(ysize, 256);
  const size_t num_dc_groups = DivCeil(xsize, 2048) * DivCeil(ysize, 2048);
  // frame header (a few bytes) + permutation flag + <= 32 bits per TOC entry + DCGlobal
  return 64 + 4 * (2 + num_dc_groups + num_groups) + globals.dc_global.size();
}

bool BuildFrameHead(size_t xsize, size_t ysize, const DistanceParams& distp, const FrameGlobals& globals,
                    const PackedSections& dc, const PackedSections& ac, std::vector<uint8_t>* head_out) {
  const size_t num_groups = DivCeil(xsize, 256) * DivCeil(ysize, 256);
  const size_t num_dc_groups = DivCeil(xsize, 2048) * DivCeil(ysize, 2048);
  if (ac.n != num_groups || dc.n != num_dc_groups || 2 + num_dc_groups + num_groups == 4) return false;
  std::vector<size_t> sizes;
  sizes.reserve(2 + num_dc_groups + num_groups);
  sizes.push_back(globals.dc_global.size());
  for (size_t g = 0; g < num_dc_groups; ++g) sizes.push_back(static_cast<size_t>(dc.offset[g + 1] - dc.offset[g]));
  sizes.push_back(globals.ac_global.size());
  for (size_t g = 0; g < num_groups; ++g) sizes.push_back(static_cast<size_t>(ac.offset[g + 1] - ac.offset[g]));
  jxl::BitWriter head;
  WriteFrameHeader(distp.x_qm_scale, distp.epf_iters, &head);
  if (!WriteTOCSizes(sizes, &head)) return false;
  head.AppendBytes(globals.dc_global.data(), globals.dc_global.size());
  *head_out = head.TakeBytes();
  return true;
}

bool FinishFrame(size_t xsize, size_t ysize, const DistanceParams& distp, const EntropyCode& dc_code,
                 const PackedSections& dc, const EntropyCode& ac_code, const PackedSections& ac,
                 FramePieces* out) {
  FrameGlobals globals;
  BuildFrameGlobals(xsize, ysize, distp, dc_code, ac_code, &globals);
  if (!BuildFrameHead(xsize, ysize, distp, globals, dc, ac, &out->head)) return false;
  out->ac_global = std::move(globals.ac_global);
  return true;
}

bool AssembleFrame(const FrameView& f, const DistanceParams& distp, jxl::BitWriter* writer,
                   int num_threads) {
  const size_t xsize_blocks = DivCeil(f.xsize, 8);
  const size_t xsize_tiles = DivCeil(f.xsize, 64);
  const size_t xsize_groups = DivCeil(f.xsize, 256), ysize_groups = DivCeil(f.ysize, 256);
  const size_t xsize_dc_groups = DivCeil(f.xsize, 2048), ysize_dc_groups = DivCeil(f.ysize, 2048);
  const size_t num_groups = xsize_groups * ysize_groups;
  const size_t num_dc_groups = xsize_dc_groups * ysize_dc_groups;

  // Raw DC-group sections (enc_frame.cc:760-761).
  std::vector<RawSection> dc_raw(num_dc_groups);
  ParallelFor(num_dc_groups, num_threads, [&](size_t i, int) {
    DCGroupSection(f, xsize_blocks, xsize_tiles, i % xsize_dc_groups, i / xsize_dc_groups, &dc_raw[i]);
  });
  std::vector<RawSection> ac_raw(num_groups);
  for (size_t i = 0; i < num_groups; ++i) {
    if (f.group_token_bytes[i] % 3 != 0) return false;
    ac_raw[i].data = f.group_tokens[i];
    ac_raw[i].size = f.group_token_bytes[i];
  }

  // Entropy-code optimisation (enc_frame.cc:846-850).
  uint8_t dc_identity[kNumDCContexts];
  for (size_t i = 0; i < kNumDCContexts; ++i) dc_identity[i] = static_cast<uint8_t>(i);
  EntropyCode dc_code, ac_code;
  std::vector<jxl::BitWriter> dc_sections, ac_sections;
  OptimizeSections(dc_raw, kNumDCContexts, dc_identity, kNumDCContexts, &dc_code, &dc_sections,
                   num_threads);
  OptimizeSections(ac_raw, kNumACPreClusters, JXLT_kACContextMap, kNumACContexts, &ac_code,
                   &ac_sections, num_threads);

  // Section order: DCGlobal, DC groups, ACGlobal, AC groups (enc_frame.cc:721-722,853-854).
  std::vector<jxl::BitWriter> sections;
  sections.reserve(2 + num_dc_groups + num_groups);
  sections.emplace_back();
  WriteDCGlobal(distp, num_dc_groups, dc_code, &sections.back());
  for (auto& s : dc_sections) sections.push_back(std::move(s));
  sections.emplace_back();
  WriteACGlobal(num_groups, ac_code, &sections.back());
  for (auto& s : ac_sections) sections.push_back(std::move(s));

  WriteFrameHeader(distp.x_qm_scale, distp.epf_iters, writer);
  if (sections.size() == 4) {  // enc_frame.cc:805-811: single group => one section
    for (size_t i = 1; i < 4; ++i) sections[0].Append(sections[i]);
    sections.resize(1);
  }
  if (!WriteTOC(sections, writer)) return false;
  writer->AppendByteAligned(&sections);
  return true;
}

}  // namespace jxlt
