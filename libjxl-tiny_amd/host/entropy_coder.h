// Host-side entropy coding for the JPEG XL tiny bitstream: hybrid-uint token
// split, per-context histograms, greedy histogram clustering (<= 8 clusters),
// length-limited Huffman codes and their brotli-style serialisation.
//
// Behavioural reference (bit-exact output is required):
//   /root/reference/encoder/token.h:32-48           (UintCoder::Encode)
//   /root/reference/encoder/enc_cluster.cc:18-131   (clustering)
//   /root/reference/encoder/enc_huffman_tree.cc:65-142
//   /root/reference/encoder/enc_entropy_code.cc:18-553
#ifndef JXLT_HOST_ENTROPY_CODER_H_
#define JXLT_HOST_ENTROPY_CODER_H_

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "encoder/enc_bit_writer.h"

namespace jxlt {

constexpr size_t kAlphabetSize = 64;
constexpr uint8_t kMaxContexts = 128;  // record contexts >= this are raw-bit escapes
constexpr size_t kMaxBitsPerToken = 24;

struct Token {
  uint32_t context;
  uint32_t value;
};

struct PrefixCode {
  uint8_t depths[kAlphabetSize];
  uint16_t bits[kAlphabetSize];
  // The histogram has exactly one used symbol.  Such a code is serialised as a simple code with
  // one symbol, which a decoder reads with ZERO bits per token (18181-1 / RFC 7932 3.4).  The
  // reference leaves the construction's placeholder depth 1 in place (enc_huffman_tree.cc:84-87,
  // "will be fixed on upper level" -- enc_entropy_code.cc:411-416 does not) and so writes one
  // bit per token of such a context: a stream no decoder can read.  Set unless
  // SetReferenceSingleSymbolEmulation(true); tokens of a flagged code are written with depth 0.
  bool single_symbol;
  uint32_t TokenDepth(uint32_t tok) const { return single_symbol ? 0u : depths[tok]; }
};

// true: reproduce the reference's one-bit-per-token output for single-symbol codes byte for byte
// (undecodable streams in exactly those cases).  Default false: conformant output.
void SetReferenceSingleSymbolEmulation(bool on);
bool ReferenceSingleSymbolEmulation();

struct Histogram {
  uint32_t counts[kAlphabetSize] = {};
  size_t total_count = 0;
  size_t bit_cost = 0;  // filled by ComputeBitCost
  void Add(uint32_t symbol) {
    ++counts[symbol];
    ++total_count;
  }
  void AddHistogram(const Histogram& o) {
    for (size_t i = 0; i < kAlphabetSize; ++i) counts[i] += o.counts[i];
    total_count += o.total_count;
  }
};

// A context map + one prefix code per cluster.  `orig_context_map` is the
// static pre-clustering that produced the histogram indices (when present the
// serialised map is the composition).
struct EntropyCode {
  std::vector<uint8_t> context_map;
  std::vector<PrefixCode> prefix_codes;
  std::vector<uint8_t> orig_context_map;  // empty if none
};

// token.h:32-48
inline void HybridUintEncode(uint32_t value, uint32_t* token, uint32_t* nbits, uint32_t* bits) {
  if (value < 16) {
    *token = value;
    *nbits = 0;
    *bits = 0;
  } else {
    const uint32_t n = 31u ^ static_cast<uint32_t>(__builtin_clz(value));
    const uint32_t m = value - (1u << n);
    *token = (n << 2) + (m >> (n - 2));
    *nbits = n - 2;
    *bits = value & ((1u << *nbits) - 1);
  }
}

// enc_entropy_code.h:34-42
inline void WriteToken(uint32_t cluster_ctx, uint32_t value, const EntropyCode& code,
                       jxl::BitWriter* writer) {
  uint32_t tok, nbits, bits;
  HybridUintEncode(value, &tok, &nbits, &bits);
  const PrefixCode& pc = code.prefix_codes[code.context_map[cluster_ctx]];
  const uint32_t depth = pc.TokenDepth(tok);
  uint64_t data = depth ? pc.bits[tok] : 0;
  data |= static_cast<uint64_t>(bits) << depth;
  writer->Write(depth + nbits, data);
}

void CreateHuffmanTree(const uint32_t* counts, size_t length, int tree_limit, uint8_t* depth);

// Code construction spreads its cost evaluations over a few helper threads that sleep between
// frames.  A caller that knows when the histograms will arrive (e.g. from the duration of the
// previous frame's device pipeline) can announce it: the helpers wake up `start_in_ms` from now
// and spin for the session until `give_up_in_ms` from now.  Purely a latency hint.
void WarmCodeConstruction(double start_in_ms, double give_up_in_ms);
// Whether a clustering of the calling thread has used the helper threads since the last call of this function
// (a caller that announces its code constructions need not announce the ones that work alone).
bool TakeClusteringShared();
void ConvertBitDepthsToSymbols(const uint8_t* depth, size_t len, uint16_t* bits);

// Clusters `histograms` (in place, result = cluster histograms) and returns the
// histogram-index -> cluster map.
void ClusterHistograms(std::vector<Histogram>* histograms, std::vector<uint8_t>* context_map);

// From raw tokens with direct context ids (no pre-clustering).
void OptimizeEntropyCode(const std::vector<Token>& tokens, size_t num_contexts, EntropyCode* code);
// From per-pre-cluster histograms; `static_map` (num_contexts -> histogram index)
// becomes orig_context_map.
void OptimizeEntropyCode(std::vector<Histogram>* histograms, const uint8_t* static_map,
                         size_t num_static_contexts, EntropyCode* code);

void WriteContextMap(const EntropyCode& code, jxl::BitWriter* writer);
// Context map given explicitly (used for the fixed block-context map).
void WriteStaticContextMap(const uint8_t* map, size_t n, jxl::BitWriter* writer);
void WriteEntropyCode(const EntropyCode& code, jxl::BitWriter* writer);

}  // namespace jxlt

#endif  // JXLT_HOST_ENTROPY_CODER_H_
