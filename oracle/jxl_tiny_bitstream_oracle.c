/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * CPU restatement ("oracle") of the libjxl-tiny BITSTREAM stage: everything the reference's
 * EncodeFile / EncodeFrame do with the outputs of the per-group pixel pipeline
 * (oracle/jxl_tiny_oracle.c) to arrive at the bytes of the .jxl file.  Plain C, written from
 * the reference's behaviour; shares no source with the product's host back-end
 * (libjxl-tiny_amd/host/): the tests compare the two.
 *
 *   file header                          enc_file.cc:30-95
 *   frame header                         enc_frame.cc:426-457
 *   DC-group sections (raw records)      enc_frame.cc:287-316 (WriteDCTokens), :329-424
 *                                        (WriteACMetadataTokens), :536-570 (WriteDCGroup)
 *   code optimisation + section rewrite  enc_frame.cc:766-802 (OptimizeSections)
 *   histogram clustering                 enc_cluster.cc:18-131
 *   Huffman code lengths                 enc_huffman_tree.cc:65-142
 *   code / context map serialisation     enc_entropy_code.cc:18-553
 *   DCGlobal, ACGlobal, TOC, assembly    enc_frame.cc:459-534, 572-595, 804-816
 *   bit order                            enc_bit_writer.cc:119-142 (LSB first)
 *
 * PARITY UNPINNED, like the pixel-pipeline oracle: the reference ships no vectors and
 * cannot be built here; the pins are the known-answer codestream SIZES of
 * tests/test_oracle_known_answers.py (which go through this file).
 *
 * One deliberate switch: `reference_single_symbol`.  The reference writes one bit per token
 * of a prefix code with a single used symbol (enc_huffman_tree.cc:84-87 leaves depth 1,
 * enc_entropy_code.h:34-42 writes it) although that code is serialised as a one-symbol code
 * which decoders read with ZERO bits.  1 = the reference's bytes; 0 = zero bits (decodable).
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "jxl_tiny_oracle.h"
#include "orc_tables.h"

#define BS_ALPHABET 64
#define BS_MAX_CONTEXTS 128 /* entropy_code.h:17: record contexts >= this are raw-bit escapes */
#define BS_CLUSTER_LIMIT 8  /* enc_cluster.cc:121 */

/* ------------------------------------------------------------------ bit sink */
/* enc_bit_writer.cc:119-142: bits are appended LSB first. */
typedef struct {
  uint8_t* data;
  size_t cap;  /* bytes */
  size_t bits; /* bits written */
} bitsink;

static void sink_reserve(bitsink* s, size_t more_bits) {
  const size_t need = (s->bits + more_bits) / 8 + 16;
  if (need <= s->cap) return;
  size_t cap = s->cap ? s->cap : 256;
  while (cap < need) cap *= 2;
  s->data = (uint8_t*)realloc(s->data, cap);
  memset(s->data + s->cap, 0, cap - s->cap);
  s->cap = cap;
}

static void sink_put(bitsink* s, unsigned nbits, uint64_t value) {
  if (nbits == 0) return;
  sink_reserve(s, nbits);
  size_t byte = s->bits >> 3;
  unsigned shift = (unsigned)(s->bits & 7);
  s->bits += nbits;
  /* at most 56 bits per call */
  value <<= shift;
  unsigned total = nbits + shift;
  while (total > 0) {
    s->data[byte++] |= (uint8_t)(value & 0xFF);
    value >>= 8;
    total = total > 8 ? total - 8 : 0;
  }
}

static void sink_pad_to_byte(bitsink* s) { s->bits = (s->bits + 7) & ~(size_t)7; sink_reserve(s, 0); }
static size_t sink_bytes(const bitsink* s) { return (s->bits + 7) >> 3; }

/* enc_bit_writer.cc:98-117 (Append): bitwise concatenation. */
static void sink_append_bits(bitsink* dst, const bitsink* src) {
  const size_t whole = src->bits >> 3, tail = src->bits & 7;
  for (size_t i = 0; i < whole; i++) sink_put(dst, 8, src->data[i]);
  if (tail) sink_put(dst, (unsigned)tail, src->data[whole] & ((1u << tail) - 1u));
}

static void sink_free(bitsink* s) {
  free(s->data);
  memset(s, 0, sizeof(*s));
}

/* ------------------------------------------------------------------ small helpers */
static uint32_t pack_signed(int32_t v) { /* common.h:54-58 */
  return ((uint32_t)v << 1) ^ (((uint32_t)(~v) >> 31) - 1u);
}
static unsigned floor_log2(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }
static unsigned ceil_log2(size_t v) { /* CeilLog2Nonzero */
  unsigned f = 63u - (unsigned)__builtin_clzll((unsigned long long)v);
  return (v & (v - 1)) ? f + 1 : f;
}
static size_t div_ceil(size_t a, size_t b) { return (a + b - 1) / b; }

/* token.h:32-48 */
static void hybrid_uint(uint32_t value, uint32_t* sym, uint32_t* nbits, uint32_t* extra) {
  if (value < 16) {
    *sym = value;
    *nbits = 0;
    *extra = 0;
    return;
  }
  const unsigned n = floor_log2(value);
  const uint32_t m = value - (1u << n);
  *sym = (n << 2) + (m >> (n - 2));
  *nbits = n - 2;
  *extra = value & ((1u << (n - 2)) - 1u);
}

/* ------------------------------------------------------------------ Huffman code lengths */
/* enc_huffman_tree.cc:65-142.  Leaves are collected from the highest symbol down, stably
 * sorted by count; the classic two-queue merge prefers a leaf on ties; if the deepest leaf
 * exceeds `limit` the counts are floored at 1, 3, 7, ... and the construction repeated.
 * A lone symbol gets the placeholder depth 1 (:84-87). */
typedef struct {
  uint32_t count;
  int left, right_or_symbol; /* left < 0: leaf */
} hnode;

static void huffman_depths(const uint32_t* counts, size_t length, int limit, uint8_t* depth) {
  hnode node[2 * BS_ALPHABET + 2];
  for (uint32_t floor_plus_1 = 1;; floor_plus_1 *= 2) {
    size_t n = 0;
    for (size_t i = length; i-- > 0;) {
      if (!counts[i]) continue;
      const uint32_t lo = floor_plus_1 - 1;
      node[n].count = counts[i] > lo ? counts[i] : lo;
      node[n].left = -1;
      node[n].right_or_symbol = (int)i;
      n++;
    }
    if (n == 0) return; /* (the reference is never called with an empty histogram) */
    if (n == 1) {
      depth[node[0].right_or_symbol] = 1;
      return;
    }
    /* stable insertion sort by count */
    for (size_t i = 1; i < n; i++) {
      const hnode key = node[i];
      size_t j = i;
      while (j > 0 && node[j - 1].count > key.count) {
        node[j] = node[j - 1];
        j--;
      }
      node[j] = key;
    }
    const hnode sentinel = {UINT32_MAX, -1, -1};
    node[n] = sentinel;
    size_t leaf = 0, inner = n + 1, end = n + 1;
    node[end] = sentinel;
    for (size_t k = n - 1; k != 0; k--) {
      size_t pick[2];
      for (int side = 0; side < 2; side++) {
        if (node[leaf].count <= node[inner].count) {
          pick[side] = leaf++;
        } else {
          pick[side] = inner++;
        }
      }
      node[end].count = node[pick[0]].count + node[pick[1]].count;
      node[end].left = (int)pick[0];
      node[end].right_or_symbol = (int)pick[1];
      end++;
      node[end] = sentinel;
    }
    /* depth of every leaf below the root node[2n - 1] (iterative walk) */
    int stack_node[2 * BS_ALPHABET + 2];
    uint8_t stack_level[2 * BS_ALPHABET + 2];
    int sp = 0;
    stack_node[sp] = (int)(2 * n - 1);
    stack_level[sp++] = 0;
    uint8_t deepest = 0;
    while (sp > 0) {
      const hnode* p = &node[stack_node[--sp]];
      const uint8_t level = stack_level[sp];
      if (p->left >= 0) {
        stack_node[sp] = p->left;
        stack_level[sp++] = (uint8_t)(level + 1);
        stack_node[sp] = p->right_or_symbol;
        stack_level[sp++] = (uint8_t)(level + 1);
      } else {
        depth[p->right_or_symbol] = level;
        if (level > deepest) deepest = level;
      }
    }
    /* the reference takes the maximum over depth[0..length), which may still hold larger
     * values of symbols that are unused now; callers always pass zeroed arrays */
    if ((int)deepest <= limit) return;
  }
}

/* enc_entropy_code.cc:262-315: canonical code, bits stored reversed (LSB-first writer). */
static void canonical_code(const uint8_t* depth, size_t len, uint16_t* bits) {
  uint16_t per_len[16] = {0}, next[16];
  for (size_t i = 0; i < len; i++) per_len[depth[i]]++;
  per_len[0] = 0;
  next[0] = 0;
  unsigned code = 0;
  for (int l = 1; l < 16; l++) {
    code = (code + per_len[l - 1]) << 1;
    next[l] = (uint16_t)code;
  }
  for (size_t i = 0; i < len; i++) {
    if (!depth[i]) continue;
    const unsigned l = depth[i];
    unsigned v = next[l]++, r = 0;
    for (unsigned b = 0; b < l; b++) r |= ((v >> b) & 1u) << (l - 1 - b);
    bits[i] = (uint16_t)r;
  }
}

/* ------------------------------------------------------------------ codes */
typedef struct {
  uint8_t depth[BS_ALPHABET];
  uint16_t bits[BS_ALPHABET];
  int used_symbols;
} prefix_code;

typedef struct {
  uint32_t count[BS_ALPHABET];
  size_t total;
  size_t cost; /* enc_cluster.cc:18-26, "not kept up-to-date" */
} histogram;

typedef struct {
  size_t num_contexts;      /* entries of context_map */
  uint8_t context_map[256]; /* context (or histogram index) -> cluster */
  size_t num_codes;
  prefix_code code[BS_ALPHABET];
  const uint8_t* static_map; /* pre-clustering that produced the histogram indices, or NULL */
  size_t num_static;
} entropy_code;

static int g_reference_single_symbol = 0;

static void hist_add(histogram* h, const histogram* o) {
  for (int i = 0; i < BS_ALPHABET; i++) h->count[i] += o->count[i];
  h->total += o->total;
}

static void hist_cost(histogram* h) { /* enc_cluster.cc:18-26 */
  h->cost = 0;
  if (h->total == 0) return;
  uint8_t d[BS_ALPHABET] = {0};
  huffman_depths(h->count, BS_ALPHABET, 15, d);
  for (int i = 0; i < BS_ALPHABET; i++) h->cost += (size_t)h->count[i] * d[i];
}

static float hist_distance(const histogram* a, const histogram* b) { /* enc_cluster.cc:28-35 */
  if (a->total == 0 || b->total == 0) return 0.0f;
  histogram both = *a;
  hist_add(&both, b);
  hist_cost(&both);
  return (float)(size_t)(both.cost - a->cost - b->cost); /* unsigned arithmetic, as written there */
}

/* enc_cluster.cc:37-131: greedy seeding with the farthest histogram, at most 8 clusters, the
 * rest joins its cheapest cluster; clusters renumbered by first use.  `h` is replaced by the
 * cluster histograms; returns their number. */
static size_t cluster_histograms(histogram* h, size_t n, uint8_t* map) {
  if (n <= 1) {
    if (n == 1) map[0] = 0;
    return n;
  }
  const size_t limit = n < BS_CLUSTER_LIMIT ? n : BS_CLUSTER_LIMIT;
  histogram* in = (histogram*)malloc(n * sizeof(histogram));
  memcpy(in, h, n * sizeof(histogram));
  histogram out[BS_CLUSTER_LIMIT];
  size_t nout = 0;
  uint32_t* sym = (uint32_t*)malloc(n * sizeof(uint32_t));
  float* far = (float*)malloc(n * sizeof(float));
  size_t pick = 0;
  for (size_t i = 0; i < n; i++) {
    sym[i] = (uint32_t)limit;
    far[i] = FLT_MAX;
    if (in[i].total == 0) {
      sym[i] = 0;
      far[i] = 0.0f;
      continue;
    }
    hist_cost(&in[i]);
    if (in[i].total > in[pick].total) pick = i;
  }
  while (nout < limit) {
    sym[pick] = (uint32_t)nout;
    out[nout++] = in[pick];
    far[pick] = 0.0f;
    pick = 0;
    for (size_t i = 0; i < n; i++) {
      if (far[i] == 0.0f) continue;
      const float d = hist_distance(&in[i], &out[nout - 1]);
      if (d < far[i]) far[i] = d;
      if (far[i] > far[pick]) pick = i;
    }
    if (far[pick] < 64.0f) break; /* kMinDistanceForDistinct */
  }
  for (size_t i = 0; i < n; i++) {
    if (sym[i] != limit) continue;
    size_t best = 0;
    float best_d = hist_distance(&in[i], &out[0]);
    for (size_t j = 1; j < nout; j++) {
      const float d = hist_distance(&in[i], &out[j]);
      if (d < best_d) {
        best = j;
        best_d = d;
      }
    }
    hist_add(&out[best], &in[i]);
    hist_cost(&out[best]);
    sym[i] = (uint32_t)best;
  }
  /* canonical numbering (enc_cluster.cc:91-111) */
  int renumber[BS_CLUSTER_LIMIT + 1];
  for (size_t j = 0; j <= BS_CLUSTER_LIMIT; j++) renumber[j] = -1;
  size_t next = 0;
  for (size_t i = 0; i < n; i++) {
    if (renumber[sym[i]] < 0) {
      renumber[sym[i]] = (int)next;
      h[next++] = out[sym[i]];
    }
    map[i] = (uint8_t)renumber[sym[i]];
  }
  free(in);
  free(sym);
  free(far);
  return next;
}

/* enc_entropy_code.cc:439-453 */
static void build_prefix_codes(const histogram* h, size_t n, entropy_code* ec) {
  ec->num_codes = n;
  for (size_t c = 0; c < n; c++) {
    prefix_code* pc = &ec->code[c];
    memset(pc, 0, sizeof(*pc));
    size_t len = BS_ALPHABET;
    while (len > 0 && h[c].count[len - 1] == 0) len--;
    huffman_depths(h[c].count, len, 15, pc->depth);
    canonical_code(pc->depth, len, pc->bits);
    for (size_t i = 0; i < len; i++) pc->used_symbols += pc->depth[i] != 0;
  }
}

/* enc_entropy_code.h:34-42 (+ the single-symbol switch, see the header of this file) */
static void put_token(bitsink* s, const entropy_code* ec, uint32_t context, uint32_t value) {
  uint32_t sym, nbits, extra;
  hybrid_uint(value, &sym, &nbits, &extra);
  const prefix_code* pc = &ec->code[ec->context_map[context]];
  unsigned depth = pc->depth[sym];
  uint64_t data = pc->bits[sym];
  if (pc->used_symbols == 1 && !g_reference_single_symbol) {
    depth = 0;
    data = 0;
  }
  data |= (uint64_t)extra << depth;
  sink_put(s, depth + nbits, data);
}

/* ---- serialisation of one prefix code (enc_entropy_code.cc:18-386) */
static void put_varlen_u16(bitsink* s, size_t n) { /* :317-327 */
  if (n == 0) {
    sink_put(s, 1, 0);
    return;
  }
  sink_put(s, 1, 1);
  const unsigned nb = floor_log2((uint32_t)n);
  sink_put(s, 4, nb);
  sink_put(s, nb, n - ((size_t)1 << nb));
}

/* run-length form of the code lengths (:109-260) */
typedef struct {
  uint8_t sym[2 * BS_ALPHABET];
  uint8_t extra[2 * BS_ALPHABET];
  size_t n;
} rle_lengths;

static void rle_reverse_tail(rle_lengths* r, size_t start) {
  for (size_t a = start, b = r->n; a + 1 < b; a++, b--) {
    uint8_t t = r->sym[a];
    r->sym[a] = r->sym[b - 1];
    r->sym[b - 1] = t;
    t = r->extra[a];
    r->extra[a] = r->extra[b - 1];
    r->extra[b - 1] = t;
  }
}
static void rle_emit(rle_lengths* r, uint8_t sym, uint8_t extra) {
  r->sym[r->n] = sym;
  r->extra[r->n] = extra;
  r->n++;
}
static void rle_run(rle_lengths* r, uint8_t prev, uint8_t value, size_t reps) {
  if (value == 0) { /* :165-194 */
    if (reps == 11) {
      rle_emit(r, 0, 0);
      reps--;
    }
    if (reps < 3) {
      for (size_t i = 0; i < reps; i++) rle_emit(r, 0, 0);
      return;
    }
    reps -= 3;
    const size_t start = r->n;
    for (;;) {
      rle_emit(r, 17, (uint8_t)(reps & 7));
      reps >>= 3;
      if (reps == 0) break;
      reps--;
    }
    rle_reverse_tail(r, start);
    return;
  }
  /* :123-163 */
  if (prev != value) {
    rle_emit(r, value, 0);
    reps--;
  }
  if (reps == 7) {
    rle_emit(r, value, 0);
    reps--;
  }
  if (reps < 3) {
    for (size_t i = 0; i < reps; i++) rle_emit(r, value, 0);
    return;
  }
  reps -= 3;
  const size_t start = r->n;
  for (;;) {
    rle_emit(r, 16, (uint8_t)(reps & 3));
    reps >>= 2;
    if (reps == 0) break;
    reps--;
  }
  rle_reverse_tail(r, start);
}

static void rle_code_lengths(const uint8_t* depth, size_t length, rle_lengths* r) { /* :224-260 */
  r->n = 0;
  size_t used = length;
  while (used > 0 && depth[used - 1] == 0) used--;
  int rle_nonzero = 0, rle_zero = 0;
  if (length > 50) { /* :196-222 */
    size_t tot_z = 0, tot_nz = 0, cnt_z = 1, cnt_nz = 1;
    for (size_t i = 0; i < used;) {
      size_t reps = 1;
      while (i + reps < used && depth[i + reps] == depth[i]) reps++;
      if (reps >= 3 && depth[i] == 0) {
        tot_z += reps;
        cnt_z++;
      }
      if (reps >= 4 && depth[i] != 0) {
        tot_nz += reps;
        cnt_nz++;
      }
      i += reps;
    }
    rle_nonzero = tot_nz > cnt_nz * 2;
    rle_zero = tot_z > cnt_z * 2;
  }
  uint8_t prev = 8;
  for (size_t i = 0; i < used;) {
    const uint8_t v = depth[i];
    size_t reps = 1;
    if ((v != 0 && rle_nonzero) || (v == 0 && rle_zero))
      while (i + reps < used && depth[i + reps] == v) reps++;
    rle_run(r, prev, v, reps);
    if (v != 0) prev = v;
    i += reps;
  }
}

static void put_complex_code(bitsink* s, const uint8_t* depth, size_t num) { /* :317-381 StoreHuffmanTree */
  rle_lengths r;
  rle_code_lengths(depth, num, &r);
  uint32_t hist[18] = {0};
  for (size_t i = 0; i < r.n; i++) hist[r.sym[i]]++;
  int distinct = 0, only = 0;
  for (int i = 0; i < 18; i++) {
    if (!hist[i]) continue;
    if (distinct == 0) {
      only = i;
      distinct = 1;
    } else {
      distinct = 2;
      break;
    }
  }
  uint8_t cl_depth[18] = {0};
  uint16_t cl_bits[18] = {0};
  huffman_depths(hist, 18, 5, cl_depth);
  canonical_code(cl_depth, 18, cl_bits);
  /* code-length code lengths (:18-66): fixed order, fixed 2..4-bit code, optional skip of
   * the first 2 or 3, trailing zeros dropped */
  static const uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t len_sym[6] = {0, 7, 3, 2, 1, 15};
  static const uint8_t len_bits[6] = {2, 4, 3, 2, 2, 4};
  size_t keep = 18;
  if (distinct > 1)
    while (keep > 0 && cl_depth[order[keep - 1]] == 0) keep--;
  size_t skip = 0;
  if (cl_depth[order[0]] == 0 && cl_depth[order[1]] == 0) skip = cl_depth[order[2]] == 0 ? 3 : 2;
  sink_put(s, 2, skip);
  for (size_t i = skip; i < keep; i++) sink_put(s, len_bits[cl_depth[order[i]]], len_sym[cl_depth[order[i]]]);
  if (distinct == 1) cl_depth[only] = 0;
  for (size_t i = 0; i < r.n; i++) { /* :68-85 */
    sink_put(s, cl_depth[r.sym[i]], cl_bits[r.sym[i]]);
    if (r.sym[i] == 16) sink_put(s, 2, r.extra[i]);
    if (r.sym[i] == 17) sink_put(s, 3, r.extra[i]);
  }
}

static void put_prefix_code(bitsink* s, const prefix_code* pc) { /* :329-366 WritePrefixCode */
  size_t count = 0, first4[4] = {0, 0, 0, 0}, length = 0;
  for (size_t i = 0; i < BS_ALPHABET; i++) {
    if (!pc->depth[i]) continue;
    if (count < 4) first4[count] = i;
    count++;
    length = i + 1;
  }
  unsigned max_bits = 0;
  for (size_t t = length - 1; t; t >>= 1) max_bits++;
  if (count <= 1) {
    sink_put(s, 4, 1);
    sink_put(s, max_bits, first4[0]);
    return;
  }
  if (count > 4) {
    put_complex_code(s, pc->depth, length);
    return;
  }
  /* simple code (:87-121): symbols ordered by depth (selection order of the reference) */
  sink_put(s, 2, 1);
  sink_put(s, 2, count - 1);
  for (size_t i = 0; i < count; i++)
    for (size_t j = i + 1; j < count; j++)
      if (pc->depth[first4[j]] < pc->depth[first4[i]]) {
        const size_t t = first4[j];
        first4[j] = first4[i];
        first4[i] = t;
      }
  for (size_t i = 0; i < count; i++) sink_put(s, max_bits, first4[i]);
  if (count == 4) sink_put(s, 1, pc->depth[first4[0]] == 1 ? 1 : 0);
}

static void put_prefix_codes(bitsink* s, const prefix_code* codes, size_t n) { /* :388-417 */
  sink_put(s, 1, 1); /* use_prefix_code */
  for (size_t i = 0; i < n; i++) {
    sink_put(s, 4, 4); /* split_exponent */
    sink_put(s, 3, 2); /* msb_in_token */
    sink_put(s, 2, 0); /* lsb_in_token */
  }
  size_t alphabet[BS_ALPHABET];
  for (size_t c = 0; c < n; c++) {
    alphabet[c] = 1;
    for (size_t i = 0; i < BS_ALPHABET; i++)
      if (codes[c].depth[i]) alphabet[c] = i + 1;
    put_varlen_u16(s, alphabet[c] - 1);
  }
  for (size_t c = 0; c < n; c++)
    if (alphabet[c] > 1) put_prefix_code(s, &codes[c]);
}

/* enc_entropy_code.cc:498-541.  The serialised map covers the original contexts: the static
 * pre-clustering composed with the optimised map when there is one. */
static void put_context_map(bitsink* s, const entropy_code* ec) {
  const size_t n = ec->static_map ? ec->num_static : ec->num_contexts;
  if (n == 0) return;
  uint8_t top = 0;
  for (size_t i = 0; i < ec->num_contexts; i++)
    if (ec->context_map[i] > top) top = ec->context_map[i];
  if (top == 0) {
    sink_put(s, 3, 1); /* simple, 0 bits per entry */
    return;
  }
  sink_put(s, 3, 0); /* not simple, no MTF, no LZ77 */
  histogram h;
  memset(&h, 0, sizeof(h));
  uint8_t* entry = (uint8_t*)malloc(n);
  for (size_t i = 0; i < n; i++) {
    entry[i] = ec->static_map ? ec->context_map[ec->static_map[i]] : ec->context_map[i];
    uint32_t sym, nb, ex;
    hybrid_uint(entry[i], &sym, &nb, &ex);
    h.count[sym]++;
    h.total++;
  }
  entropy_code mc;
  memset(&mc, 0, sizeof(mc));
  mc.num_contexts = 1;
  build_prefix_codes(&h, 1, &mc);
  put_prefix_codes(s, mc.code, 1);
  for (size_t i = 0; i < n; i++) put_token(s, &mc, 0, entry[i]);
  free(entry);
}

static void put_entropy_code(bitsink* s, const entropy_code* ec) { /* :543-546 */
  put_context_map(s, ec);
  put_prefix_codes(s, ec->code, ec->num_codes);
}

/* ------------------------------------------------------------------ raw records */
typedef struct {
  uint8_t* p;
  size_t n, cap;
} recbuf;

static void rec_put(recbuf* r, uint8_t ctx, uint32_t value) { /* Write(8, ctx); Write(16, value) */
  if (r->n + 3 > r->cap) {
    r->cap = r->cap ? r->cap * 2 : 4096;
    r->p = (uint8_t*)realloc(r->p, r->cap);
  }
  r->p[r->n++] = ctx;
  r->p[r->n++] = (uint8_t)(value & 0xFF);
  r->p[r->n++] = (uint8_t)((value >> 8) & 0xFF);
}

static int32_t clamped_gradient(int32_t n, int32_t w, int32_t l) { /* enc_frame.cc:158-176 */
  const int32_t lo = n < w ? n : w, hi = n < w ? w : n;
  const int32_t g = (int32_t)((uint32_t)n + (uint32_t)w - (uint32_t)l);
  const int32_t a = l < lo ? hi : g;
  return l > hi ? lo : a;
}

typedef struct {
  size_t xsize, ysize;
  size_t xblocks, yblocks, xtiles, ytiles, xgroups, ygroups, xdc, ydc;
} geometry;

static geometry make_geometry(size_t xsize, size_t ysize) { /* enc_frame.cc:48-93 */
  geometry g;
  g.xsize = xsize;
  g.ysize = ysize;
  g.xblocks = div_ceil(xsize, 8);
  g.yblocks = div_ceil(ysize, 8);
  g.xtiles = div_ceil(xsize, 64);
  g.ytiles = div_ceil(ysize, 64);
  g.xgroups = div_ceil(xsize, 256);
  g.ygroups = div_ceil(ysize, 256);
  g.xdc = div_ceil(xsize, 2048);
  g.ydc = div_ceil(ysize, 2048);
  return g;
}

static uint8_t strategy_code(uint8_t cell) { /* ac_strategy.h:59-62 */
  static const uint8_t lut[3] = {0, 6, 7};
  return lut[cell >> 1];
}

/* WriteDCGroup in its OPTIMIZE_CODE form (enc_frame.cc:536-570): the raw records of DC group
 * (dx, dy), cut out of the image-absolute grids.  Contexts go through the static DC map,
 * which is the identity in this configuration (static_entropy_codes.h:18-24). */
static void dc_group_records(const orc_bs_input* in, const geometry* g, size_t dx, size_t dy, recbuf* r) {
  const size_t bx0 = dx * 256, by0 = dy * 256;
  const size_t nbx = g->xblocks - bx0 < 256 ? g->xblocks - bx0 : 256;
  const size_t nby = g->yblocks - by0 < 256 ? g->yblocks - by0 : 256;
  const size_t tx0 = dx * 32, ty0 = dy * 32;
  const size_t ntx = div_ceil(nbx * 8, 64), nty = div_ceil(nby * 8, 64);
  rec_put(r, BS_MAX_CONTEXTS + 6, 12); /* extra_dc_precision (2 bits) + modular group header (4 bits) */
  /* WriteDCTokens (:287-316): channel order Y, X, B */
  static const int order[3] = {1, 0, 2};
  for (int k = 0; k < 3; k++) {
    const int16_t* q = in->quant_dc[order[k]];
    for (size_t y = 0; y < nby; y++) {
      const int16_t* row = q + (by0 + y) * g->xblocks + bx0;
      const int16_t* up = row - g->xblocks;
      for (size_t x = 0; x < nbx; x++) {
        const int64_t left = x ? row[x - 1] : (y ? up[x] : 0);
        const int64_t top = y ? up[x] : left;
        const int64_t topleft = (x && y) ? up[x - 1] : left;
        const int32_t guess = clamped_gradient((int32_t)top, (int32_t)left, (int32_t)topleft);
        int64_t prop = 512 + top + left - topleft;
        prop = prop < 0 ? 0 : prop > 1023 ? 1023 : prop;
        rec_put(r, ORC_kGradientContextLut[prop], pack_signed((int32_t)row[x] - guess));
      }
    }
  }
  /* (:547-563) number of AC blocks, then the second modular group header */
  size_t first_blocks = 0;
  for (size_t y = 0; y < nby; y++)
    for (size_t x = 0; x < nbx; x++) first_blocks += in->ac_strategy[(by0 + y) * g->xblocks + bx0 + x] & 1;
  const unsigned nb_bits = ceil_log2(nbx * nby);
  if (nb_bits) rec_put(r, (uint8_t)(BS_MAX_CONTEXTS + nb_bits), (uint32_t)(first_blocks - 1));
  rec_put(r, BS_MAX_CONTEXTS + 4, 3);
  /* WriteACMetadataTokens (:329-424) */
  for (int c = 0; c < 2; c++) {
    const int8_t* map = c == 0 ? in->ytox_map : in->ytob_map;
    for (size_t y = 0; y < nty; y++) {
      const int8_t* row = map + (ty0 + y) * g->xtiles + tx0;
      const int8_t* up = row - g->xtiles;
      for (size_t x = 0; x < ntx; x++) {
        const int32_t left = x ? row[x - 1] : (y ? up[x] : 0);
        const int32_t top = y ? up[x] : left;
        const int32_t topleft = (x && y) ? up[x - 1] : left;
        rec_put(r, (uint8_t)(2 - c), pack_signed((int32_t)row[x] - clamped_gradient(top, left, topleft)));
      }
    }
  }
  int32_t prev = 0;
  for (size_t y = 0; y < nby; y++)
    for (size_t x = 0; x < nbx; x++) {
      const uint8_t cell = in->ac_strategy[(by0 + y) * g->xblocks + bx0 + x];
      if (!(cell & 1)) continue;
      const int32_t cur = strategy_code(cell);
      rec_put(r, prev > 11 ? 7 : prev > 5 ? 8 : prev > 3 ? 9 : 10, pack_signed(cur));
      prev = cur;
    }
  prev = strategy_code(in->ac_strategy[by0 * g->xblocks + bx0]);
  for (size_t y = 0; y < nby; y++)
    for (size_t x = 0; x < nbx; x++) {
      const size_t at = (by0 + y) * g->xblocks + bx0 + x;
      if (!(in->ac_strategy[at] & 1)) continue;
      const int32_t cur = (int32_t)in->raw_quant_field[at] - 1;
      rec_put(r, prev > 11 ? 3 : prev > 5 ? 4 : prev > 3 ? 5 : 6, pack_signed(cur - prev));
      prev = cur;
    }
  for (size_t i = 0; i < nbx * nby; i++) rec_put(r, 0, pack_signed(4)); /* EPF sharpness */
}

/* OptimizeSections (enc_frame.cc:766-802): histograms of all sections' records -> clustered
 * code -> every section re-written with it.  `static_map` = the pre-clustering whose indices
 * the records carry. */
static void histograms_of_records(const uint8_t* rec, size_t bytes, histogram* h) {
  for (size_t j = 0; j + 2 < bytes; j += 3) {
    if (rec[j] >= BS_MAX_CONTEXTS) continue;
    uint32_t sym, nb, ex;
    hybrid_uint((uint32_t)rec[j + 1] | ((uint32_t)rec[j + 2] << 8), &sym, &nb, &ex);
    h[rec[j]].count[sym]++;
    h[rec[j]].total++;
  }
}

static void code_from_histograms(histogram* h, size_t n, const uint8_t* static_map, size_t num_static,
                                 entropy_code* ec) { /* enc_entropy_code.cc:476-487 */
  memset(ec, 0, sizeof(*ec));
  ec->num_contexts = n;
  const size_t clusters = cluster_histograms(h, n, ec->context_map);
  ec->static_map = static_map;
  ec->num_static = num_static;
  build_prefix_codes(h, clusters, ec);
}

static void rewrite_section(const uint8_t* rec, size_t bytes, const entropy_code* ec, bitsink* s) {
  for (size_t j = 0; j + 2 < bytes; j += 3) {
    const uint32_t value = (uint32_t)rec[j + 1] | ((uint32_t)rec[j + 2] << 8);
    if (rec[j] >= BS_MAX_CONTEXTS)
      sink_put(s, rec[j] - BS_MAX_CONTEXTS, value);
    else
      put_token(s, ec, rec[j], value);
  }
}

/* ------------------------------------------------------------------ headers and globals */
static void put_size(bitsink* s, uint32_t size_minus_1) { /* enc_file.cc:27-38 */
  static const unsigned kBits[4] = {9, 13, 18, 30};
  for (unsigned i = 0; i < 4; i++)
    if (size_minus_1 < (1u << kBits[i])) {
      sink_put(s, 2, i);
      sink_put(s, kBits[i], size_minus_1);
      return;
    }
}

static void put_quant_scales(bitsink* s, int global_scale, int quant_dc) { /* enc_frame.cc:459-488 */
  if (global_scale < 2049) {
    sink_put(s, 2, 0);
    sink_put(s, 11, (uint64_t)(global_scale - 1));
  } else if (global_scale < 4097) {
    sink_put(s, 2, 1);
    sink_put(s, 11, (uint64_t)(global_scale - 2049));
  } else if (global_scale < 8193) {
    sink_put(s, 2, 2);
    sink_put(s, 12, (uint64_t)(global_scale - 4097));
  } else {
    sink_put(s, 2, 3);
    sink_put(s, 16, (uint64_t)(global_scale - 8193));
  }
  if (quant_dc == 16) {
    sink_put(s, 2, 0);
  } else if (quant_dc < 33) {
    sink_put(s, 2, 1);
    sink_put(s, 5, (uint64_t)(quant_dc - 1));
  } else if (quant_dc < 257) {
    sink_put(s, 2, 2);
    sink_put(s, 8, (uint64_t)(quant_dc - 1));
  } else {
    sink_put(s, 2, 3);
    sink_put(s, 16, (uint64_t)(quant_dc - 1));
  }
}

static void put_context_tree(bitsink* s, size_t num_dc_groups) { /* enc_frame.cc:490-505 */
  enum { kTokens = 313, kTreeContexts = 6 };
  uint32_t ctx[kTokens], val[kTokens];
  for (int i = 0; i < kTokens; i++) {
    ctx[i] = ORC_kContextTreeTokens[2 * i];
    val[i] = ORC_kContextTreeTokens[2 * i + 1];
  }
  val[1] = pack_signed((int32_t)(1 + num_dc_groups));
  histogram h[kTreeContexts];
  memset(h, 0, sizeof(h));
  for (int i = 0; i < kTokens; i++) {
    uint32_t sym, nb, ex;
    hybrid_uint(val[i], &sym, &nb, &ex);
    h[ctx[i]].count[sym]++;
    h[ctx[i]].total++;
  }
  entropy_code ec;
  code_from_histograms(h, kTreeContexts, NULL, 0, &ec);
  sink_put(s, 1, 1); /* tree present */
  sink_put(s, 1, 0); /* no lz77 */
  put_entropy_code(s, &ec);
  for (int i = 0; i < kTokens; i++) put_token(s, &ec, ctx[i], val[i]);
}

static void put_dc_global(bitsink* s, const orc_distance_params* dp, size_t num_dc_groups,
                          const entropy_code* dc_code) { /* enc_frame.cc:507-523 */
  sink_put(s, 1, 1); /* default DC dequantisation */
  put_quant_scales(s, dp->global_scale, dp->quant_dc);
  sink_put(s, 1, 0);  /* non-default block context map */
  sink_put(s, 16, 0); /* no DC thresholds, no quant-field thresholds */
  {
    entropy_code bm;
    memset(&bm, 0, sizeof(bm));
    bm.num_contexts = sizeof(ORC_kCompactBlockContextMap);
    memcpy(bm.context_map, ORC_kCompactBlockContextMap, sizeof(ORC_kCompactBlockContextMap));
    put_context_map(s, &bm);
  }
  sink_put(s, 1, 1); /* default chroma-from-luma DC part */
  put_context_tree(s, num_dc_groups);
  sink_put(s, 1, 0); /* no lz77 */
  put_entropy_code(s, dc_code);
}

static void put_ac_global(bitsink* s, size_t num_groups, const entropy_code* ac_code) { /* :525-534 */
  sink_put(s, 1, 1); /* default quant matrices */
  const unsigned nb = ceil_log2(num_groups);
  if (nb) sink_put(s, nb, 0);
  sink_put(s, 2, 3);
  sink_put(s, 13, 0); /* default coefficient orders */
  sink_put(s, 1, 0);  /* no lz77 */
  put_entropy_code(s, ac_code);
}

static void put_frame_header(bitsink* s, const orc_distance_params* dp) { /* enc_frame.cc:426-457 */
  sink_put(s, 1, 0);
  sink_put(s, 2, 0);
  sink_put(s, 1, 0);
  sink_put(s, 2, 2);
  sink_put(s, 8, 111);
  sink_put(s, 2, 0);
  sink_put(s, 3, dp->x_qm_scale);
  sink_put(s, 3, 2);
  sink_put(s, 2, 0);
  sink_put(s, 1, 0);
  sink_put(s, 2, 0);
  sink_put(s, 1, 1);
  sink_put(s, 2, 0);
  if (dp->epf_iters == 2) {
    sink_put(s, 1, 1);
  } else {
    sink_put(s, 1, 0);
    sink_put(s, 1, 0);
    sink_put(s, 2, dp->epf_iters);
    if (dp->epf_iters > 0) sink_put(s, 3, 0);
    sink_put(s, 2, 0);
  }
  sink_put(s, 2, 0);
}

static void put_file_header(bitsink* s, size_t xsize, size_t ysize) { /* enc_file.cc:40-94 */
  sink_put(s, 8, 0xFF);
  sink_put(s, 8, 0x0A);
  sink_put(s, 1, 0);
  put_size(s, (uint32_t)ysize - 1);
  sink_put(s, 3, 0);
  put_size(s, (uint32_t)xsize - 1);
  sink_put(s, 1, 0);
  sink_put(s, 1, 0);
  sink_put(s, 1, 1);
  sink_put(s, 2, 0);
  sink_put(s, 4, 7);
  sink_put(s, 1, 0);
  sink_put(s, 2, 0);
  sink_put(s, 1, 1);
  sink_put(s, 1, 0);
  sink_put(s, 1, 0);
  sink_put(s, 2, 0);
  sink_put(s, 2, 1);
  sink_put(s, 2, 1);
  sink_put(s, 1, 0);
  sink_put(s, 2, 2);
  sink_put(s, 4, 6);
  sink_put(s, 2, 1);
  sink_put(s, 2, 0);
  sink_put(s, 1, 1);
  sink_pad_to_byte(s);
}

/* ------------------------------------------------------------------ entry points */
static void table_of(const entropy_code* ec, size_t nctx, uint32_t* table) {
  memset(table, 0, 64 * 64 * sizeof(uint32_t));
  for (size_t c = 0; c < nctx; c++) {
    const prefix_code* pc = &ec->code[ec->context_map[c]];
    for (int sym = 0; sym < BS_ALPHABET; sym++) {
      unsigned depth = pc->depth[sym];
      unsigned bits = pc->bits[sym];
      if (pc->used_symbols == 1 && !g_reference_single_symbol) depth = bits = 0;
      table[c * 64 + sym] = (depth << 16) | bits;
    }
  }
}

void orc_bs_build_code_tables(const uint32_t* ac_hist, const uint32_t* dc_hist, int reference_single_symbol,
                              uint32_t* ac_table, uint32_t* dc_table) {
  g_reference_single_symbol = reference_single_symbol;
  for (int kind = 0; kind < 2; kind++) {
    const size_t n = kind ? 45 : 64;
    const uint32_t* src = kind ? dc_hist : ac_hist;
    histogram h[64];
    memset(h, 0, sizeof(h));
    for (size_t c = 0; c < n; c++)
      for (int i = 0; i < BS_ALPHABET; i++) {
        h[c].count[i] = src[c * 64 + i];
        h[c].total += src[c * 64 + i];
      }
    entropy_code ec;
    code_from_histograms(h, n, NULL, 0, &ec);
    table_of(&ec, n, kind ? dc_table : ac_table);
  }
}

int orc_bs_dc_group_records(const orc_bs_input* in, size_t dc_group, uint8_t** out, size_t* out_size) {
  if (!in || !out || !out_size) return 1;
  const geometry g = make_geometry(in->xsize, in->ysize);
  if (dc_group >= g.xdc * g.ydc) return 1;
  recbuf r = {0, 0, 0};
  dc_group_records(in, &g, dc_group % g.xdc, dc_group / g.xdc, &r);
  *out = r.p;
  *out_size = r.n;
  return 0;
}

int orc_bs_encode_file(const orc_bs_input* in, float distance, int reference_single_symbol, uint8_t** out,
                       size_t* out_size) {
  if (!in || !out || !out_size || in->xsize == 0 || in->ysize == 0) return 1;
  /* enc_file.cc:57-68 */
  if (!(distance > 0.0f)) return 1;
  if (distance <= 0.03f) distance = 0.03f;
  g_reference_single_symbol = reference_single_symbol;
  const geometry g = make_geometry(in->xsize, in->ysize);
  const size_t ndc = g.xdc * g.ydc, ngroups = g.xgroups * g.ygroups;
  orc_distance_params dp;
  orc_compute_distance_params(distance, &dp);

  /* DC-group sections as raw records, then both code optimisations (enc_frame.cc:846-850) */
  recbuf* dcrec = (recbuf*)calloc(ndc, sizeof(recbuf));
  histogram hist[64];
  memset(hist, 0, sizeof(hist));
  for (size_t i = 0; i < ndc; i++) {
    dc_group_records(in, &g, i % g.xdc, i / g.xdc, &dcrec[i]);
    histograms_of_records(dcrec[i].p, dcrec[i].n, hist);
  }
  entropy_code dc_code, ac_code;
  code_from_histograms(hist, 45, NULL, 0, &dc_code); /* static DC map = identity over 45 contexts */
  dc_code.static_map = NULL;
  memset(hist, 0, sizeof(hist));
  for (size_t i = 0; i < ngroups; i++) histograms_of_records(in->group_tokens[i], in->group_token_bytes[i], hist);
  code_from_histograms(hist, 64, ORC_kACContextMap, sizeof(ORC_kACContextMap), &ac_code);

  /* sections: DCGlobal, DC groups, ACGlobal, AC groups (enc_frame.cc:834, 853-854) */
  const size_t nsec = 2 + ndc + ngroups;
  bitsink* sec = (bitsink*)calloc(nsec, sizeof(bitsink));
  put_dc_global(&sec[0], &dp, ndc, &dc_code);
  for (size_t i = 0; i < ndc; i++) {
    rewrite_section(dcrec[i].p, dcrec[i].n, &dc_code, &sec[1 + i]);
    free(dcrec[i].p);
  }
  free(dcrec);
  put_ac_global(&sec[1 + ndc], ngroups, &ac_code);
  for (size_t i = 0; i < ngroups; i++)
    rewrite_section(in->group_tokens[i], in->group_token_bytes[i], &ac_code, &sec[2 + ndc + i]);

  bitsink file = {0, 0, 0};
  put_file_header(&file, in->xsize, in->ysize);
  put_frame_header(&file, &dp);
  /* CombineSections (enc_frame.cc:804-816) + WriteTOC (:572-595) */
  size_t used = nsec;
  if (nsec == 4) {
    for (size_t i = 1; i < 4; i++) sink_append_bits(&sec[0], &sec[i]);
    used = 1;
  }
  sink_put(&file, 1, 0); /* no permutation */
  sink_pad_to_byte(&file);
  for (size_t i = 0; i < used; i++) {
    const size_t size = sink_bytes(&sec[i]);
    static const unsigned kBits[4] = {10, 14, 22, 30};
    size_t base = 0;
    for (unsigned k = 0; k < 4; k++) {
      if (size < base + ((size_t)1 << kBits[k])) {
        sink_put(&file, 2, k);
        sink_put(&file, kBits[k], size - base);
        break;
      }
      base += (size_t)1 << kBits[k];
    }
  }
  sink_pad_to_byte(&file);
  for (size_t i = 0; i < used; i++) {
    const size_t size = sink_bytes(&sec[i]);
    sink_reserve(&file, size * 8);
    if (size) memcpy(file.data + (file.bits >> 3), sec[i].data, size);
    file.bits += size * 8;
  }
  for (size_t i = 0; i < nsec; i++) sink_free(&sec[i]);
  free(sec);
  *out_size = sink_bytes(&file);
  *out = file.data;
  return 0;
}

void orc_bs_free(void* p) { free(p); }
