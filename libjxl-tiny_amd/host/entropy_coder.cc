// See entropy_coder.h.  The algorithms are the brotli-lineage ones the JPEG XL
// reference encoder uses; tie-breaking and iteration order are part of the
// output contract and are kept identical (citations inline).
#include "entropy_coder.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <thread>

namespace jxlt {
namespace {

struct Node {
  uint32_t count;
  int16_t left;   // -1 for a leaf
  int16_t right;  // child index, or the symbol for a leaf
};

void AssignDepths(const Node* pool, int root, uint8_t* depth) {
  // Iterative pre-order walk (enc_huffman_tree.cc:26-35 is the recursive form).
  struct Item { int node; uint8_t level; };
  Item stack[2 * kAlphabetSize + 8];
  int sp = 0;
  stack[sp++] = {root, 0};
  while (sp) {
    Item it = stack[--sp];
    const Node& n = pool[it.node];
    if (n.left >= 0) {
      stack[sp++] = {n.right, static_cast<uint8_t>(it.level + 1)};
      stack[sp++] = {n.left, static_cast<uint8_t>(it.level + 1)};
    } else {
      depth[n.right] = it.level;
    }
  }
}

}  // namespace

// enc_huffman_tree.cc:65-142.  Leaves are gathered from the highest symbol
// down, stably sorted by count, then merged with the classic two-queue scheme
// (ties prefer the leaf queue).  If the tree is deeper than tree_limit the
// minimum count is doubled and the construction repeated.
void CreateHuffmanTree(const uint32_t* counts, size_t length, int tree_limit, uint8_t* depth) {
  // Called ~1000 times per frame by the clustering (Distance): fixed-size node pool on the stack.
  if (length > kAlphabetSize) length = kAlphabetSize;
  Node tree[2 * kAlphabetSize + 2];
  for (uint32_t count_limit = 1;; count_limit *= 2) {
    // Stable sort by count of the leaves in gathering order (highest symbol first): the keys
    // (count, gathering position) are unique, so an ordinary sort of packed keys is stable.
    uint64_t keys[kAlphabetSize];
    size_t n = 0;
    for (size_t i = length; i != 0;) {
      --i;
      if (counts[i]) {
        keys[n] = (static_cast<uint64_t>(std::max(counts[i], count_limit - 1)) << 16) | (n << 8) | i;
        ++n;
      }
    }
    std::sort(keys, keys + n);
    for (size_t k = 0; k < n; ++k)
      tree[k] = {static_cast<uint32_t>(keys[k] >> 16), -1, static_cast<int16_t>(keys[k] & 0xFF)};
    if (n == 0) return;  // (unreachable from the encoder; reference would misbehave)
    if (n == 1) {
      depth[tree[0].right] = 1;  // "fake" depth, kept as-is by the callers
      return;
    }
    const Node sentinel = {std::numeric_limits<uint32_t>::max(), -1, -1};
    tree[n] = sentinel;
    tree[n + 1] = sentinel;  // first parent slot
    size_t size = n + 2;
    size_t leaf = 0, inner = n + 1;
    uint8_t height[2 * kAlphabetSize + 2] = {};  // of the subtree below each node (leaves: 0)
    for (size_t k = n - 1; k != 0; --k) {
      size_t l, r;
      if (tree[leaf].count <= tree[inner].count) l = leaf++; else l = inner++;
      if (tree[leaf].count <= tree[inner].count) r = leaf++; else r = inner++;
      const size_t parent = size - 1;
      tree[parent].count = tree[l].count + tree[r].count;
      tree[parent].left = static_cast<int16_t>(l);
      tree[parent].right = static_cast<int16_t>(r);
      height[parent] = static_cast<uint8_t>(std::max(height[l], height[r]) + 1);
      tree[size++] = sentinel;
    }
    // The deepest leaf sits at the root's height: too deep -> next limit without walking the tree.
    // (symbols outside this tree keep the depth the caller initialised, 0, as in the reference)
    if (height[2 * n - 1] > tree_limit) continue;
    AssignDepths(tree, static_cast<int>(2 * n - 1), depth);
    return;
  }
}

namespace {
uint16_t ReverseBits(int num_bits, uint16_t bits) {
  uint16_t r = 0;
  for (int i = 0; i < num_bits; ++i) r = static_cast<uint16_t>((r << 1) | ((bits >> i) & 1));
  return r;
}
}  // namespace

// enc_entropy_code.cc:297-324: canonical code assignment, bit-reversed for the
// LSB-first writer.
void ConvertBitDepthsToSymbols(const uint8_t* depth, size_t len, uint16_t* bits) {
  uint16_t bl_count[16] = {0};
  for (size_t i = 0; i < len; ++i) ++bl_count[depth[i]];
  bl_count[0] = 0;
  uint16_t next_code[16];
  next_code[0] = 0;
  int code = 0;
  for (int i = 1; i < 16; ++i) {
    code = (code + bl_count[i - 1]) << 1;
    next_code[i] = static_cast<uint16_t>(code);
  }
  for (size_t i = 0; i < len; ++i) {
    if (depth[i]) bits[i] = ReverseBits(depth[i], next_code[depth[i]]++);
  }
}

// ---------------------------------------------------------------------------
// Clustering (enc_cluster.cc)
// ---------------------------------------------------------------------------
namespace {

// Ascending sort of n <= 64 distinct keys.  With AVX-512 by ranks: a key's place is the number of keys below it --
// n * n / 8 vector comparisons without a branch; std::sort spends ~30 cycles per key on mispredicted comparisons
// at these sizes.
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512vl"))) void SortKeysByRank(uint64_t* keys, size_t n) {
  // (256-bit vectors: the 512-bit ones make some hosts lower their clock for a code path this short)
  alignas(64) uint64_t in[kAlphabetSize];
  const size_t vectors = (n + 3) >> 2;
  for (size_t i = 0; i < n; ++i) in[i] = keys[i];
  for (size_t i = n; i < 4 * vectors; ++i) in[i] = ~uint64_t(0);
  for (size_t i = 0; i < n; ++i) {
    const __m256i key = _mm256_set1_epi64x(static_cast<long long>(in[i]));
    unsigned below = 0;
    for (size_t j = 0; j < vectors; ++j) {
      const __m256i v = _mm256_load_si256(reinterpret_cast<const __m256i*>(in + 4 * j));
      below += static_cast<unsigned>(__builtin_popcount(_mm256_cmplt_epu64_mask(v, key)));
    }
    keys[below] = in[i];
  }
}
void SortKeys(uint64_t* keys, size_t n) {
  static const bool by_rank = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
  if (by_rank) {
    SortKeysByRank(keys, n);
  } else {
    std::sort(keys, keys + n);
  }
}
#else
void SortKeys(uint64_t* keys, size_t n) { std::sort(keys, keys + n); }
#endif

// sum of counts[i] * depth[i] over the code CreateHuffmanTree(counts, kAlphabetSize, 15) would build, without
// building the depths when the first tree is not deeper than 15 (then the sum is the sum of the inner nodes'
// weights): the clustering evaluates about a thousand such costs per frame and needs nothing else of them.
thread_local size_t t_cost_evaluations = 0;  // (diagnostics: JXLT_TRACE prints them per clustering)
size_t HuffmanBitCost(const uint32_t* counts) {
  ++t_cost_evaluations;
  alignas(64) uint64_t keys[kAlphabetSize];
  size_t n = 0;
  for (size_t i = kAlphabetSize; i != 0;) {  // the gathering order and the keys of CreateHuffmanTree, count_limit 1
    --i;
    if (counts[i]) {
      keys[n] = (static_cast<uint64_t>(counts[i]) << 16) | (n << 8) | i;
      ++n;
    }
  }
  if (n == 0) return 0;
  if (n == 1) return counts[keys[0] & 0xFF];  // ("fake" depth 1)
  SortKeys(keys, n);
  // two-queue merge (ties prefer the leaf queue); node weights wrap at 32 bits as the tree's do, the sums
  // that make up the cost do not
  uint32_t weight[2 * kAlphabetSize + 2];
  size_t exact[2 * kAlphabetSize + 2];
  uint8_t height[2 * kAlphabetSize + 2];
  for (size_t k = 0; k < n; ++k) {
    weight[k] = static_cast<uint32_t>(keys[k] >> 16);
    exact[k] = weight[k];
    height[k] = 0;
  }
  const uint32_t kSentinel = std::numeric_limits<uint32_t>::max();
  weight[n] = kSentinel;
  weight[n + 1] = kSentinel;
  size_t size = n + 2, leaf = 0, inner = n + 1, cost = 0;
  for (size_t k = n - 1; k != 0; --k) {
    size_t l, r;
    if (weight[leaf] <= weight[inner]) l = leaf++; else l = inner++;
    if (weight[leaf] <= weight[inner]) r = leaf++; else r = inner++;
    const size_t parent = size - 1;
    weight[parent] = weight[l] + weight[r];
    exact[parent] = exact[l] + exact[r];
    height[parent] = static_cast<uint8_t>(std::max(height[l], height[r]) + 1);
    cost += exact[parent];
    weight[size++] = kSentinel;
  }
  if (height[2 * n - 1] <= 15) return cost;
  // Deeper than the limit (four in ten of the clustering's evaluations on ordinary DC histograms): the later
  // rounds of CreateHuffmanTree -- every count below count_limit - 1 raised to it, count_limit doubled until the
  // tree fits -- without their sorts.  A round's leaf order follows from the first round's: the raised leaves
  // (equal counts) in gathering order, i.e. the set bits of a mask of gathering positions, then the others as they
  // are.  A round is given up at the first node that is too high (the root can only be higher), and only the
  // round that fits walks its tree.  (The general function here cost 5-9 k cycles per call, three quarters of
  // the clustering's time.)
  uint8_t symbol_at[kAlphabetSize];  // gathering position -> symbol
  for (size_t k = 0; k < n; ++k) symbol_at[(keys[k] >> 8) & 0xFF] = static_cast<uint8_t>(keys[k] & 0xFF);
  uint8_t leaf_symbol[kAlphabetSize];
  uint8_t left[2 * kAlphabetSize + 2], right[2 * kAlphabetSize + 2];
  size_t raised = 0;        // keys[0 .. raised) have counts <= count_limit - 1
  uint64_t raised_at = 0;   // their gathering positions
  // (count_limit = 2 -- a floor of 1 under counts that are all >= 1 -- is the round above over again, enc_huffman_tree.cc:71-80
  // with count_limit - 1 = 1: same weights, same order, same tree, too deep again.  The first round that can differ: 4.)
  for (uint32_t count_limit = 4;; count_limit *= 2) {
    const uint32_t floor = count_limit - 1;
    while (raised < n && (keys[raised] >> 16) <= floor) {
      raised_at |= uint64_t(1) << ((keys[raised] >> 8) & 0xFF);
      ++raised;
    }
    size_t k = 0;
    for (uint64_t m = raised_at; m != 0; m &= m - 1, ++k) {
      weight[k] = floor;
      leaf_symbol[k] = symbol_at[__builtin_ctzll(m)];
    }
    for (size_t q = raised; q < n; ++q, ++k) {
      weight[k] = static_cast<uint32_t>(keys[q] >> 16);
      leaf_symbol[k] = static_cast<uint8_t>(keys[q] & 0xFF);
    }
    weight[n] = kSentinel;
    weight[n + 1] = kSentinel;
    size = n + 2, leaf = 0, inner = n + 1;
    bool fits = true;
    for (size_t j = n - 1; j != 0; --j) {
      size_t l, r;
      if (weight[leaf] <= weight[inner]) l = leaf++; else l = inner++;
      if (weight[leaf] <= weight[inner]) r = leaf++; else r = inner++;
      const size_t parent = size - 1;
      weight[parent] = weight[l] + weight[r];
      left[parent] = static_cast<uint8_t>(l);
      right[parent] = static_cast<uint8_t>(r);
      height[parent] = static_cast<uint8_t>(std::max(height[l], height[r]) + 1);
      if (height[parent] > 15) {
        fits = false;
        break;
      }
      weight[size++] = kSentinel;
    }
    if (!fits) continue;
    // depths from the root down (children have smaller indices than their parent), as AssignDepths finds them
    uint8_t level[2 * kAlphabetSize + 2];
    level[2 * n - 1] = 0;
    for (size_t node = 2 * n - 1; node > n; --node) {
      level[left[node]] = level[right[node]] = static_cast<uint8_t>(level[node] + 1);
    }
    cost = 0;
    for (size_t q = 0; q < n; ++q) cost += static_cast<size_t>(counts[leaf_symbol[q]]) * level[q];
    return cost;
  }
}

void ComputeBitCost(Histogram* h) {  // enc_cluster.cc:18-26
  h->bit_cost = h->total_count == 0 ? 0 : HuffmanBitCost(h->counts);
}

// enc_cluster.cc:28-35; *combined_cost (optional): the bit cost of the two histograms added up
float Distance(const Histogram& a, const Histogram& b, size_t* combined_cost = nullptr) {
  if (a.total_count == 0 || b.total_count == 0) {
    if (combined_cost) *combined_cost = a.bit_cost + b.bit_cost;  // (the cost of the one that is not empty)
    return 0;
  }
  uint32_t counts[kAlphabetSize];
  for (size_t i = 0; i < kAlphabetSize; ++i) counts[i] = a.counts[i] + b.counts[i];
  const size_t cost = HuffmanBitCost(counts);
  if (combined_cost) *combined_cost = cost;
  // size_t arithmetic (may wrap) converted to float, as in the reference.
  return static_cast<float>(cost - a.bit_cost - b.bit_cost);
}

}  // namespace

namespace {

// A handful of helper threads for the clustering's independent Huffman-cost evaluations
// (integer arithmetic: the results do not depend on who computes them).  The helpers sleep
// on a condition variable between sessions and spin between the ~130 short parallel loops of
// one session, whose bodies are only a few microseconds long.  Helpers are optional: a loop
// is complete when all its items are done, whoever ran them (the caller takes part).
class ClusterPool {
 public:
  static ClusterPool& Get() {
    static ClusterPool pool;
    return pool;
  }
  // Exclusive use for one clustering; false if there are no helpers or another host thread
  // holds the pool (the caller then runs its loops serially).
  bool Open() {
    if (workers_.empty() || !session_mu_.try_lock()) return false;
    used_ = 0;
    cur_.store(nullptr, std::memory_order_relaxed);
    closed_.store(false, std::memory_order_release);
    {
      std::lock_guard<std::mutex> g(mu_);
      open_ = true;
      warm_ = false;
    }
    open_flag_.store(true, std::memory_order_release);
    cv_.notify_all();
    return true;
  }
  // Announces a session for about `start`: the helpers wake up then and spin until a session
  // opens or `deadline` passes, so that the session does not start with sleeping helpers
  // (a futex wake-up costs as much as half of a clustering).
  void Warm(std::chrono::steady_clock::time_point start, std::chrono::steady_clock::time_point deadline) {
    if (workers_.empty()) return;
    {
      std::lock_guard<std::mutex> g(mu_);
      if (open_) return;
      warm_ = true;
      warm_start_ = start;
      warm_deadline_ = deadline;
    }
    cv_.notify_all();
  }
  void Close() {
    open_flag_.store(false, std::memory_order_release);
    {
      std::lock_guard<std::mutex> g(mu_);
      open_ = false;
    }
    closed_.store(true, std::memory_order_release);
    // job slots are reused by the next session: no helper may still look at them
    while (spinning_.load(std::memory_order_acquire) != 0) Pause();
    session_mu_.unlock();
  }
  // fn(i) for i in [0, n).  Only between Open() and Close().
  void Run(size_t n, const std::function<void(size_t)>& fn) {
    if (used_ == kMaxJobs) {
      for (size_t i = 0; i < n; i++) fn(i);
      return;
    }
    Job& j = jobs_[used_++];
    j.fn = &fn;
    j.n = n;
    // (items are claimed a sixteenth of the loop at a time: with one shared counter increment per item the eight
    // participants of a 64-item loop spent more time passing its cache line around than evaluating)
    j.chunk = n >= 32 ? n / 16 : 1;
    j.next.store(0, std::memory_order_relaxed);
    j.done.store(0, std::memory_order_relaxed);
    cur_.store(&j, std::memory_order_release);
    Drain(&j);
    while (j.done.load(std::memory_order_acquire) != n) Pause();
  }

 private:
  struct Job {
    const std::function<void(size_t)>* fn = nullptr;
    size_t n = 0, chunk = 1;
    std::atomic<size_t> next{0}, done{0};
  };
  static constexpr size_t kMaxJobs = 256;

  ClusterPool() {
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n = hw >= 16 ? 7 : hw >= 4 ? hw / 2 - 1 : 0;
    for (unsigned i = 0; i < n; i++) workers_.emplace_back([this] { Work(); });
  }
  ~ClusterPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  static void Pause() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  static void Drain(Job* j) {
    size_t ran = 0;
    const size_t n = j->n, chunk = j->chunk;
    for (size_t i0; (i0 = j->next.fetch_add(chunk, std::memory_order_relaxed)) < n;) {
      const size_t i1 = std::min(n, i0 + chunk);
      for (size_t i = i0; i < i1; ++i) (*j->fn)(i);
      ran += i1 - i0;
    }
    if (ran) j->done.fetch_add(ran, std::memory_order_release);
  }
  void Work() {
    for (;;) {
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [this] { return open_ || quit_ || warm_; });
        if (quit_) return;
        if (!open_) {
          // announced session: sleep until its expected start, then spin (unlocked) for it
          const auto start = warm_start_, deadline = warm_deadline_;
          if (cv_.wait_until(g, start, [this] { return open_ || quit_; })) {
            if (quit_) return;
          } else {
            g.unlock();
            while (!open_flag_.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < deadline) Pause();
            g.lock();
          }
          if (!open_) {
            if (std::chrono::steady_clock::now() >= warm_deadline_) warm_ = false;
            continue;
          }
        }
        spinning_.fetch_add(1, std::memory_order_relaxed);
      }
      Job* last = nullptr;
      while (!closed_.load(std::memory_order_acquire)) {
        Job* j = cur_.load(std::memory_order_acquire);
        if (j != nullptr && j != last) {
          Drain(j);
          last = j;
        } else {
          Pause();
        }
      }
      spinning_.fetch_sub(1, std::memory_order_release);
    }
  }
  std::vector<std::thread> workers_;
  std::mutex session_mu_, mu_;
  std::condition_variable cv_;
  bool open_ = false, quit_ = false, warm_ = false;
  std::chrono::steady_clock::time_point warm_start_, warm_deadline_;
  std::atomic<bool> open_flag_{false};
  std::atomic<bool> closed_{true};
  std::atomic<int> spinning_{0};
  std::atomic<Job*> cur_{nullptr};
  Job jobs_[kMaxJobs];
  size_t used_ = 0;
};

}  // namespace

namespace {
thread_local bool t_clustering_shared = false;
}
bool TakeClusteringShared() {
  const bool was = t_clustering_shared;
  t_clustering_shared = false;
  return was;
}

void WarmCodeConstruction(double start_in_ms, double give_up_in_ms) {
  const auto now = std::chrono::steady_clock::now();
  ClusterPool::Get().Warm(now + std::chrono::microseconds(static_cast<long long>(start_in_ms * 1e3)),
                          now + std::chrono::microseconds(static_cast<long long>(give_up_in_ms * 1e3)));
}

void ClusterHistograms(std::vector<Histogram>* histograms, std::vector<uint8_t>* context_map) {
  if (histograms->size() <= 1) return;  // enc_cluster.cc:121
  const size_t max_histograms = std::min<size_t>(8, histograms->size());
  std::vector<Histogram> in(*histograms);
  std::vector<Histogram>& out = *histograms;
  out.clear();
  out.reserve(max_histograms);
  std::vector<uint32_t> symbols(in.size(), static_cast<uint32_t>(max_histograms));
  std::vector<float> dists(in.size(), std::numeric_limits<float>::max());
  // The independent cost evaluations of each step run on the helper pool when it is free;
  // selections and ties are then resolved serially in the reference's order.
  ClusterPool& pool = ClusterPool::Get();
  // Whether sharing pays depends on the histograms: a Huffman cost over a dozen symbols takes a fifth of a
  // microsecond where the histogram is cached, and several times that on a core that has to fetch it first, so
  // few, small evaluations are done faster alone.  (16384^2 bench frame, EPYC 9575F, 7 helpers, inside complete
  // encodes, tools/code_probe.sh -- AC, 64 histograms with 379 non-zero counts: shared 0.18-0.19 ms, alone 0.09;
  // AC at distance 0.5, 607 counts: 0.25 against 0.23; DC, 45 histograms, 830 counts: 0.31 against 0.33; DC at
  // distance 0.5, 1029 counts: 0.29 against 0.40.  Until the depth-limited costs lost their sorts -- HuffmanBitCost
  // -- alone was twice as slow and the line lay at 450 counts.)  A fixed rule separates these cases -- shared from
  // 900 non-zero counts on -- and makes the latency of a code construction a function of its input alone (until
  // round 3 the choice was learnt per calling thread, with an exploratory call every 32nd time: the results never
  // depended on it, the latency did).
  size_t nonzero = 0;
  for (const Histogram& h : in)
    for (size_t i = 0; i < kAlphabetSize; ++i) nonzero += h.counts[i] != 0;
  static const int forced = [] {
    const char* e = getenv("JXLT_POOL_MODE");  // (experiment knob, tools/code_probe.sh: 0 = never, 1 = always)
    return e ? atoi(e) : -1;
  }();
  const int mode = forced >= 0 ? (forced != 0) : (nonzero >= 900 ? 1 : 0);
  static const bool trace = getenv("JXLT_TRACE") != nullptr;
  if (trace)
    fprintf(stderr, "jxlt trace: clustering %zu histograms, %zu non-zero counts, %s\n", in.size(), nonzero,
            mode ? "shared" : "alone");
  const bool pooled = mode == 1 && in.size() >= 16 && pool.Open();
  // (only the loops over all histograms are handed out -- the bit costs, the selection rounds; a loop of a handful of
  // evaluations -- the assignment phase compares one histogram with <= 8 clusters -- costs less than the hand-over.
  // tools/code_bench.py, EPYC 9575F, shared: DC at d = 0.5 0.22 -> 0.19 ms, AC at d = 0.1 0.34 -> 0.24; alone 0.21 / 0.35)
  const size_t min_shared = 32;
  auto parallel_for = [&](size_t n, const std::function<void(size_t)>& fn) {
    if (pooled && n >= min_shared) {
      pool.Run(n, fn);
    } else {
      for (size_t i = 0; i < n; i++) fn(i);
    }
  };
  size_t largest = 0;
  parallel_for(in.size(), [&](size_t i) {  // enc_cluster.cc:48-58
    if (in[i].total_count == 0) {
      symbols[i] = 0;
      dists[i] = 0.0f;
    } else {
      ComputeBitCost(&in[i]);
    }
  });
  for (size_t i = 0; i < in.size(); i++) {
    if (in[i].total_count != 0 && in[i].total_count > in[largest].total_count) largest = i;
  }
  constexpr float kMinDistanceForDistinct = 64.0f;
  float cand[8];
  size_t cand_cost[8];
  if (pooled) {
    // (with the helper threads: every evaluation of a round at once, as the reference orders them)
    while (out.size() < max_histograms) {  // enc_cluster.cc:61-73
      symbols[largest] = static_cast<uint32_t>(out.size());
      out.push_back(in[largest]);
      dists[largest] = 0.0f;
      const Histogram& newest = out.back();
      parallel_for(in.size(), [&](size_t i) {
        if (dists[i] != 0.0f) dists[i] = std::min(Distance(in[i], newest), dists[i]);
      });
      // (every dists[largest] the reference compares against is already updated: largest < i)
      largest = 0;
      for (size_t i = 0; i < in.size(); i++) {
        if (dists[i] == 0.0f) continue;
        if (dists[i] > dists[largest]) largest = i;
      }
      if (dists[largest] < kMinDistanceForDistinct) break;
    }
    for (size_t i = 0; i < in.size(); i++) {  // enc_cluster.cc:75-90
      if (symbols[i] != max_histograms) continue;
      parallel_for(out.size(), [&](size_t j) { cand[j] = Distance(in[i], out[j], &cand_cost[j]); });
      size_t best = 0;
      float best_dist = cand[0];
      for (size_t j = 1; j < out.size(); j++) {
        if (cand[j] < best_dist) {
          best = j;
          best_dist = cand[j];
        }
      }
      out[best].AddHistogram(in[i]);
      out[best].bit_cost = cand_cost[best];  // (what ComputeBitCost(&out[best]) would find again)
      symbols[i] = static_cast<uint32_t>(best);
    }
  } else {
    // Alone: the same selections with fewer evaluations (round 4).  What the selection rounds need of a histogram's
    // distance to the clusters chosen so far -- dists[i], the minimum over them -- is only WHO has the largest, and
    // a minimum can only fall when a cluster is added: the value left from earlier rounds is an upper bound, and a
    // histogram whose bound is below another's exact value cannot be the round's choice, whatever its distance to the
    // new cluster is.  So a round evaluates the histogram with the largest bound (the first of them: the reference's
    // `>` keeps the lowest index among equals) against the clusters it has not met yet, until the largest bound is
    // an exact value -- that is the reference's largest_idx, with the reference's dists[largest_idx].  (A histogram
    // whose true minimum has reached 0 and is skipped by the reference from then on simply meets that cluster later
    // here, or never: 0 is the lowest value there is.)  The assignment phase then takes every distance to a cluster
    // that has not grown yet from what the rounds have evaluated (cluster j still is histogram `seed[j]` alone).
    // 16384^2 bench frame: DC code 529 -> 2xx evaluations, AC code 495 -> 2xx (tools/code_bench.py).
    const size_t n = in.size();
    std::vector<uint8_t> met(n, 0);                 // clusters [0, met[i]) have been evaluated against histogram i
    std::vector<size_t> pair_cost(n * 8, 0);         // [i * 8 + j]: bit cost of in[i] + the cluster's seed histogram
    std::vector<float> pair_dist(n * 8, 0.0f);
    size_t seed_of[8] = {};
    auto meet = [&](size_t i, size_t j) {            // Distance(in[i], out[j]) while out[j] is its seed alone
      size_t cost = 0;
      const float d = Distance(in[i], in[seed_of[j]], &cost);
      pair_cost[i * 8 + j] = cost;
      pair_dist[i * 8 + j] = d;
      return d;
    };
    while (out.size() < max_histograms) {  // enc_cluster.cc:61-73
      const size_t k = out.size();
      symbols[largest] = static_cast<uint32_t>(k);
      seed_of[k] = largest;
      out.push_back(in[largest]);
      dists[largest] = 0.0f;
      met[largest] = static_cast<uint8_t>(k + 1);
      for (;;) {
        largest = 0;
        for (size_t i = 0; i < n; i++) {
          if (dists[i] == 0.0f) continue;
          if (dists[i] > dists[largest]) largest = i;
        }
        if (dists[largest] == 0.0f || met[largest] == k + 1) break;  // exact: the round's choice
        // (one cluster at a time: the bound may fall below the next histogram's before all are met)
        const size_t j = met[largest]++;
        dists[largest] = std::min(meet(largest, j), dists[largest]);
      }
      if (dists[largest] < kMinDistanceForDistinct) break;
    }
    const size_t nclusters = out.size();
    bool grown[8] = {};
    for (size_t i = 0; i < n; i++) {  // enc_cluster.cc:75-90
      if (symbols[i] != max_histograms) continue;
      for (size_t j = 0; j < nclusters; j++) {
        if (!grown[j]) {
          if (j >= met[i]) meet(i, j);
          cand[j] = pair_dist[i * 8 + j];
          cand_cost[j] = pair_cost[i * 8 + j];
        } else {
          cand[j] = Distance(in[i], out[j], &cand_cost[j]);
        }
      }
      size_t best = 0;
      float best_dist = cand[0];
      for (size_t j = 1; j < nclusters; j++) {
        if (cand[j] < best_dist) {
          best = j;
          best_dist = cand[j];
        }
      }
      out[best].AddHistogram(in[i]);
      out[best].bit_cost = cand_cost[best];  // (what ComputeBitCost(&out[best]) would find again)
      grown[best] = true;
      symbols[i] = static_cast<uint32_t>(best);
    }
  }
  if (trace) fprintf(stderr, "jxlt trace: ... %zu Huffman-cost evaluations by this thread\n", t_cost_evaluations);
  t_cost_evaluations = 0;
  if (pooled) pool.Close();
  if (pooled) t_clustering_shared = true;
  // Canonical renumbering in order of first use (enc_cluster.cc:98-115).
  std::vector<Histogram> tmp(out);
  std::map<uint32_t, uint32_t> new_index;
  uint32_t next = 0;
  for (uint32_t s : symbols) {
    if (new_index.find(s) == new_index.end()) {
      new_index[s] = next;
      out[next] = tmp[s];
      ++next;
    }
  }
  out.resize(next);
  context_map->resize(symbols.size());
  for (size_t i = 0; i < symbols.size(); ++i) (*context_map)[i] = static_cast<uint8_t>(new_index[symbols[i]]);
}

// ---------------------------------------------------------------------------
// Code construction (enc_entropy_code.cc:455-514)
// ---------------------------------------------------------------------------
namespace {

void BuildHuffmanCodes(const std::vector<Histogram>& histograms, EntropyCode* code) {
  code->prefix_codes.assign(histograms.size(), PrefixCode{});
  for (size_t i = 0; i < histograms.size(); ++i) {
    PrefixCode& pc = code->prefix_codes[i];
    const uint32_t* counts = histograms[i].counts;
    size_t length = kAlphabetSize;
    while (length > 0 && counts[length - 1] == 0) --length;
    CreateHuffmanTree(counts, length, 15, pc.depths);
    ConvertBitDepthsToSymbols(pc.depths, length, pc.bits);
    size_t used = 0;
    for (size_t s = 0; s < length; ++s) used += counts[s] != 0;
    pc.single_symbol = used == 1 && !ReferenceSingleSymbolEmulation();
  }
}

}  // namespace

namespace {
std::atomic<bool> g_reference_single_symbol{false};
}
void SetReferenceSingleSymbolEmulation(bool on) { g_reference_single_symbol.store(on); }
bool ReferenceSingleSymbolEmulation() { return g_reference_single_symbol.load(); }

void OptimizeEntropyCode(const std::vector<Token>& tokens, size_t num_contexts, EntropyCode* code) {
  std::vector<Histogram> histograms(num_contexts);
  for (const Token& t : tokens) {
    uint32_t tok, nbits, bits;
    HybridUintEncode(t.value, &tok, &nbits, &bits);
    histograms[t.context].Add(tok);
  }
  code->orig_context_map.clear();
  code->context_map.clear();
  ClusterHistograms(&histograms, &code->context_map);
  if (code->context_map.empty()) code->context_map.assign(num_contexts, 0);
  BuildHuffmanCodes(histograms, code);
}

void OptimizeEntropyCode(std::vector<Histogram>* histograms, const uint8_t* static_map,
                         size_t num_static_contexts, EntropyCode* code) {
  const size_t num_hist = histograms->size();
  code->context_map.clear();
  ClusterHistograms(histograms, &code->context_map);
  if (code->context_map.empty()) code->context_map.assign(num_hist, 0);
  code->orig_context_map.assign(static_map, static_map + num_static_contexts);
  BuildHuffmanCodes(*histograms, code);
}

// ---------------------------------------------------------------------------
// Serialisation (enc_entropy_code.cc:18-453, 516-553)
// ---------------------------------------------------------------------------
namespace {

constexpr int kCodeLengthCodes = 18;

// The code-length sequence of a complex prefix code (what enc_entropy_code.cc:125-275 produces; the thresholds
// decide bytes, the structure is this file's own): the depths, trailing zeros dropped, are cut into RUNS of equal
// values; a run leaves either as that many literals or as one repeat code -- symbol 16 "previous non-zero length
// again" with 2 extra bits per code, symbol 17 "zeros" with 3 -- whose count is spread over as many codes as its
// digits need.
struct LengthSymbol {
  uint8_t symbol;  // 0…15 a literal depth, 16 / 17 the repeat codes
  uint8_t extra;   // the repeat code's extra bits
};

struct RepeatCode {
  uint8_t symbol;
  int extra_bits;
  // A count of 3 + 2^extra_bits is the first that needs two repeat codes; one literal + one code is cheaper.
  size_t FirstTwoCodeCount() const { return 3 + (size_t{1} << extra_bits); }
};
constexpr RepeatCode kRepeatPrevious = {16, 2};
constexpr RepeatCode kRepeatZeros = {17, 3};

// Calls f(value, reps, first index) for the maximal runs of depth[0, n).
template <typename F>
void ForEachRun(const uint8_t* depth, size_t n, F f) {
  for (size_t i = 0, end; i < n; i = end) {
    for (end = i + 1; end < n && depth[end] == depth[i]; ++end) {
    }
    f(depth[i], end - i, i);
  }
}

// `count` further copies of `literal`: literals while they are cheaper, otherwise the repeat code.  A chain of k
// repeat codes with extras e_1 … e_k (most significant first) stands for 3 + e_k + Σ_{j<k} (e_j + 1)·B^(k-j),
// B = 2^extra_bits -- each code before the last scales what follows, hence the "minus one" per digit.
void AppendCopies(uint8_t literal, size_t count, const RepeatCode& rc, std::vector<LengthSymbol>* out) {
  if (count == rc.FirstTwoCodeCount()) {
    out->push_back({literal, 0});
    --count;
  }
  if (count < 3) {
    out->insert(out->end(), count, LengthSymbol{literal, 0});
    return;
  }
  const size_t mask = (size_t{1} << rc.extra_bits) - 1;
  uint8_t digits[24];
  int num = 0;
  for (size_t rest = count - 3;; rest = (rest >> rc.extra_bits) - 1) {
    digits[num++] = static_cast<uint8_t>(rest & mask);
    if ((rest >> rc.extra_bits) == 0) break;
  }
  while (num > 0) out->push_back({rc.symbol, digits[--num]});
}

std::vector<LengthSymbol> CodeLengthSequence(const uint8_t* depth, size_t length) {
  size_t used = length;
  while (used > 0 && depth[used - 1] == 0) --used;
  // Whether repeat codes are used at all, per class (zeros / non-zeros): only alphabets above 50 symbols, and only
  // when the runs long enough to gain (zeros from 3, others from 4) cover more than twice their number + 1.
  bool repeat_zeros = false, repeat_others = false;
  if (length > 50) {
    size_t covered[2] = {0, 0}, runs[2] = {1, 1};
    ForEachRun(depth, used, [&](uint8_t value, size_t reps, size_t) {
      const int cls = value != 0;
      if (reps >= size_t(3 + cls)) {
        covered[cls] += reps;
        ++runs[cls];
      }
    });
    repeat_zeros = covered[0] > 2 * runs[0];
    repeat_others = covered[1] > 2 * runs[1];
  }
  std::vector<LengthSymbol> out;
  uint8_t previous = 8;  // the format's initial "previous non-zero length"
  ForEachRun(depth, used, [&](uint8_t value, size_t reps, size_t) {
    if (value == 0) {
      if (repeat_zeros) AppendCopies(0, reps, kRepeatZeros, &out);
      else out.insert(out.end(), reps, LengthSymbol{0, 0});
      return;
    }
    if (!repeat_others) {
      out.insert(out.end(), reps, LengthSymbol{value, 0});
    } else {
      // Symbol 16 repeats the PREVIOUS non-zero length: a new value goes out once as a literal first.
      if (previous != value) {
        out.push_back({value, 0});
        --reps;
      }
      AppendCopies(value, reps, kRepeatPrevious, &out);
    }
    previous = value;
  });
  return out;
}

// enc_entropy_code.cc:326-375 + :22-66 + :68-87
void StoreComplexPrefixCode(const uint8_t* depths, size_t num, jxl::BitWriter* writer) {
  const std::vector<LengthSymbol> tree = CodeLengthSequence(depths, num);
  uint32_t histogram[kCodeLengthCodes] = {0};
  for (const LengthSymbol& s : tree) ++histogram[s.symbol];
  int num_codes = 0, single_code = 0;
  for (int i = 0; i < kCodeLengthCodes; ++i) {
    if (histogram[i]) {
      if (num_codes == 0) {
        single_code = i;
        num_codes = 1;
      } else if (num_codes == 1) {
        num_codes = 2;
        break;
      }
    }
  }
  uint8_t cl_depth[kCodeLengthCodes] = {0};
  uint16_t cl_bits[kCodeLengthCodes] = {0};
  CreateHuffmanTree(histogram, kCodeLengthCodes, 5, cl_depth);
  ConvertBitDepthsToSymbols(cl_depth, kCodeLengthCodes, cl_bits);

  // Code-length-code lengths in the fixed storage order with a fixed code.
  static const uint8_t kStorageOrder[kCodeLengthCodes] = {1, 2, 3, 4, 0, 5, 17, 6, 16,
                                                          7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kLenSymbols[6] = {0, 7, 3, 2, 1, 15};
  static const uint8_t kLenBits[6] = {2, 4, 3, 2, 2, 4};
  size_t codes_to_store = kCodeLengthCodes;
  if (num_codes > 1) {
    for (; codes_to_store > 0; --codes_to_store) {
      if (cl_depth[kStorageOrder[codes_to_store - 1]] != 0) break;
    }
  }
  size_t skip_some = 0;
  if (cl_depth[kStorageOrder[0]] == 0 && cl_depth[kStorageOrder[1]] == 0) {
    skip_some = 2;
    if (cl_depth[kStorageOrder[2]] == 0) skip_some = 3;
  }
  writer->Write(2, skip_some);
  for (size_t i = skip_some; i < codes_to_store; ++i) {
    const size_t l = cl_depth[kStorageOrder[i]];
    writer->Write(kLenBits[l], kLenSymbols[l]);
  }
  if (num_codes == 1) cl_depth[single_code] = 0;
  for (const LengthSymbol& s : tree) {
    writer->Write(cl_depth[s.symbol], cl_bits[s.symbol]);
    if (s.symbol == kRepeatPrevious.symbol) writer->Write(kRepeatPrevious.extra_bits, s.extra);
    if (s.symbol == kRepeatZeros.symbol) writer->Write(kRepeatZeros.extra_bits, s.extra);
  }
}

// enc_entropy_code.cc:89-123
void StoreSimplePrefixCode(const uint8_t* depths, size_t symbols[4], size_t num_symbols,
                           size_t max_bits, jxl::BitWriter* writer) {
  writer->Write(2, 1);
  writer->Write(2, num_symbols - 1);
  for (size_t i = 0; i < num_symbols; i++) {
    for (size_t j = i + 1; j < num_symbols; j++) {
      if (depths[symbols[j]] < depths[symbols[i]]) std::swap(symbols[j], symbols[i]);
    }
  }
  for (size_t i = 0; i < num_symbols; ++i) writer->Write(max_bits, symbols[i]);
  if (num_symbols == 4) writer->Write(1, depths[symbols[0]] == 1 ? 1 : 0);
}

void StoreVarLenUint16(size_t n, jxl::BitWriter* writer) {  // :377-387
  if (n == 0) {
    writer->Write(1, 0);
  } else {
    writer->Write(1, 1);
    const size_t nbits = 63 ^ static_cast<size_t>(__builtin_clzll(n));
    writer->Write(4, nbits);
    writer->Write(nbits, n - (1ULL << nbits));
  }
}

void WritePrefixCode(const PrefixCode& code, jxl::BitWriter* writer) {  // :389-423
  size_t count = 0, s4[4] = {0}, length = 0;
  for (size_t i = 0; i < kAlphabetSize; i++) {
    if (code.depths[i]) {
      if (count < 4) s4[count] = i;
      count++;
      length = i + 1;
    }
  }
  size_t max_bits_counter = length - 1, max_bits = 0;
  while (max_bits_counter) {
    max_bits_counter >>= 1;
    ++max_bits;
  }
  if (count <= 1) {
    writer->Write(4, 1);
    writer->Write(max_bits, s4[0]);
    return;
  }
  if (count <= 4) StoreSimplePrefixCode(code.depths, s4, count, max_bits, writer);
  else StoreComplexPrefixCode(code.depths, length, writer);
}

size_t NumSymbols(const PrefixCode& pc) {
  size_t n = 1;
  for (size_t i = 0; i < kAlphabetSize; i++)
    if (pc.depths[i]) n = i + 1;
  return n;
}

void WritePrefixCodes(const std::vector<PrefixCode>& codes, jxl::BitWriter* writer) {  // :425-453
  writer->Write(1, 1);  // use_prefix_code
  for (size_t i = 0; i < codes.size(); ++i) {
    writer->Write(4, 4);  // split_exponent
    writer->Write(3, 2);  // msb_in_token
    writer->Write(2, 0);  // lsb_in_token
  }
  for (const PrefixCode& pc : codes) StoreVarLenUint16(NumSymbols(pc) - 1, writer);
  for (const PrefixCode& pc : codes) {
    if (NumSymbols(pc) > 1) WritePrefixCode(pc, writer);
  }
}

void WriteMapEntries(const std::vector<uint8_t>& entries, jxl::BitWriter* writer) {
  // enc_entropy_code.cc:516-548 after the "all zero" early-out.
  writer->Write(3, 0);  // no simple code, no MTF, no LZ77
  EntropyCode map_code;
  map_code.context_map.assign(1, 0);
  std::vector<Histogram> h(1);
  for (uint8_t e : entries) {
    uint32_t tok, nbits, bits;
    HybridUintEncode(e, &tok, &nbits, &bits);
    h[0].Add(tok);
  }
  BuildHuffmanCodes(h, &map_code);
  WritePrefixCodes(map_code.prefix_codes, writer);
  for (uint8_t e : entries) WriteToken(0, e, map_code, writer);
}

}  // namespace

void WriteContextMap(const EntropyCode& code, jxl::BitWriter* writer) {
  const size_t num_contexts =
      code.orig_context_map.empty() ? code.context_map.size() : code.orig_context_map.size();
  if (num_contexts == 0) return;
  if (*std::max_element(code.context_map.begin(), code.context_map.end()) == 0) {
    writer->Write(3, 1);  // simple code, 0 bits per entry
    return;
  }
  std::vector<uint8_t> entries;
  if (!code.orig_context_map.empty()) {
    for (uint8_t o : code.orig_context_map) entries.push_back(code.context_map[o]);
  } else {
    entries = code.context_map;
  }
  WriteMapEntries(entries, writer);
}

void WriteStaticContextMap(const uint8_t* map, size_t n, jxl::BitWriter* writer) {
  EntropyCode code;
  code.context_map.assign(map, map + n);
  WriteContextMap(code, writer);
}

void WriteEntropyCode(const EntropyCode& code, jxl::BitWriter* writer) {
  WriteContextMap(code, writer);
  WritePrefixCodes(code.prefix_codes, writer);
}

}  // namespace jxlt
