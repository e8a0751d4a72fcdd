// TEST INFRASTRUCTURE -- a tiny HIP *execution model* on the CPU.
//
// Lets tests compile libjxl-tiny_amd/csrc/jxlt_device.h (pure device code)
// with g++ and run its kernels workgroup by workgroup, so every kernel can be
// compared bit-for-bit with the oracle in a container that has no GPU.  It is
// never linked into the product libraries (those are built by hipcc only).
//
// Model: one workgroup at a time; every HIP thread is a ucontext fiber.
//  * __syncthreads(): block barrier (all unfinished fibers must arrive;
//    anything else is reported as a divergent-barrier deadlock).
//  * __shfl/__shfl_xor/__shfl_up: point-to-point exchange keyed by a per-lane
//    call counter, so lanes that communicate must execute the same shuffle
//    sequence (as on hardware) but unrelated lane groups may diverge.
//  * __ballot: wave-wide rendezvous of all unfinished lanes of the wave.
#ifndef HIPSIM_HIP_RUNTIME_H_
#define HIPSIM_HIP_RUNTIME_H_

#include <math.h>
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 {
  float x, y;
} __attribute__((aligned(8)));
struct float4 {
  float x, y, z, w;
} __attribute__((aligned(16)));
struct uint2 {
  uint32_t x, y;
} __attribute__((aligned(8)));
struct uint4 {
  uint32_t x, y, z, w;
} __attribute__((aligned(16)));

namespace hipsim {

constexpr int kWave = 64;
constexpr int kRing = 256;

enum State { kRunnable, kAtBlockBarrier, kAtWaveBarrier, kWaitLane, kDone };

struct Fiber {
  ucontext_t ctx;    // entered once per workgroup (makecontext / setcontext); every later switch is a
  jmp_buf jb;        // _setjmp / _longjmp pair -- swapcontext saves the signal mask with a system call per switch,
  bool started = false;  // which was a third of the simulator's run time
  std::vector<char> stack;
  State state = kRunnable;
  dim3 tid;
  // Point-to-point exchanges are keyed by a per-lane count of exchanges, kept per scope: [0] the
  // general shuffles / DPP row moves (octets stay in step), [1] the quad-wide DPP broadcasts
  // (only the four lanes of a quad are known to execute them together).
  uint64_t shfl_epoch[2] = {0, 0};
  uint64_t ring[2][kRing];
  int wait_lane = -1;        // absolute thread index we are waiting for
  uint64_t wait_epoch = 0;
  int wait_scope = 0;
  uint64_t ballot_arg = 0;
  uint32_t wave_xchg[2] = {0, 0};  // payload of the whole-wave exchanges (two slots: a lane is at most one ahead)
  uint32_t wave_xchg_count = 0;
};

struct Machine {
  std::vector<Fiber> fibers;
  ucontext_t sched;
  jmp_buf sched_jb;
  int current = -1;
  dim3 block_idx, block_dim, grid_dim;
  std::function<void()> body;
  std::vector<uint64_t> ballot_result;  // per wave
};

inline Machine& M() {
  static Machine m;
  return m;
}
inline Fiber& cur() { return M().fibers[M().current]; }
inline void yield_to_scheduler() {
  if (_setjmp(cur().jb) == 0) _longjmp(M().sched_jb, 1);
}
inline void resume(Fiber& f) {
  if (_setjmp(M().sched_jb) == 0) {
    if (!f.started) {
      f.started = true;
      setcontext(&f.ctx);
    } else {
      _longjmp(f.jb, 1);
    }
  }
}

inline void trampoline() {
  M().body();
  cur().state = kDone;
  yield_to_scheduler();
}

inline void run_block(const std::function<void()>& body, dim3 grid, dim3 block, dim3 bidx) {
  Machine& m = M();
  const int n = (int)(block.x * block.y * block.z);
  m.body = body;
  m.block_idx = bidx;
  m.block_dim = block;
  m.grid_dim = grid;
  if ((int)m.fibers.size() != n) m.fibers.assign(n, Fiber());
  m.ballot_result.assign((n + kWave - 1) / kWave, 0);
  for (int i = 0; i < n; i++) {
    Fiber& f = m.fibers[i];
    if (f.stack.empty()) f.stack.resize(128 * 1024);
    f.state = kRunnable;
    f.started = false;
    f.shfl_epoch[0] = f.shfl_epoch[1] = 0;
    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &m.sched;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  for (;;) {
    bool progress = false;
    int done = 0;
    for (int i = 0; i < n; i++) {
      Fiber& f = m.fibers[i];
      if (f.state == kWaitLane) {
        const Fiber& src = m.fibers[f.wait_lane];
        if (src.shfl_epoch[f.wait_scope] >= f.wait_epoch) f.state = kRunnable;
        else if (src.state == kDone) {
          fprintf(stderr, "hipsim: thread %d shuffles from finished thread %d\n", i, f.wait_lane);
          abort();
        }
      }
      if (f.state == kRunnable) {
        m.current = i;
        resume(f);
        progress = true;
      }
      if (f.state == kDone) done++;
    }
    if (done == n) break;
    // wave barriers (ballot)
    const int nw = (n + kWave - 1) / kWave;
    for (int w = 0; w < nw; w++) {
      int waiting = 0, live = 0;
      uint64_t mask = 0;
      for (int l = 0; l < kWave && w * kWave + l < n; l++) {
        const Fiber& f = m.fibers[w * kWave + l];
        if (f.state == kDone) continue;
        live++;
        if (f.state == kAtWaveBarrier) {
          waiting++;
          if (f.ballot_arg) mask |= 1ull << l;
        }
      }
      if (live > 0 && waiting == live) {
        m.ballot_result[w] = mask;
        for (int l = 0; l < kWave && w * kWave + l < n; l++) {
          Fiber& f = m.fibers[w * kWave + l];
          if (f.state == kAtWaveBarrier) f.state = kRunnable;
        }
        progress = true;
      }
    }
    // block barrier
    {
      int waiting = 0, live = 0;
      for (int i = 0; i < n; i++) {
        if (m.fibers[i].state == kDone) continue;
        live++;
        if (m.fibers[i].state == kAtBlockBarrier) waiting++;
      }
      if (live > 0 && waiting == live) {
        for (int i = 0; i < n; i++)
          if (m.fibers[i].state == kAtBlockBarrier) m.fibers[i].state = kRunnable;
        progress = true;
      }
    }
    if (!progress) {
      int nb = 0, nwv = 0, nl = 0;
      for (int i = 0; i < n; i++) {
        nb += m.fibers[i].state == kAtBlockBarrier;
        nwv += m.fibers[i].state == kAtWaveBarrier;
        nl += m.fibers[i].state == kWaitLane;
      }
      fprintf(stderr,
              "hipsim: DEADLOCK in block %u (divergent barrier/shuffle): %d at __syncthreads, "
              "%d at ballot, %d waiting for a shuffle partner, %d done\n",
              bidx.x, nb, nwv, nl, done);
      for (int i = 0; i < n; i++) {
        const Fiber& f = m.fibers[i];
        if (f.state == kWaitLane)
          fprintf(stderr, "  thread %d (epoch %llu) waits for lane %d (epoch %llu)\n", i,
                  (unsigned long long)f.wait_epoch, f.wait_lane,
                  (unsigned long long)m.fibers[f.wait_lane].shfl_epoch[f.wait_scope]);
      }
      abort();
    }
  }
}

// Launches `kernel(args...)` over the whole grid, one workgroup after another.
template <typename K, typename... Args>
void launch(K kernel, dim3 grid, dim3 block, Args... args) {
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++)
        run_block([&]() { kernel(args...); }, grid, block, dim3(bx, by, bz));
}

inline uint64_t exchange(uint64_t v, int src_lane_in_wave, int scope = 0) {
  Machine& m = M();
  Fiber& f = cur();
  const int me = m.current;
  const int wave_base = me - (me % kWave);
  int src = wave_base + src_lane_in_wave;
  const int n = (int)m.fibers.size();
  const uint64_t e = ++f.shfl_epoch[scope];
  f.ring[scope][e % kRing] = v;
  if (src < 0 || src >= n || src == me) return v;
  Fiber& s = m.fibers[src];
  if (s.shfl_epoch[scope] < e) {
    f.state = kWaitLane;
    f.wait_lane = src;
    f.wait_epoch = e;
    f.wait_scope = scope;
    yield_to_scheduler();
  }
  if (s.shfl_epoch[scope] - e >= (uint64_t)kRing) {
    fprintf(stderr, "hipsim: shuffle ring overflow\n");
    abort();
  }
  return s.ring[scope][e % kRing];
}

}  // namespace hipsim

#define threadIdx (hipsim::cur().tid)
#define blockIdx (hipsim::M().block_idx)
#define blockDim (hipsim::M().block_dim)
#define gridDim (hipsim::M().grid_dim)

// (one workgroup runs at a time and stores are not reordered: fences are no-ops)
inline void __threadfence_system() {}
inline void __threadfence() {}

inline void __syncthreads() {
  hipsim::cur().state = hipsim::kAtBlockBarrier;
  hipsim::yield_to_scheduler();
}

template <typename T>
inline T hipsim_shfl_bits(T v, int src_lane, int scope = 0) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  u = hipsim::exchange(u, src_lane, scope);
  T r;
  memcpy(&r, &u, sizeof(T));
  return r;
}
inline int hipsim_lane() { return hipsim::M().current % hipsim::kWave; }

template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  (void)width;
  return hipsim_shfl_bits(v, hipsim_lane() ^ mask);
}
template <typename T>
inline T __shfl(T v, int src_lane, int width = 64) {
  const int lane = hipsim_lane();
  const int src = (lane & ~(width - 1)) | (src_lane & (width - 1));
  return hipsim_shfl_bits(v, src);
}
template <typename T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {
  (void)width;
  const int lane = hipsim_lane();
  const int src = lane - (int)delta;
  // lanes below delta keep their own value, but must still take part in the exchange
  return hipsim_shfl_bits(v, src < 0 ? lane : src);
}
inline unsigned long long __ballot(int pred) {
  hipsim::Fiber& f = hipsim::cur();
  f.ballot_arg = pred ? 1 : 0;
  f.state = hipsim::kAtWaveBarrier;
  hipsim::yield_to_scheduler();
  return hipsim::M().ballot_result[hipsim::M().current / hipsim::kWave];
}

inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }

// DPP move (llvm.amdgcn.update.dpp): lane i reads `src` of the lane selected by dpp_ctrl; lanes
// whose row/bank is masked out, or whose source lane is outside the row, keep `old`
// (or get 0 with bound_ctrl).  Supported controls: quad_perm (0x00-0xFF), row_shl:n
// (0x101-0x10F, reads lane i+n), row_shr:n (0x111-0x11F, reads lane i-n).
inline int __builtin_amdgcn_update_dpp(int old, int src, int dpp_ctrl, int row_mask, int bank_mask,
                                       bool bound_ctrl) {
  const int lane = hipsim_lane();
  const int row = lane >> 4, in_row = lane & 15, bank = in_row >> 2;
  int src_lane = -1;
  if (dpp_ctrl >= 0 && dpp_ctrl <= 0xFF) {
    src_lane = (lane & ~3) | ((dpp_ctrl >> (2 * (lane & 3))) & 3);
  } else if (dpp_ctrl >= 0x101 && dpp_ctrl <= 0x10F) {
    const int t = in_row + (dpp_ctrl - 0x100);
    src_lane = t < 16 ? (lane & ~15) | t : -1;
  } else if (dpp_ctrl >= 0x111 && dpp_ctrl <= 0x11F) {
    const int t = in_row - (dpp_ctrl - 0x110);
    src_lane = t >= 0 ? (lane & ~15) | t : -1;
  } else if (dpp_ctrl == 0x138) {  // wave_shr:1 -- lane i reads lane i - 1 of the whole wave
    src_lane = lane >= 1 ? lane - 1 : -1;
  } else {
    fprintf(stderr, "hipsim: unsupported dpp_ctrl 0x%x\n", dpp_ctrl);
    abort();
  }
  // every lane publishes `src`; masked-out lanes read nobody (their nominal source lane may
  // belong to a diverged octet)
  const bool enabled = ((row_mask >> row) & 1) && ((bank_mask >> bank) & 1);
  const bool quad_broadcast = dpp_ctrl >= 0 && dpp_ctrl <= 0xFF && dpp_ctrl % 0x55 == 0;
  const int got = hipsim_shfl_bits(src, (src_lane < 0 || !enabled) ? lane : src_lane, quad_broadcast ? 1 : 0);
  if (!enabled) return old;
  if (src_lane < 0) return bound_ctrl ? 0 : old;
  return got;
}

// v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950): odd 16-lane rows of the first operand trade places with
// the even rows of the second (16), the upper half of the first with the lower half of the second (32); [0] is the
// first operand afterwards, [1] the second (checked on the hardware by tools/permlane_probe.hip).
struct hipsim_uint2 {
  unsigned v[2];
  unsigned operator[](int i) const { return v[i]; }
};
// (whole-wave rendezvous through the __ballot barrier: the rows that trade places belong to different octets,
// whose shuffle counters -- the key of the point-to-point exchange above -- may differ)
inline unsigned hipsim_wave_read(unsigned v, int src_lane) {
  hipsim::Fiber& f = hipsim::cur();
  const uint32_t slot = f.wave_xchg_count++ & 1u;
  f.wave_xchg[slot] = v;
  (void)__ballot(0);  // every unfinished lane of the wave has published its value
  const int me = hipsim::M().current;
  return hipsim::M().fibers[me - (me % hipsim::kWave) + src_lane].wave_xchg[slot];
}
inline hipsim_uint2 __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned src0, bool, bool) {
  const int lane = hipsim_lane();
  const bool odd = (lane >> 4) & 1;
  const unsigned a = hipsim_wave_read(src0, odd ? lane - 16 : lane);
  const unsigned b = hipsim_wave_read(vdst, odd ? lane : lane + 16);
  return {{odd ? a : vdst, odd ? src0 : b}};
}
inline hipsim_uint2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src0, bool, bool) {
  const int lane = hipsim_lane();
  const bool upper = lane >= 32;
  const unsigned a = hipsim_wave_read(src0, upper ? lane - 32 : lane);
  const unsigned b = hipsim_wave_read(vdst, upper ? lane : lane + 32);
  return {{upper ? a : vdst, upper ? src0 : b}};
}

// v_sqrt_f32: the hardware result is within 1 ulp of the root, not always correctly rounded.
// The model returns the correctly rounded root pushed one ulp up or down (pseudo-randomly,
// never for 0), so that code relying on more than the 1-ulp guarantee fails here.
inline float __builtin_amdgcn_sqrtf(float x) {
  float s = sqrtf(x);
  if (!(x > 0.0f) || !(s > 0.0f) || s != s || s > 3.0e38f) return s;
  uint32_t b;
  memcpy(&b, &s, 4);
  uint32_t h;
  memcpy(&h, &x, 4);
  h = (h ^ (h >> 15)) * 0x2c1b3c6du;
  h ^= h >> 12;
  const uint32_t pick = h % 3u;  // 0: exact, 1: one ulp up, 2: one ulp down
  if (pick == 1) b += 1;
  if (pick == 2) b -= 1;
  memcpy(&s, &b, 4);
  return s;
}

// v_readfirstlane: only used on values that are wave-uniform by construction.
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// v_readlane: every lane of the wave executes it (wave-uniform control flow) and gets lane l's value
inline unsigned hipsim_wave_read(unsigned v, int src_lane);
inline int __builtin_amdgcn_readlane(int v, int l) { return (int)hipsim_wave_read((unsigned)v, l); }

inline void __builtin_amdgcn_s_setprio(int) {}  // issue priority: no effect on results

inline void __builtin_amdgcn_sched_barrier(int) {}  // compiler scheduling fence
#define JXLT_TOUCH_VGPR(x) ((void)(x))
#define JXLT_LUT_WRAP(off) ((off) & (uint32_t)(kSqrtLutSize * 4 - 4))  // (see jxlt_device_common.h)
#define JXLT_OCTET_SUM_PORTABLE 1
#define JXLT_SQRT_PORTABLE 1  // (sqrt_exact_by_rsq: sqrtf)  // (octet_sum: the exchange steps instead of the inline assembly)
#define JXLT_LAUNDER_VGPR(x) ((void)(x))
#define JXLT_DEFINE_VGPR(x) ((x) = 0)
#define JXLT_LDS_STORE_ROW(row_base, off, val) ((row_base)[(off) / 4 + hipsim_lane()] = (val))
#define JXLT_LAUNDER_SGPR(x) ((void)(x))
#define JXLT_GLOBAL_POINTER_TYPES
typedef const char* JxltGlobalBytes;
typedef const float* JxltGlobalFloats;
typedef int16_t* JxltGlobalShorts;
typedef const int16_t* JxltGlobalConstShorts;
typedef const uint32_t* JxltGlobalConstWords;
#define JXLT_SCALAR_STORE64(p, i, v) ((p)[i] = (v))
#define JXLT_SCALAR_STORES_DONE() ((void)0)
#define JXLT_COMPILER_FENCE() ((void)0)
typedef float4 JxltFloat4;  // (see jxlt_device_common.h: issue order and waits have no meaning here)
#define JXLT_LDS_LOAD4_NOW(dst, p, byte_off) ((dst) = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p) + (byte_off)))
#define JXLT_LDS_ADDRESS(p) ((uintptr_t)(p))  // (no separate LDS address space here: the pointer itself)
#define JXLT_LDS_LOAD4_NOW_AT(dst, addr, byte_off) ((dst) = *reinterpret_cast<const float4*>((addr) + (byte_off)))
#define JXLT_LDS_WAIT4(n, a, b, c, d) ((void)0)
#define JXLT_LDS_DRAIN4(a, b, c, d) ((void)0)
#define JXLT_STORES_WRITTEN() ((void)0)

// v_rcp_f32 (1 ulp on hardware; the model returns the correctly rounded reciprocal)
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline uint32_t __builtin_amdgcn_ubfe(uint32_t v, uint32_t offset, uint32_t width) {  // v_bfe_u32
  return (v >> offset) & ((width >= 32) ? 0xFFFFFFFFu : ((1u << width) - 1u));
}

// v_alignbyte_b32: ({hi, lo} >> (8 * (shift & 3))) & 0xffffffff
inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t shift) {
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (shift & 3)));
}

// v_mbcnt_lo/hi: add the number of set mask bits below the calling lane
inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t add) {
  const int lane = hipsim_lane();
  const uint32_t below = lane >= 32 ? 0xFFFFFFFFu : ((1u << lane) - 1u);
  return add + (uint32_t)__builtin_popcount(mask & below);
}
inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t add) {
  const int lane = hipsim_lane();
  const uint32_t below = lane <= 32 ? 0u : (lane >= 64 ? 0xFFFFFFFFu : ((1u << (lane - 32)) - 1u));
  return add + (uint32_t)__builtin_popcount(mask & below);
}

template <typename T>
inline T atomicMax(T* p, T v) {
  const T old = *p;
  if (v > old) *p = v;
  return old;
}

inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline int32_t __float_as_int(float f) { int32_t u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline long long clock64() { return 0; }

// One workgroup runs at a time and fibers never preempt: plain RMW is atomic.
template <typename T>
inline T atomicOr(T* p, T v) {
  T old = *p;
  *p = old | v;
  return old;
}
template <typename T>
inline T atomicAdd(T* p, T v) {
  T old = *p;
  *p = old + v;
  return old;
}

// The product's octet transposes rely on a wave's LDS operations executing in program order
// (jxlt_device.h: JXLT_OCTET_SYNC is a compiler fence there).  Here lanes are fibers, so "in order
// within the wave" has to be made explicit: a butterfly of dummy exchanges synchronises exactly the
// eight lanes of the octet (octets may have diverged, a wave-wide barrier would deadlock the model).
#define JXLT_OCTET_SYNC()     \
  do {                        \
    (void)__shfl_xor(0, 1);   \
    (void)__shfl_xor(0, 2);   \
    (void)__shfl_xor(0, 4);   \
  } while (0)

// A whole wave: every unfinished lane of the wave has to arrive (a ballot is such a rendezvous).
#define JXLT_WAVE_SYNC() ((void)__ballot(0))

#endif  // HIPSIM_HIP_RUNTIME_H_
