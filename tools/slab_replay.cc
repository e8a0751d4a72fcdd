// tools/slab_replay.cc -- test / measurement infrastructure (not on the product path).
// The seven REPLAYING participants of tools/slab_of_8.py as native threads (in round 5 they were
// Python threads whose callbacks took turns on the interpreter lock -- two of thirty steps took 3.4 and 64 ms and the
// 1.76-ms median was a Python-harness number too).  Each thread joins the shard group of the run as one rank and runs
// the product's exchange protocol (jxlt_shard_encode_ops, host/frame_shards.cc) on slab operations that hand back what
// a device context produced for that rank's rectangle beforehand: histograms, section sizes, section bytes.  They
// answer at once, so participant 0 -- the real device context, driven by the caller -- never waits for a slower peer.
//   g++ -O2 -shared -fPIC -I../include -o libslab_replay.so slab_replay.cc -L../libjxl-tiny_amd/host -ljxltiny_host -pthread
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "jxl_tiny_amd.h"
#include "jxl_tiny_amd_testing.h"

extern "C" {
// What a device context produced for one rank's rectangle (recorded with the product's kernels).
typedef struct {
  const uint32_t* dc_hist;  // [64 * 64]
  const uint32_t* ac_hist;  // [64 * 64]
  const uint8_t* bytes[2];  // packed sections of kind 0 (DC groups) / 1 (AC groups), back to back
  const uint64_t* off[2];   // [nsec + 1] byte offsets
  const uint32_t* bits[2];  // [nsec] bit sizes
  size_t nsec[2];
} slab_replay_record;
}

namespace {
struct Peer {
  slab_replay_record rec;
  jxlt_slab_ops ops;
  jxlt_shard_group* group = nullptr;
  std::thread thread;
  std::string error;
};
std::vector<Peer*> g_peers;

int OpEnqueue(void*, const jxlt_params*) { return JXLT_OK; }
int OpDcHistogram(void* self, const uint32_t** h) {
  *h = static_cast<Peer*>(self)->rec.dc_hist;
  return JXLT_OK;
}
int OpBeginDcPack(void*, const uint32_t*) { return JXLT_OK; }
int OpAcHistogram(void* self, const uint32_t** h) {
  *h = static_cast<Peer*>(self)->rec.ac_hist;
  return JXLT_OK;
}
int OpMeasure(void* self, const uint32_t*, jxlt_packed_sections* dc, jxlt_packed_sections* ac) {
  const slab_replay_record& r = static_cast<Peer*>(self)->rec;
  jxlt_packed_sections* out[2] = {dc, ac};
  for (int k = 0; k < 2; k++) {
    out[k]->bytes = nullptr;
    out[k]->section_offset = r.off[k];
    out[k]->section_bits = r.bits[k];
    out[k]->num_sections = r.nsec[k];
  }
  return JXLT_OK;
}
int OpWrite(void* self, uint8_t* out, const jxlt_section_run* dc_runs, size_t n_dc, const jxlt_section_run* ac_runs,
            size_t n_ac) {
  const slab_replay_record& r = static_cast<Peer*>(self)->rec;
  const jxlt_section_run* runs[2] = {dc_runs, ac_runs};
  const size_t n[2] = {n_dc, n_ac};
  for (int k = 0; k < 2; k++)
    for (size_t i = 0; i < n[k]; i++) {
      const jxlt_section_run& run = runs[k][i];
      const uint64_t lo = r.off[k][run.first_section], hi = r.off[k][run.first_section + run.num_sections];
      if (hi > lo) memcpy(out + run.dst_offset, r.bytes[k] + lo, hi - lo);
    }
  return JXLT_OK;
}
int OpFinish(void*) { return JXLT_OK; }
}  // namespace

extern "C" {

// Ranks first_rank .. world - 1 of the group `shm_name` (rank 0 has created it) as threads that run `frames` frames
// each.  records[r - first_rank]: rank r's recorded results (the arrays stay the caller's and must outlive the join).
int slab_replay_start(const char* shm_name, int world, int first_rank, size_t capacity, size_t max_sections,
                      const slab_replay_record* records, size_t xsize, size_t ysize, float distance, int frames) {
  for (int r = first_rank; r < world; r++) {
    Peer* p = new Peer;
    p->rec = records[r - first_rank];
    p->ops = {p, OpEnqueue, OpDcHistogram, OpBeginDcPack, OpAcHistogram, OpMeasure, OpWrite, OpFinish};
    if (jxlt_shard_group_open(shm_name, r, world, capacity, max_sections, &p->group) != JXLT_OK) {
      delete p;
      return -1;
    }
    g_peers.push_back(p);
  }
  for (Peer* p : g_peers)
    p->thread = std::thread([p, xsize, ysize, distance, frames] {
      for (int f = 0; f < frames; f++) {
        const uint8_t* bytes = nullptr;
        size_t size = 0;
        const int rc = jxlt_shard_encode_ops(p->group, &p->ops, xsize, ysize, distance, &bytes, &size);
        if (rc != JXLT_OK) {
          p->error = std::string("frame ") + std::to_string(f) + ": " + jxlt_shard_group_last_error(p->group);
          return;
        }
      }
    });
  return 0;
}

// Waits for the threads; the number of participants that failed (their messages go to stderr).
int slab_replay_join(void) {
  int failed = 0;
  for (Peer* p : g_peers) {
    if (p->thread.joinable()) p->thread.join();
    if (!p->error.empty()) {
      fprintf(stderr, "slab_replay: %s\n", p->error.c_str());
      failed++;
    }
    jxlt_shard_group_close(p->group);
    delete p;
  }
  g_peers.clear();
  return failed;
}

}  // extern "C"
