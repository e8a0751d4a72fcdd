/* jxl_tiny_amd_testing.h -- entry points of libjxltiny_hip.so / libjxltiny_host.so that exist for the
 * test-suite and the profiling tools only.  Nothing a libjxl-tiny maintainer binds is declared here: the drop-in
 * surface is include/jxl_tiny_amd.h. */
#ifndef JXL_TINY_AMD_TESTING_H_
#define JXL_TINY_AMD_TESTING_H_

#include "jxl_tiny_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- libjxltiny_hip.so ------------------------------------------------- */

/* Debug intermediates of the last encode run with JXLT_FLAG_DEBUG_DUMP.
 * what: 0,1,2 = XYB planes f32 (xsize_blocks*8 x ysize_blocks*8);
 *       3 = quant field f32 per block; 4 = masking f32 per block;
 *       5 = entropy estimates f32, 8 per 2x2-block cell, grid
 *           (xsize_blocks/2+1) x (ysize_blocks/2+1).
 *       6 = u64[16] shader cycles per tile_kernel phase, summed over tiles (needs
 *           JXLT_FLAG_PROFILE instead of JXLT_FLAG_DEBUG_DUMP).
 *       7 = u32: encodes of this context in which tiles were redone with computed roots (jxlt_encode_stats;
 *           always allowed). */
int jxlt_debug_fetch(jxlt_context* ctx, int what, void* host_dst, size_t bytes);

/* The self-check jxlt_context_create runs once per device: the bit pattern of 3.0f x 2^-147 as the DEVICE computes it.
 * 12 when FP32 denormals are kept, which tile12_kernel's table offsets rely on (csrc/jxlt_tile_kernel.h); 0 when the
 * device code flushes them (a build without -fno-gpu-flush-denormals-to-zero on a target that defaults to flushing). */
int jxlt_debug_denormal_probe(jxlt_context* ctx, uint32_t* bits);

/* Timing of the device stages of the last jxlt_encode_enqueue (HIP events on the context's
 * stream): writes up to `cap` entries; returns the number of stages, or < 0.
 * Waits for the device pipeline of that encode.  Not for a batch lane's frames (a context with
 * jxlt_context_set_wait_mode(ctx, 1), resident frame of up to 1024 groups): those carry no stage
 * events -- every event is a packet the device has to work through -- and the call fails. */
typedef struct {
  const char* name;
  float milliseconds;
} jxlt_kernel_time;
int jxlt_kernel_times(jxlt_context* ctx, jxlt_kernel_time* out, int cap);


/* The side-band grids and the AC symbol histograms of the last encode without its raw tokens (out->tokens is NULL):
 * what a reference-style caller that packs on the device would fetch.  The test-suite compares the grids. */
int jxlt_fetch_side_info(jxlt_context* ctx, jxlt_frame_result* out, const uint32_t** ac_histograms);
/* jxlt_pack_begin + jxlt_pack_sizes + jxlt_pack_deliver into a page-locked buffer of the context + jxlt_synchronize:
 * the byte-aligned sections of one kind, concatenated in section order, with their bytes (out->bytes), for
 * comparisons with the host's and the oracle's bit packing. */
int jxlt_pack_sections(jxlt_context* ctx, int kind, const uint32_t* code_table, jxlt_packed_sections* out);

/* ---- libjxltiny_host.so ------------------------------------------------ */

/* jxlt_assemble_frame with separately allocated per-group token buffers. */
int jxlt_assemble_frame_groups(const jxlt_frame_result* frame, const uint8_t* const* group_tokens,
                               const size_t* group_token_bytes, const jxlt_distance_params* distp,
                               int num_threads, uint8_t** out_bytes, size_t* out_size);


/* Building blocks of the sharded frame, for stage-by-stage comparisons with the oracle: code tables from summed
 * histograms, and the complete codestream from summed histograms + all packed sections in raster order. */
int jxlt_build_code_tables(const uint32_t* ac_histograms, const uint32_t* dc_histograms,
                           uint32_t* ac_code_table, uint32_t* dc_code_table);
int jxlt_finish_frame(size_t xsize, size_t ysize, float distance, const uint32_t* ac_histograms,
                      const uint32_t* dc_histograms, const jxlt_packed_sections* dc_sections,
                      const jxlt_packed_sections* ac_sections, uint8_t** out_bytes, size_t* out_size);


/* The protocol of jxlt_shard_encode over caller-supplied slab operations instead of a device context (what jxlt_shard_encode
 * binds to the jxlt_* calls named on the right); lets the CPU test-suite run the protocol without a GPU.
 * Every callback returns JXLT_OK or an error; `write` may be asynchronous, `finish` completes it. */
typedef struct {
  void* self;
  int (*enqueue)(void* self, const jxlt_params* params);                     /* jxlt_encode_enqueue */
  int (*dc_histogram)(void* self, const uint32_t** histogram);               /* jxlt_fetch_dc_histogram */
  int (*begin_dc_pack)(void* self, const uint32_t* dc_code_table);           /* jxlt_pack_begin(0) */
  int (*ac_histogram)(void* self, const uint32_t** histogram);               /* jxlt_fetch_histograms */
  int (*measure)(void* self, const uint32_t* ac_code_table, jxlt_packed_sections* dc,
                 jxlt_packed_sections* ac);                                  /* jxlt_pack_begin(1) + jxlt_pack_sizes(0 / 1) */
  int (*write)(void* self, uint8_t* out, const jxlt_section_run* dc_runs, size_t num_dc_runs,
               const jxlt_section_run* ac_runs, size_t num_ac_runs);         /* jxlt_pack_deliver(0 / 1, out, runs) */
  int (*finish)(void* self);                                                 /* jxlt_synchronize */
} jxlt_slab_ops;
int jxlt_shard_encode_ops(jxlt_shard_group* group, const jxlt_slab_ops* ops, size_t xsize, size_t ysize,
                          float distance, const uint8_t** bytes, size_t* size);

/* Host-side stage times of the caller's last frame on this group, milliseconds from the start of its
 * jxlt_shard_encode* call: [0] device pipeline enqueued, [1] own DC histogram here, [2] own AC histogram here, [3] both
 * code tables here (the sum over the participants + the code construction on participants 0 and 1 lie in front),
 * [4] own section sizes here, [5] the layout here (participant 0: made), [6] own hand-over issued, [7] participant 0:
 * every participant has placed its sections.  What tools/slab_of_8.py splits a rank's serial stage with. */
int jxlt_shard_group_last_timeline(const jxlt_shard_group* group, double* ms8);

/* jxlt_shard_pipeline_* over slab operations: lane l of the pipeline runs its frames through lane_ops[l]. */
int jxlt_shard_pipeline_open_ops(const char* shm_name, int rank, int world, const jxlt_slab_ops* lane_ops, int depth,
                                 size_t output_capacity, size_t max_sections, jxlt_shard_pipeline** out);
int jxlt_shard_pipeline_submit_ops(jxlt_shard_pipeline* pipeline, size_t xsize, size_t ysize, float distance,
                                   uint64_t* ticket);

/* The raw 3-byte records of DC group `dc_group_index` exactly as the host
 * tokeniser (WriteDCGroup in raw-record form, enc_frame.cc:536-570) produces them. */
int jxlt_debug_dc_records(const jxlt_frame_result* frame, size_t dc_group_index, uint8_t** out_bytes,
                          size_t* out_size);

#ifdef __cplusplus
}
#endif
#endif /* JXL_TINY_AMD_TESTING_H_ */
