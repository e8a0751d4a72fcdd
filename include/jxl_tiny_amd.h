/* jxl_tiny_amd.h -- C ABI of the MI355X-native JPEG XL "tiny" encoder hot path.
 *
 * The reference (libjxl-tiny) has no plugin/FFI surface; its boundary is the
 * C++ call chain cjxl_tiny -> EncodeFile -> EncodeFrame.  This ABI is what the
 * replacement EncodeFrame binds instead of the serial per-stripe loop at
 *   /root/reference/encoder/enc_frame.cc:716-757   (ProcessDCGroup body:
 *     CopyAndPadImage, ToXYB, ProcessTile{ComputeAdaptiveQuantFieldTile,
 *     ComputeCmapTile, FindBest16x16Transform, AdjustQuantField}, WriteACGroup)
 * Plain pointers and sizes only; no C++ or torch types; no exceptions cross it.
 * Every function returns 0 on success and a negative JXLT_ERR_* otherwise;
 * jxlt_last_error() gives a human-readable message.
 *
 * Two libraries export these symbols:
 *   libjxltiny_hip.so   (hipcc, gfx950)  jxlt_context_* / jxlt_image_* /
 *                                        jxlt_encode_* / jxlt_fetch_* / jxlt_debug_*
 *   libjxltiny_host.so  (g++)            jxlt_assemble_frame, jxlt_encode_file_planar,
 *                                        jxlt_compute_distance_params
 */
#ifndef JXL_TINY_AMD_H_
#define JXL_TINY_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JXLT_OK 0
#define JXLT_ERR_INVALID_ARGUMENT (-1)
#define JXLT_ERR_NO_DEVICE (-2)   /* no usable HIP device / HIP runtime error */
#define JXLT_ERR_OUT_OF_MEMORY (-3)
#define JXLT_ERR_UNSUPPORTED (-4) /* image that fits one 8x8 block (reference traps); a frame with values the     \
                                     codestream cannot carry: a quantised coefficient whose token needs more than   \
                                     16 bits (reference: debug assert only, enc_bit_writer.cc:120) or a DC value    \
                                     beyond int16 -- reported by every call about that encode */
#define JXLT_ERR_INTERNAL (-5)

/* Scalars of the reference's DistanceParams (enc_frame.cc:104-156) that the
 * device pipeline consumes, plus mode flags. */
typedef struct {
  float distance;      /* butteraugli distance */
  float scale;         /* global_scale / 65536 */
  float inv_scale;     /* 1 / scale */
  float scale_dc;      /* quant_dc * scale */
  uint32_t x_qm_scale; /* 2..5 */
  uint32_t flags;      /* JXLT_FLAG_* */
} jxlt_params;

#define JXLT_FLAG_FORCE_DCT8 1u /* == OPTIMIZE_BLOCK_SIZES 0 (config.h:12) */
#define JXLT_FLAG_DEBUG_DUMP 2u /* keep XYB/quant-field/masking/entropy intermediates */
#define JXLT_FLAG_PROFILE 4u    /* record per-kernel HIP event timings */

/* Full DistanceParams for the host back-end (enc_frame.cc:104-156). */
typedef struct {
  float distance;
  int32_t global_scale;
  int32_t quant_dc;
  float scale;
  float inv_scale;
  float scale_dc;
  uint32_t x_qm_scale;
  uint32_t epf_iters;
} jxlt_distance_params;

/* Host-visible result of one pass of the hot path over the current image.
 * All pointers are owned by the context (pinned host memory) and stay valid
 * until the next jxlt_fetch_result()/jxlt_context_destroy() on that context.
 * Grids are image-absolute: blocks = 8x8 px (pitch xsize_blocks), tiles = 64x64
 * px (pitch xsize_tiles).  Mirrors DCGroupData (dc_group_data.h:19-37) and the
 * per-group raw token sections (enc_group.cc:468-470,483-485). */
typedef struct {
  size_t xsize, ysize;
  size_t xsize_blocks, ysize_blocks;
  size_t xsize_tiles, ysize_tiles;
  size_t num_groups;                 /* ceil(x/256) * ceil(y/256), raster order */
  const int16_t* quant_dc[3];        /* X, Y, B */
  const uint8_t* raw_quant_field;
  const uint8_t* ac_strategy;        /* (type << 1) | is_first_block */
  const int8_t* ytox_map;
  const int8_t* ytob_map;
  const uint8_t* tokens;             /* all groups, concatenated 3-byte records */
  const uint64_t* group_token_offset; /* [num_groups + 1], byte offsets into tokens */
} jxlt_frame_result;

typedef struct jxlt_context jxlt_context; /* one per (host thread, device) */

/* ---- libjxltiny_hip.so ------------------------------------------------- */

int jxlt_context_create(int device_ordinal, jxlt_context** ctx);
void jxlt_context_destroy(jxlt_context* ctx);
/* ctx may be NULL: returns the message of the last failed create. */
const char* jxlt_last_error(const jxlt_context* ctx);
/* The device ordinal the context was created for. */
int jxlt_context_device(const jxlt_context* ctx);
/* While a device has another living context of this library, jxlt_context_destroy keeps the destroyed context's
 * device buffers (blocks of 1 MB and more, at most 8 GB) for the next context instead of returning them to the HIP
 * runtime: on this stack memory that was freed and is handed out again makes kernels and downloads measurably slower
 * (DESIGN.md 3).  When the LAST context of a device is destroyed everything kept for that device is returned, so a
 * co-resident allocator (PyTorch, ...) sees the memory again.  JXLT_DEVICE_CACHE_MB=<n> (environment) opts in to
 * keeping up to n MB beyond the last context (0: never keep anything).  This call returns all kept blocks of
 * `device_ordinal` (-1: every device) to the runtime at once; the number of bytes released. */
size_t jxlt_release_cached_memory(int device_ordinal);

/* Number of usable HIP devices (0: none -- every other entry then fails with JXLT_ERR_NO_DEVICE). */
int jxlt_device_count(void);
/* Restricts the calling thread -- and the threads it creates afterwards, e.g. the helper threads of the code
 * construction -- to the CPUs next to the device (its PCI function's local_cpulist, i.e. the GPU's NUMA node), so
 * that the host side of an encode does not run on the other socket.  JXLT_ERR_UNSUPPORTED when the system does
 * not say which CPUs those are.  The library binds only threads it owns (pipeline lanes, the code construction's
 * workers); jxlt_shard_encode binds its caller for the duration of the call and restores the caller's mask before it
 * returns (JXLT_NO_AFFINITY=1 disables both).  An application that wants its encoding thread next to the GPU for
 * good calls this itself. */
int jxlt_bind_thread_near_device(int device_ordinal);

/* Copies three planar f32 linear-sRGB planes (row pitch in bytes, as
 * Image3F::bytes_per_row(), image.h:382) into context-owned HBM.  Replaces the
 * host reads of CopyAndPadImage (enc_frame.cc:597-617). */
int jxlt_image_upload(jxlt_context* ctx, const float* const planes[3], size_t pitch_bytes,
                      size_t xsize, size_t ysize);
/* Page-locked host memory (hipHostMalloc) for image planes: uploads from it run at PCIe
 * speed without staging.  Returns NULL when no HIP device is usable (callers fall back to
 * ordinary memory).  Free with jxlt_pinned_free. */
void* jxlt_pinned_alloc(size_t bytes);
void jxlt_pinned_free(void* p);
/* Page-locks memory the caller already owns (e.g. a shared-memory mapping that several processes assemble
 * one codestream in) so that the devices can copy from / to it directly.  Undo with jxlt_pinned_unregister
 * before the memory goes away. */
int jxlt_pinned_register(void* p, size_t bytes);
void jxlt_pinned_unregister(void* p);

/* Borrows planes already resident in HBM (e.g. a torch tensor's data_ptr). */
int jxlt_image_set_device(jxlt_context* ctx, const void* const device_planes[3],
                          size_t pitch_bytes, size_t xsize, size_t ysize);

/* PFM ingest on the device (read_pfm.cc:199-209 semantics without a host pass over the pixels):
 * the frame is the sample payload of a PFM file -- xsize * ysize interleaved RGB f32 triples,
 * bottom row first, byte-reversed when the file is big endian (positive scale line).  The
 * kernels read it in place.  _upload_pfm copies the payload from host memory (page-locked
 * memory from jxlt_pinned_alloc goes over PCIe in one piece); _set_device_pfm takes a payload
 * that already is in device memory (not copied, must stay valid until the encode is done). */
int jxlt_image_upload_pfm(jxlt_context* ctx, const void* host_payload, size_t xsize, size_t ysize,
                          int big_endian);
int jxlt_image_set_device_pfm(jxlt_context* ctx, const void* device_payload, size_t xsize, size_t ysize,
                              int big_endian);

/* Upload pipelined under the kernels.  The frame stays in the caller's PAGE-LOCKED memory (jxlt_pinned_alloc /
 * jxlt_pinned_register; anything else is refused) until the next jxlt_encode_enqueue on the context, which
 * fetches it in rows of DC groups (2048 pixel rows) and starts tile_kernel on each row as soon as that row
 * has arrived: all kernels but the last row's run under the PCIe transfer.  Planar planes (row pitch
 * pitch_bytes) or the raw payload of a PFM file (see jxlt_image_upload_pfm).  The memory must stay valid and
 * unchanged until that encode has been synchronised (any jxlt_fetch_* / jxlt_synchronize / jxlt_encode_resident*).
 * Replaces, like jxlt_image_upload, the host reads of CopyAndPadImage (enc_frame.cc:597-617) and of
 * ReadPFM's sample loop (read_pfm.cc:196-213). */
int jxlt_image_attach_host(jxlt_context* ctx, const float* const planes[3], size_t pitch_bytes, size_t xsize,
                           size_t ysize);
int jxlt_image_attach_host_pfm(jxlt_context* ctx, const void* host_payload, size_t xsize, size_t ysize,
                               int big_endian);

/* Size of the image currently set on the context. */
int jxlt_image_size(const jxlt_context* ctx, size_t* xsize, size_t* ysize);

/* Enqueues the whole per-group pipeline for the current image on the context's
 * stream (asynchronous).  One call == one "step" of the hot path. */
int jxlt_encode_enqueue(jxlt_context* ctx, const jxlt_params* params);
/* Blocks until the enqueued work has finished. */
int jxlt_synchronize(jxlt_context* ctx);
/* The reference computes two multipliers of its transform search from the distance of the FIRST
 * EncodeFile call of the process and reuses them afterwards (function-local static constants,
 * enc_ac_strategy.cc:178-185).  To reproduce a later call of such a process bit for bit, pass
 * that first distance here; 0 (default) = every encode uses its own distance. */
int jxlt_set_strategy_distance(jxlt_context* ctx, float first_call_distance);
/* How the library's calls wait for the device on this context.  0 (default), for one frame at a time: a wait that was
 * long the last two times (the kernels of a large frame) sleeps through most of the expected time and then polls, so
 * that results are seen within microseconds without a core spinning for milliseconds.  1, for contexts that share a GPU
 * and the host's CPUs with others (what the lanes of a jxlt_batch_encoder set): a few microseconds of polling, then
 * short sleeps -- a later wake-up is covered by the other contexts' frames; the context also trades a frame's latency for
 * fewer launches and events (resident frames of up to 1024 groups: every kernel of the frame on one stream, in order --
 * the other contexts' frames fill the device --, the DC histogram leaves with the AC histogram in one publication, both
 * section kinds' tile plans are one launch, a hand-over's completion is read off its copy stream instead of being
 * published by a kernel, no stage events: jxlt_kernel_times is not available).  Results do not depend on it. */
int jxlt_context_set_wait_mode(jxlt_context* ctx, int shared_device);
/* Copies results to pinned host memory (blocking) and fills *out. */
int jxlt_fetch_result(jxlt_context* ctx, jxlt_frame_result* out);

/* Production path: the raw tokens stay in HBM; only the two [64][64] symbol histograms leave the device -- AC
 * (pre-clustered contexts, static_entropy_codes.h:165; what OptimizeSections counts at enc_frame.cc:767-782) and DC
 * (the 45 DC / metadata contexts of WriteDCTokens / WriteACMetadataTokens, enc_frame.cc:287-424, which the device
 * tokenises too).  The DC histogram is complete before the AC tokenisation runs: jxlt_fetch_dc_histogram returns it
 * as soon as it is there, so that the DC code is built while the device is still busy.  jxlt_histograms_ready: 1
 * when jxlt_fetch_histograms would return without waiting, 0 when not yet (a read of host memory, no call into the
 * HIP runtime), < 0 on error. */
int jxlt_fetch_dc_histogram(jxlt_context* ctx, const uint32_t** dc_histogram);
int jxlt_fetch_histograms(jxlt_context* ctx, const uint32_t** ac_histograms, const uint32_t** dc_histograms);
int jxlt_histograms_ready(jxlt_context* ctx);

/* Section packing on the device -- the WriteToken loops of enc_frame.cc:784-800 with the caller's prefix codes
 * (code_table[ctx * 64 + symbol] = (depth << 16) | bits), kind 0 = DC-group sections (the raw-record form of
 * WriteDCGroup, enc_frame.cc:536-570), kind 1 = AC-group sections -- in three calls:
 *
 *   jxlt_pack_begin(kind, table)    asynchronous: lays the sections of the kind out back to back (byte aligned,
 *                                   enc_frame.cc:804-816) and entropy-codes them into a device blob -- in one pass
 *                                   (frames of up to 1024 groups) or a measuring and a writing pass
 *   jxlt_pack_sizes(kind, &out)     waits until the sizes are known: byte offsets + exact bit counts of the sections
 *                                   (all the TOC needs, enc_frame.cc:572-595); out->bytes is NULL
 *   jxlt_pack_deliver(kind, dst..)  asynchronous: the sections leave the device for `dst` -- page-locked host
 *                                   memory (jxlt_output_buffer / jxlt_pinned_alloc / jxlt_pinned_register) or
 *                                   device memory -- range by range while later sections are still being coded.
 *                                   Kind 1 waits for the sizes of the sections it sends (a fraction of a millisecond
 *                                   behind jxlt_pack_begin); kind 0 never waits: if the DC-group sections' sizes
 *                                   have not arrived, the library issues the copies itself when they do (from its
 *                                   next wait, at the latest in jxlt_pack_sizes / jxlt_synchronize).  Complete
 *                                   after jxlt_synchronize.
 *
 * jxlt_pack_deliver places all sections of the kind back to back STARTING at dst (end_aligned 0) or ENDING at dst
 * (end_aligned 1: for a kind whose total size the caller does not know yet -- the DC-group sections are set against
 * the start of ACGlobal from the right, the frame's head against them).  With `runs` the sections go out in pieces
 * instead: run r = sections [first_section, first_section + num_sections) of this context's frame, back to back at
 * dst + dst_offset (a slab of a frame that several GPUs share owns several ranges of the codestream). */
typedef struct {
  const uint8_t* bytes;            /* NULL, or pinned host memory owned by the context */
  const uint64_t* section_offset;  /* [num_sections + 1] byte offsets of the sections laid out back to back */
  const uint32_t* section_bits;    /* [num_sections] exact bit length of each section */
  size_t num_sections;
} jxlt_packed_sections;
typedef struct {
  uint32_t first_section, num_sections;
  uint64_t dst_offset;
} jxlt_section_run;
int jxlt_pack_begin(jxlt_context* ctx, int kind, const uint32_t* code_table);
int jxlt_pack_sizes(jxlt_context* ctx, int kind, jxlt_packed_sections* out);
int jxlt_pack_deliver(jxlt_context* ctx, int kind, uint8_t* dst, const jxlt_section_run* runs, size_t num_runs,
                      int end_aligned);
/* A page-locked, context-owned buffer of at least `bytes` bytes to assemble a codestream in (valid until the next
 * call that asks for a larger one / destroy; it keeps its contents when it grows). */
int jxlt_output_buffer(jxlt_context* ctx, size_t bytes, uint8_t** out);

/* Statistics of the last encode of `ctx` (waits for its tile kernels).  The entropy estimate of the transform
 * search takes a square root per coefficient (enc_ac_strategy.cc:118-126); the kernel reads the roots of quantised
 * magnitudes below 1024 from a table, and a tile that meets a larger one (tiny distances, samples far above 1.0)
 * is done again with computed roots by a second, small launch -- same results, the tile's work twice. */
typedef struct {
  uint32_t tiles;                      /* 64x64 tiles of the frame */
  uint32_t tiles_redone_exact_roots;   /* ... of which were redone with computed roots */
  uint32_t encodes_with_redone_tiles;  /* encodes of this context so far in which any tile was */
  /* Host time inside the copy commands the library has issued for the last encode (hipMemcpyAsync of the packed
   * sections): how many, and the longest CALL in microseconds -- a call that runs into the runtime creating a copy
   * engine's queue takes milliseconds; bench.py attributes slow steps with it. */
  uint32_t copy_calls;
  float longest_copy_call_us;
} jxlt_encode_stats_t;
int jxlt_encode_stats(jxlt_context* ctx, jxlt_encode_stats_t* out);

/* ---- libjxltiny_host.so ------------------------------------------------ */

void jxlt_compute_distance_params(float distance, jxlt_distance_params* out);

/* Host bitstream back-end (enc_frame.cc:765-858 after the pixel pipeline):
 * appends frame header, TOC and all sections for `frame` to a malloc'ed buffer
 * (*out_bytes, caller frees with jxlt_free).  num_threads <= 0: all cores. */
int jxlt_assemble_frame(const jxlt_frame_result* frame, const jxlt_distance_params* distp,
                        int num_threads, uint8_t** out_bytes, size_t* out_size);
/* Full codestream (file header + frame) of the image currently set on `ctx`
 * (jxlt_image_upload / jxlt_image_set_device): device pipeline, device-side
 * section packing, host assembly.  malloc'ed result, free with jxlt_free. */
int jxlt_encode_resident(jxlt_context* ctx, float distance, int num_threads, uint8_t** out_bytes,
                         size_t* out_size);

/* Same, without the final host copy: the codestream is assembled in the context's
 * page-locked output buffer (the packed AC sections are copied from the device straight
 * into place); *bytes stays valid until the next encode on `ctx`. */
int jxlt_encode_resident_view(jxlt_context* ctx, float distance, int num_threads, const uint8_t** bytes,
                              size_t* size);

/* C entry to the drop-in EncodeFile (enc_file.h:20-21): planar f32 in host
 * memory -> complete .jxl codestream (malloc'ed, free with jxlt_free).
 * device_ordinal selects the GPU. */
int jxlt_encode_file_planar(const float* const planes[3], size_t pitch_bytes, size_t xsize,
                            size_t ysize, float distance, int device_ordinal,
                            uint8_t** out_bytes, size_t* out_size);
/* The same over a LIST of devices (jxl::SetEncoderDevices + jxl::EncodeFile in one call): a frame of more than one DC
 * group is cut into rectangles of whole DC groups, one per listed device; same bytes.  The list and the encoder built
 * on it belong to the calling thread (two threads with two lists encode side by side).  num_devices <= 1, or a frame
 * of one DC group: as jxlt_encode_file_planar on device_ordinals[0]. */
int jxlt_encode_file_planar_devices(const float* const planes[3], size_t pitch_bytes, size_t xsize, size_t ysize,
                                    float distance, const int* device_ordinals, int num_devices, uint8_t** out_bytes,
                                    size_t* out_size);
/* cjxl_tiny's whole job in one call: PFM file -> .jxl codestream (malloc'ed, free with
 * jxlt_free), with the PFM payload de-interleaved / flipped / byte-swapped by the device
 * kernels (jxlt_image_upload_pfm) instead of by a host pass. */
int jxlt_encode_pfm_file(const char* filename, float distance, int device_ordinal, uint8_t** out_bytes,
                         size_t* out_size);
/* ---- frame batches (BASELINE config #5) ------------------------------------------
 * A batch of independent frames on one GPU: `lanes` device contexts, each driven by its own host
 * thread with its own HIP stream, take frames from a shared queue, so that the upload of one
 * frame (PCIe) overlaps the kernels of another and the download of a third.  Every frame gives
 * the same bytes as jxlt_encode_file_planar / jxlt_encode_pfm_file.  The reference has no batch
 * entry (cjxl_tiny is one image per process, cjxl_main.cc:49-101); this is the loop a caller
 * would write around EncodeFile (enc_file.h:20-21), moved behind the boundary so that the
 * overlap is the library's business.  Frames may differ in size.  Page-locked source memory
 * (jxlt_pinned_alloc) is copied at PCIe speed; ordinary memory is staged by the lane's thread. */
typedef struct {
  const float* planes[3];  /* planar f32 with row pitch pitch_bytes; all NULL when pfm_payload is set */
  size_t pitch_bytes;
  const void* pfm_payload; /* or: interleaved bottom-up RGB f32 payload of a PFM file (read_pfm.cc:199-209) */
  int pfm_big_endian;
  size_t xsize, ysize;
  int in_device_memory;    /* non-zero: planes / pfm_payload point into device memory of the encoder's GPU and
                              are read in place (jxlt_image_set_device*); they must stay valid during the run */
  int device_ordinal;      /* in_device_memory frames of a multi-device encoder: the GPU that holds them */
} jxlt_batch_frame;
typedef struct jxlt_batch_encoder jxlt_batch_encoder;
/* lanes <= 0: 3 (upload / encode / download in flight at once). */
int jxlt_batch_encoder_create(int device_ordinal, int lanes, jxlt_batch_encoder** out);
/* The same over several GPUs (BASELINE config #5: frames round-robin over 8 GPUs): lanes_per_device contexts
 * on every listed device, all fed from one frame queue.  Frames in host memory go to whichever lane is free
 * (page-locked memory from jxlt_pinned_alloc is visible to every device); a frame with in_device_memory set
 * is taken by a lane of the device named in its device_ordinal field. */
int jxlt_batch_encoder_create_multi(const int* device_ordinals, int num_devices, int lanes_per_device,
                                    jxlt_batch_encoder** out);
void jxlt_batch_encoder_destroy(jxlt_batch_encoder* enc);
/* Encodes frames[0..num_frames) at `distance`.  out_bytes[i] (malloc'ed, free with jxlt_free) and
 * out_sizes[i] receive the codestream of frame i.  Returns the first failing frame's error; the
 * outputs of failed frames are NULL / 0. */
int jxlt_batch_encoder_run(jxlt_batch_encoder* enc, const jxlt_batch_frame* frames, size_t num_frames,
                           float distance, uint8_t** out_bytes, size_t* out_sizes);
/* jxl::EmulateReferenceStaticConstants (host/encoder/enc_frame.h): latch the first distance this
 * process encodes with for the transform search's two multipliers, as the reference library
 * does (enc_ac_strategy.cc:178-185).  Off by default. */
void jxlt_emulate_reference_static_constants(int on);
/* jxl::EmulateReferenceSingleSymbolCodes (host/encoder/enc_frame.h).  A prefix code with a single
 * used symbol is serialised as a one-symbol code, which decoders read with zero bits per token;
 * the reference nevertheless writes one bit per such token (enc_huffman_tree.cc:84-87 leaves a
 * placeholder depth that enc_entropy_code.cc:411-416 never resets), so its output cannot be
 * decoded in those (rare, flat-content) cases.  Default 0: conformant zero-bit tokens -- the
 * codestream equals the reference's whenever the reference's is decodable.  1: the reference's
 * bytes in every case. */
void jxlt_emulate_reference_single_symbol_codes(int on);
/* Host-side stage times of the calling thread's last frame through jxl::EncodeFrame / EncodeFile /
 * jxlt_encode_resident* (what JXLT_TRACE=1 prints), milliseconds from the start of the call: the DC histogram in host
 * memory (= the tile kernels are done), the AC histogram (= tokenisation done), both codes built and handed to the
 * device, both kinds' section sizes known, the last section byte in host memory (= the call's end).  bench.py
 * attributes slow steps with it.  Returns JXLT_ERR_INVALID_ARGUMENT before the thread's first frame. */
typedef struct {
  double dc_histogram_ms, ac_histogram_ms, codes_ms, sizes_ms, done_ms;
} jxlt_frame_timeline;
int jxlt_last_frame_timeline(jxlt_frame_timeline* out);
/* Codestream + image headers that precede the frame (enc_file.cc:70-95). */
int jxlt_write_file_header(size_t xsize, size_t ysize, uint8_t** out_bytes, size_t* out_size);
/* ---- one frame sharded over several GPUs (BASELINE config #4) -----------------
 * The reference's unit of independent work is the DC group (the loop enc_frame.cc:839-844); its one global
 * synchronisation point is the code optimisation over all sections (enc_frame.cc:846-850).  Here a frame is
 * cut into rectangles of whole DC groups (2048 x 2048 pixels; jxlt_shard_rect), one per participant; every participant
 * runs the complete device pipeline on its slab with its own device context.  The only exchange is on the
 * host: the 2 x 64 x 64 symbol histograms are summed by participant 0, which builds the two prefix codes and
 * hands the code tables back; every participant then entropy-codes its sections on its GPU and copies them
 * straight into its byte range of ONE output buffer, where participant 0 writes frame header, TOC and the
 * global sections in front.  No device-to-device traffic, no RCCL.
 *
 * Participants are host threads of one process (jxlt_multi_encoder_*, also behind jxl::EncodeFile when
 * jxl::SetEncoderDevices / JXLT_DEVICES names several GPUs) or one process per GPU that meet in a POSIX
 * shared-memory segment (jxlt_shard_group_*: what bench.py's ranks use). */

/* The pixels [*x0, *x1) x [*y0, *y1) of participant `rank` of `world` for an xsize x ysize frame: a rectangle of
 * whole DC groups ("groups shard by index": the frame's grid of DC groups is cut into bands of rows and every band
 * into column ranges, so that the largest rectangle is as small as possible -- 16384^2 over 8: eight rows of DC
 * groups; 16384 x 2048 over 8: one DC group each; 8192^2 over 8: 2 x 1 DC groups each).  An empty rectangle when
 * there are fewer DC groups than participants.  A rectangle's sections are not one range of the codestream: every
 * participant's device writes one run of bytes per row of its DC groups / groups (jxlt_pack_deliver, runs). */
int jxlt_shard_rect(size_t xsize, size_t ysize, int world, int rank, size_t* x0, size_t* y0, size_t* x1, size_t* y1);

typedef struct jxlt_multi_encoder jxlt_multi_encoder;
/* One device context + one host thread per entry of device_ordinals (an ordinal may repeat: several
 * contexts on one GPU). */
int jxlt_multi_encoder_create(const int* device_ordinals, int num_devices, jxlt_multi_encoder** out);
void jxlt_multi_encoder_destroy(jxlt_multi_encoder* enc);
const char* jxlt_multi_encoder_last_error(const jxlt_multi_encoder* enc);
/* EncodeFile (enc_file.h:20-21) over all devices: every device uploads its slab of the caller's planes
 * (planar f32, row pitch pitch_bytes; page-locked memory from jxlt_pinned_alloc goes over each GPU's own
 * PCIe link at full speed) and encodes it.  *bytes: the complete codestream, owned by the encoder, valid
 * until its next encode.  Same bytes as jxlt_encode_file_planar. */
int jxlt_multi_encoder_encode(jxlt_multi_encoder* enc, const float* const planes[3], size_t pitch_bytes,
                              size_t xsize, size_t ysize, float distance, const uint8_t** bytes, size_t* size);
/* The same for the raw sample payload of a PFM file (jxlt_image_upload_pfm). */
int jxlt_multi_encoder_encode_pfm(jxlt_multi_encoder* enc, const void* host_payload, size_t xsize, size_t ysize,
                                  int big_endian, float distance, const uint8_t** bytes, size_t* size);
/* A frame that already is in the devices' HBM: slab `slab` -- the rectangle jxlt_shard_rect(xsize, ysize,
 * num_devices, slab), xsize_of_slab x rows pixels -- as three planes in the memory of that slab's device; then
 * encode.  The planes are read in place. */
int jxlt_multi_encoder_set_device_slab(jxlt_multi_encoder* enc, int slab, const void* const device_planes[3],
                                       size_t pitch_bytes, size_t xsize_of_slab, size_t rows);
int jxlt_multi_encoder_encode_resident(jxlt_multi_encoder* enc, size_t xsize, size_t ysize, float distance,
                                       const uint8_t** bytes, size_t* size);

/* One process per GPU.  Rank 0 creates the segment `shm_name` ("/name"; control block, section-size tables
 * for up to max_sections sections, output_capacity bytes for the codestream), the others attach to it AFTER
 * rank 0's call has returned (callers barrier in between) and take its geometry (their own capacity arguments
 * are not consulted).  Closing detaches; rank 0 also unlinks. */
typedef struct jxlt_shard_group jxlt_shard_group;
int jxlt_shard_group_open(const char* shm_name, int rank, int world, size_t output_capacity, size_t max_sections,
                          jxlt_shard_group** out);
void jxlt_shard_group_close(jxlt_shard_group* group);
const char* jxlt_shard_group_last_error(const jxlt_shard_group* group);
/* Collective over the group's ranks: `ctx` holds this rank's slab of an xsize x ysize frame (rows
 * jxlt_shard_rect(xsize, ysize, world, rank); jxlt_image_upload* / jxlt_image_set_device* with its width and height; ranks
 * with an empty range pass any context).  On rank 0 *bytes / *size receive the complete codestream (inside
 * the segment, valid until the next encode); NULL / 0 on the other ranks. */
int jxlt_shard_encode(jxlt_shard_group* group, jxlt_context* ctx, size_t xsize, size_t ysize, float distance,
                      const uint8_t** bytes, size_t* size);
/* Frames in flight over the group.  What does not shrink with the number of GPUs is the stage between the
 * kernels and the section packing (histogram hand-over, code construction: enc_frame.cc:846-850, layout), during
 * which every GPU of a jxlt_shard_group waits.  A pipeline is `depth` such groups (segments <shm_name>.<lane>),
 * each with a device context of its own and a host thread: frame k runs on lane k % depth, so the kernels of
 * frame k + 1 run on every GPU while frame k is in that stage.  Same contract as jxlt_shard_group_open: rank 0
 * opens first, then the others; every rank submits the same frames in the same order.  jxlt_shard_pipeline_close
 * encodes what has been submitted before it returns.  A frame that fails on any rank fails on all of them, and the
 * failure is sticky for its lane (the ranks cannot agree on a reset while some of them may still be inside the
 * frame): close the pipeline on every rank and open a new one. */
typedef struct jxlt_shard_pipeline jxlt_shard_pipeline;
int jxlt_shard_pipeline_open(const char* shm_name, int rank, int world, int device_ordinal, int depth,
                             size_t output_capacity, size_t max_sections, jxlt_shard_pipeline** out);
void jxlt_shard_pipeline_close(jxlt_shard_pipeline* pipeline);
const char* jxlt_shard_pipeline_last_error(const jxlt_shard_pipeline* pipeline);
/* This rank's slab of the next frame (the rectangle jxlt_shard_rect(xsize, ysize, world, rank), device_planes
 * pointing at its first sample, slab_rows its height; slab_rows 0: none), resident in
 * device memory and read in place: it must stay valid until the frame's wait has returned.  Returns at once
 * unless the lane's previous frame (depth frames back) is still being encoded. */
int jxlt_shard_pipeline_submit_device(jxlt_shard_pipeline* pipeline, const void* const device_planes[3],
                                      size_t pitch_bytes, size_t xsize, size_t ysize, size_t slab_rows, float distance,
                                      uint64_t* ticket);
/* Waits for frame `ticket`.  On rank 0 *bytes / *size receive the codestream (inside the lane's segment: valid
 * until `depth` further frames have been submitted); NULL / 0 on the other ranks. */
int jxlt_shard_pipeline_wait(jxlt_shard_pipeline* pipeline, uint64_t ticket, const uint8_t** bytes, size_t* size);

void jxlt_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* JXL_TINY_AMD_H_ */
