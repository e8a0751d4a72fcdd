// Drop-in for /root/reference/encoder/enc_frame.h:19-20.
#ifndef JXLT_HOST_ENCODER_ENC_FRAME_H_
#define JXLT_HOST_ENCODER_ENC_FRAME_H_

#include "encoder/base/data_parallel.h"
#include "encoder/base/status.h"
#include "encoder/enc_bit_writer.h"
#include "encoder/image.h"

namespace jxl {

// Encodes one frame of `linear` (planar linear-sRGB f32) at butteraugli
// `distance` and appends it to `writer` (byte aligned on entry).  The per-group
// pixel pipeline runs on the MI355X selected by SetEncoderDevice(); `pool` only
// supplies the host thread count.  Returns false on failure (no GPU, invalid
// arguments); never falls back to a CPU path.
Status EncodeFrame(float distance, const Image3F& linear, ThreadPool* pool, BitWriter* writer);

// Not in the reference: HIP device ordinal used by the calling thread's
// subsequent EncodeFrame/EncodeFile calls (default 0).
void SetEncoderDevice(int device_ordinal);
// Not in the reference: several GPUs for ONE frame, for the calling thread's subsequent calls (like
// SetEncoderDevice: every thread has its own list and its own encoder over it, so two threads with two lists -- or
// with the same list -- encode side by side; until round 4 the list was process-wide and concurrent callers were
// serialised).  With more than one ordinal, frames of
// at least two DC-group rows (> 2048 pixel rows) are cut into row slabs of whole DC groups, one per device
// (the reference's loop over DC groups, enc_frame.cc:839-844, spread over the GPUs; include/jxl_tiny_amd.h,
// jxlt_multi_encoder_*); same bytes as on one GPU.  n <= 1 returns to the single-device path.  The environment
// variable JXLT_DEVICES ("0,1,2,3" or "all") is the list of every thread that has not called this function
// (unmodified callers such as cjxl_tiny).
void SetEncoderDevices(const int* device_ordinals, int n);

// Not in the reference: the reference derives two multipliers of its transform search from the
// distance of the FIRST frame the process encodes (function-local static constants,
// enc_ac_strategy.cc:178-185) and reuses them for every later frame.  Off (default): every frame
// uses its own distance, which is what a single-image cjxl_tiny run does anyway.  On: the first
// distance this process encodes with is latched the same way, so that a multi-image process is
// byte-identical to the reference library used the same way.
void EmulateReferenceStaticConstants(bool on);
// Not in the reference either: when a (clustered) histogram has a single used symbol, the reference
// serialises a one-symbol prefix code -- which a decoder reads with zero bits per token -- but
// still writes the construction's placeholder depth of one bit per token
// (enc_huffman_tree.cc:84-87, enc_entropy_code.cc:411-416, enc_entropy_code.h:34-42), i.e. a
// stream that cannot be decoded.  Rare (flat synthetic content).  Off (default): such tokens get
// zero bits, output conformant; everything else is byte-identical to the reference.  On: the
// reference's bytes in those cases too.
void EmulateReferenceSingleSymbolCodes(bool on);

}  // namespace jxl

#endif  // JXLT_HOST_ENCODER_ENC_FRAME_H_
