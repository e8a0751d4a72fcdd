// LSB-first bit packer used by the host bitstream back-end.
// Drop-in counterpart of the reference's jxl::BitWriter
// (/root/reference/encoder/enc_bit_writer.h:27-119): same observable bit order
// (bits fill bytes from the least significant bit, bytes in increasing address),
// same 56-bit-per-call limit.  Storage management is different: a plain
// std::vector plus a 64-bit accumulator, no Allotment bookkeeping.
#ifndef JXLT_HOST_ENCODER_ENC_BIT_WRITER_H_
#define JXLT_HOST_ENCODER_ENC_BIT_WRITER_H_

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace jxl {

class BitWriter {
 public:
  static constexpr size_t kMaxBitsPerCall = 56;

  BitWriter() = default;
  BitWriter(const BitWriter&) = delete;
  BitWriter& operator=(const BitWriter&) = delete;
  BitWriter(BitWriter&&) = default;
  BitWriter& operator=(BitWriter&&) = default;

  size_t BitsWritten() const { return bytes_.size() * 8 + pending_bits_; }

  void Reserve(size_t bytes) { bytes_.reserve(bytes_.size() + bytes); }

  // Appends the low n_bits (<= 56) of `bits`.
  inline void Write(size_t n_bits, uint64_t bits) {
    if (n_bits > 32) {  // keep pending(<32) + n <= 63 so the shift cannot overflow
      Write(32, bits & 0xFFFFFFFFu);
      bits >>= 32;
      n_bits -= 32;
    }
    acc_ |= bits << pending_bits_;
    pending_bits_ += static_cast<uint32_t>(n_bits);
    if (pending_bits_ >= 32) {
      const uint32_t low = static_cast<uint32_t>(acc_);
      const size_t pos = bytes_.size();
      bytes_.resize(pos + 4);
      memcpy(bytes_.data() + pos, &low, 4);  // little-endian host
      acc_ >>= 32;
      pending_bits_ -= 32;
    }
  }

  void ZeroPadToByte() {
    const uint32_t rem = pending_bits_ & 7;
    if (rem != 0) Write(8 - rem, 0);
  }

  // Bitwise concatenation (reference: enc_bit_writer.cc:90-108).
  void Append(const BitWriter& other) {
    for (uint8_t b : other.bytes_) Write(8, b);
    uint64_t acc = other.acc_;
    uint32_t left = other.pending_bits_;
    while (left >= 8) {
      Write(8, acc & 0xFF);
      acc >>= 8;
      left -= 8;
    }
    if (left) Write(left, acc & ((1u << left) - 1));
  }

  // Byte-aligned concatenation of raw bytes; *this must be byte aligned.
  void AppendBytes(const uint8_t* data, size_t size) {
    Flush();
    bytes_.insert(bytes_.end(), data, data + size);
  }

  // Pads every writer to a byte boundary and appends them
  // (reference: enc_bit_writer.cc:58-88).
  void AppendByteAligned(std::vector<BitWriter>* others) {
    for (BitWriter& w : *others) {
      w.ZeroPadToByte();
      w.Flush();
      AppendBytes(w.bytes_.data(), w.bytes_.size());
    }
  }

  // Requires byte alignment.  Moves the bytes out.
  std::vector<uint8_t> TakeBytes() {
    Flush();
    return std::move(bytes_);
  }
  // Requires byte alignment.
  const std::vector<uint8_t>& Bytes() {
    Flush();
    return bytes_;
  }

 private:
  void Flush() {  // moves whole pending bytes into storage
    while (pending_bits_ >= 8) {
      bytes_.push_back(static_cast<uint8_t>(acc_ & 0xFF));
      acc_ >>= 8;
      pending_bits_ -= 8;
    }
  }
  std::vector<uint8_t> bytes_;
  uint64_t acc_ = 0;
  uint32_t pending_bits_ = 0;  // < 32 between calls
};

}  // namespace jxl

#endif  // JXLT_HOST_ENCODER_ENC_BIT_WRITER_H_
