// inflate_fast.h -- raw DEFLATE (RFC 1951) decoder for BGZF blocks.
//
// A BGZF block is a complete deflate stream of at most 64 KiB whose inflated size is known from the gzip trailer, and
// inflating is what an end-to-end run of `conga` waits for.  zlib's inflate() is a general,
// resumable, byte-at-a-time state machine; this one decodes a whole block in one call with a 64-bit bit buffer that is
// refilled eight bytes at a time, 10- / 8-bit first-level tables with second-level tables behind them, and word-wise
// match copies.  Every block is still checked against its CRC32 by the caller, which falls back to zlib if this decoder
// refuses a stream or gets the checksum wrong (tests/test_inflate.py compares the two on thousands of streams).
#pragma once
#include <cstddef>
#include <cstdint>

namespace conga_host {

// Inflates in[0, in_len) into out[0, out_len).  true iff the stream is well formed, ends with its final block inside
// the input, and produces exactly out_len bytes.  Never reads or writes outside the two buffers.
bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);

} // namespace conga_host
