// inflate_fast.cpp -- see inflate_fast.h; the decoder itself is inflate_core.h (shared with the GPU prototype).
#include "inflate_fast.h"

#include "inflate_core.h"

namespace conga_host {

bool inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
	static thread_local inflate_core::Decoder dec;
	return inflate_core::inflate_block_stream(dec, in, in_len, out, out_len);
}

} // namespace conga_host
