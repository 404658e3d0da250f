// inflate_core.h -- the block decoder of inflate_fast.h, as host + device source: the BAM reader calls it through
// inflate_fast.cpp (one scratch Decoder per thread), tools/gpu_inflate.hip runs the same code one BGZF block per GPU
// lane.  Written against RFC 1951; no code of zlib or any other inflater is used.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define CONGA_INFL_HD __host__ __device__
#else
#define CONGA_INFL_HD
#endif

namespace conga_host {
namespace inflate_core {


// ---- table entries (32 bits) -----------------------------------------------------------------------------------
//   [7:0]   bits to consume: the codeword's length in this table level (for a pointer entry: the first-level width)
//   [12:8]  number of extra bits (length / distance symbols), or the width of the second-level table (pointer)
//   [27:13] literal byte / length base / distance base / start of the second-level table
//   [28]    entry is a codeword (0: no codeword maps here -> the stream is invalid)
//   [29]    end of block        [30] pointer to a second-level table        [31] literal
constexpr uint32_t kValid = 1u << 28, kEnd = 1u << 29, kSub = 1u << 30, kLiteral = 1u << 31;
constexpr int kLitBits = 10, kDistBits = 8, kPreBits = 7;
// Second-level space: zlib, whose tables are laid out the same way, proves 852 entries in all enough for 286 symbols
// behind a 9-bit first level and 592 for 30 symbols behind a 6-bit one (enough.c); wider first levels need less behind
// them.  1024 and 512 second-level entries are comfortably above that; build_table refuses what does not fit and the
// callers fall back (zlib on the host, the host decoders for the GPU stage).  Small tables matter on the GPU: a decoder
// scratch per lane, 17 KB instead of 64.
constexpr int kLitSize = (1 << kLitBits) + 1024, kDistSize = (1 << kDistBits) + 512, kPreSize = 1 << kPreBits;


enum { kKindLitLen, kKindDist, kKindPre };

CONGA_INFL_HD inline uint32_t symbol_entry(int kind, int sym)
{
	// (function-local so that device code can index them too)
	const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
	const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
	const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
	const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
	if (kind == kKindLitLen) {
		if (sym < 256)
			return kValid | kLiteral | ((uint32_t) sym << 13);
		if (sym == 256)
			return kValid | kEnd;
		if (sym <= 285)
			return kValid | ((uint32_t) kLenBase[sym - 257] << 13) | ((uint32_t) kLenExtra[sym - 257] << 8);
		return 0; // 286, 287: in the fixed code, never legal in data
	}
	if (kind == kKindDist)
		return sym < 30 ? kValid | ((uint32_t) kDistBase[sym] << 13) | ((uint32_t) kDistExtra[sym] << 8) : 0;
	return kValid | ((uint32_t) sym << 13);
}

CONGA_INFL_HD inline uint32_t reverse_bits(uint32_t v, int n)
{
	uint32_t r = 0;
	for (int i = 0; i < n; i++) {
		r = (r << 1) | (v & 1u);
		v >>= 1;
	}
	return r;
}

// Canonical Huffman code of `lens` (RFC 1951 3.2.2) as a two-level lookup table indexed by the next bits of the
// stream, least significant bit first.  false: over-subscribed or (beyond what zlib lets pass) incomplete set of lengths.
CONGA_INFL_HD inline bool build_table(const uint8_t *lens, int n, int kind, int first_bits, uint32_t *table, int capacity)
{
	int count[16] = {0};
	for (int i = 0; i < n; i++)
		count[lens[i]]++;
	count[0] = 0;
	int left = 1, max_len = 0;
	for (int len = 1; len <= 15; len++) {
		left = (left << 1) - count[len];
		if (left < 0)
			return false;
		if (count[len])
			max_len = len;
	}
	// an incomplete code is only legal as "no code at all" or one-bit codes (a single distance code): zlib's rule
	if (left > 0 && max_len != 0 && (kind == kKindPre || max_len != 1))
		return false;
	uint32_t next[16];
	uint32_t code = 0;
	for (int len = 1; len <= 15; len++) {
		code = (code + (uint32_t) count[len - 1]) << 1;
		next[len] = code;
	}
	const int first_size = 1 << first_bits;
	const uint32_t first_mask = (uint32_t) first_size - 1u;
	memset(table, 0, (size_t) first_size * sizeof(uint32_t));
	uint16_t rev[288];
	uint8_t second_bits[1 << kLitBits];
	memset(second_bits, 0, (size_t) first_size);
	for (int i = 0; i < n; i++) {
		const int len = lens[i];
		if (!len)
			continue;
		rev[i] = (uint16_t) reverse_bits(next[len]++, len);
		if (len > first_bits) {
			uint8_t &sb = second_bits[rev[i] & first_mask];
			if (len - first_bits > sb)
				sb = (uint8_t) (len - first_bits);
		}
	}
	int used = first_size;
	for (int p = 0; p < first_size; p++) {
		if (!second_bits[p])
			continue;
		const int size = 1 << second_bits[p];
		if (used + size > capacity)
			return false;
		table[p] = kValid | kSub | ((uint32_t) used << 13) | ((uint32_t) second_bits[p] << 8) | (uint32_t) first_bits;
		memset(table + used, 0, (size_t) size * sizeof(uint32_t));
		used += size;
	}
	for (int i = 0; i < n; i++) {
		const int len = lens[i];
		if (!len)
			continue;
		const uint32_t e = symbol_entry(kind, i);
		if (len <= first_bits) {
			for (uint32_t j = rev[i]; j < (uint32_t) first_size; j += 1u << len)
				table[j] = e | (uint32_t) len;
		} else {
			const uint32_t ptr = table[rev[i] & first_mask];
			const uint32_t start = (ptr >> 13) & 0x7FFFu, sb = (ptr >> 8) & 31u;
			const int l2 = len - first_bits;
			for (uint32_t j = (uint32_t) rev[i] >> first_bits; j < (1u << sb); j += 1u << l2)
				table[start + j] = e | (uint32_t) l2;
		}
	}
	return true;
}

struct Decoder {
	uint32_t lit[kLitSize];
	uint32_t dist[kDistSize];
	uint32_t pre[kPreSize];
	uint32_t fixed_lit[kLitSize > 1024 ? 1024 : kLitSize]; // the fixed code has no codeword longer than 9 bits
	uint32_t fixed_dist[1 << kDistBits];
	bool fixed_ready = false;
};

CONGA_INFL_HD inline uint64_t load64(const uint8_t *p)
{
	uint64_t v;
	memcpy(&v, p, 8); // little-endian hosts only (x86-64, the GPU boxes)
	return v;
}

struct Bits {
	const uint8_t *ip, *iend;
	uint64_t buf = 0;
	int cnt = 0; // valid bits in buf; negative once more bits were consumed than the input holds

	// the eight bytes at ip (at least eight are left)
#if defined(__HIP_DEVICE_COMPILE__)
	// On the GPU a load is the better part of a microsecond, so the input is read AHEAD: three aligned words slide along
	// the stream, the newest of them asked for long before its bytes are needed, and next8() only shifts registers.
	// (Reads up to 24 bytes past ip and 7 in front of the stream's first byte: the caller's buffer has that slack.)
	const uint8_t *wbase = nullptr;
	uint64_t w0 = 0, w1 = 0, w2 = 0;
	__device__ inline uint64_t next8()
	{
		size_t off = (size_t) (ip - wbase);
		if (off >= 16) { // first use, or the stream position jumped (stored block)
			wbase = (const uint8_t *) ((uintptr_t) ip & ~(uintptr_t) 7);
			w0 = *reinterpret_cast<const uint64_t *>(wbase);
			w1 = *reinterpret_cast<const uint64_t *>(wbase + 8);
			w2 = *reinterpret_cast<const uint64_t *>(wbase + 16);
			off = (size_t) (ip - wbase);
		} else if (off >= 8) {
			w0 = w1;
			w1 = w2;
			wbase += 8;
			w2 = *reinterpret_cast<const uint64_t *>(wbase + 16);
			off -= 8;
		}
		return off ? (w0 >> (8 * off)) | (w1 << (64 - 8 * off)) : w0;
	}
#else
	inline uint64_t next8() { return load64(ip); }
#endif

	// at least 56 valid bits afterwards, or everything that is left of the input
	CONGA_INFL_HD inline void refill()
	{
		if (cnt < 0)
			return; // (the caller is about to notice)
		if (iend - ip >= 8) {
			buf |= next8() << cnt; // (bits above cnt are the stream's own next bits: or-ing them again later is harmless)
			ip += (63 - cnt) >> 3;
			cnt |= 56;
		} else {
			while (cnt <= 56 && ip < iend) {
				buf |= (uint64_t) *ip++ << cnt;
				cnt += 8;
			}
		}
	}
	// the same with at least eight bytes of input left (the caller knows)
	CONGA_INFL_HD inline void refill_fast()
	{
		buf |= next8() << cnt;
		ip += (63 - cnt) >> 3;
		cnt |= 56;
	}
	CONGA_INFL_HD inline uint32_t peek(int n) const { return (uint32_t) (buf & ((1ull << n) - 1ull)); }
	CONGA_INFL_HD inline void drop(int n)
	{
		buf >>= n;
		cnt -= n;
	}
	CONGA_INFL_HD inline uint32_t take(int n)
	{
		const uint32_t v = peek(n);
		drop(n);
		return v;
	}
};

// next symbol of a two-level table; 0 (not kValid) when the bits match no codeword
CONGA_INFL_HD inline uint32_t decode(Bits &b, const uint32_t *table, int first_bits)
{
	uint32_t e = table[b.peek(first_bits)];
	if (e & kSub) {
		b.drop((int) (e & 0xFFu));
		e = table[((e >> 13) & 0x7FFFu) + b.peek((int) ((e >> 8) & 31u))];
	}
	b.drop((int) (e & 0xFFu));
	return e;
}


// How a block's symbols are turned into bytes is a policy: this one is the sequential decoder (one stream per thread);
// a GPU wave that decodes one stream with all its lanes brings its own (tools/gpu_inflate_wave.hip).
struct SequentialSymbols {
	// 0: the block's end-of-block symbol was reached; -1: the stream is invalid
	CONGA_INFL_HD static inline int run(Bits &b, const uint32_t *lit, const uint32_t *dist, uint8_t *out, uint8_t *&op, uint8_t *oend)
	{
		// ---- the block's symbols.  One refill covers a whole length / distance pair: 15 + 5 + 15 + 13 = 48 bits.
		for (bool end_of_block = false; !end_of_block;) {
			// Fast trips: with 16 bytes of input ahead neither refill of a trip can run dry (each takes 7 bytes at most), and
			// with 274 bytes of room neither three literals nor a 258-byte match copied in eight-byte steps can run over:
			// no bounds checks inside.
			while ((b.iend - b.ip) >= 16 && (oend - op) >= 274) {
				b.refill_fast();
				uint32_t e = decode(b, lit, kLitBits);
				if (e & kLiteral) {
					*op++ = (uint8_t) (e >> 13);
					e = decode(b, lit, kLitBits);
					if (e & kLiteral) {
						*op++ = (uint8_t) (e >> 13);
						e = decode(b, lit, kLitBits);
						if (e & kLiteral) {
							*op++ = (uint8_t) (e >> 13);
							continue;
						}
					}
					b.refill_fast();
				}
				if (!(e & kValid))
					return -1;
				if (e & kEnd) {
					end_of_block = true;
					break;
				}
				const size_t length = ((e >> 13) & 0x7FFFu) + b.take((int) ((e >> 8) & 31u));
				e = decode(b, dist, kDistBits);
				if (!(e & kValid))
					return -1;
				const size_t offset = ((e >> 13) & 0x7FFFu) + b.take((int) ((e >> 8) & 31u));
				if (offset > (size_t) (op - out) || length > 258) // (no table built by build_table yields more than 258)
					return -1;
				// Eight bytes at a time (up to seven more than `length` are written; there is room).  A match closer than eight
				// bytes is periodic with its distance: after one period has been copied byte by byte the same bytes are also a
				// match at twice the distance, so at most three short rounds get it to eight.  (On the GPU the lanes of a wave
				// wait for the longest inner loop among them: a 258-byte run copied byte by byte was everybody's pace.)
				const uint8_t *src = op - offset;
				uint8_t *d = op;
				ptrdiff_t left = (ptrdiff_t) length;
				size_t dist = offset;
				while (dist < 8 && left > 0) {
					for (size_t i = 0; i < dist; i++)
						d[i] = src[i];
					d += dist;
					left -= (ptrdiff_t) dist;
					dist *= 2;
				}
				while (left > 0) {
					memcpy(d, src, 8);
					d += 8;
					src += 8;
					left -= 8;
				}
				op += length;
			}
			if (end_of_block)
				break;
			// One careful trip (the ends of the input and of the output)
			if (b.cnt < 0)
				return -1;
			b.refill();
			uint32_t e = decode(b, lit, kLitBits);
			if (e & kLiteral) {
				if (op == oend)
					return -1;
				*op++ = (uint8_t) (e >> 13);
				continue;
			}
			if (!(e & kValid))
				return -1;
			if (e & kEnd)
				break;
			const size_t length = ((e >> 13) & 0x7FFFu) + b.take((int) ((e >> 8) & 31u));
			e = decode(b, dist, kDistBits);
			if (!(e & kValid))
				return -1;
			const size_t offset = ((e >> 13) & 0x7FFFu) + b.take((int) ((e >> 8) & 31u));
			if (b.cnt < 0 || offset > (size_t) (op - out) || length > (size_t) (oend - op))
				return -1;
			const uint8_t *src = op - offset;
			for (size_t i = 0; i < length; i++)
				op[i] = src[i];
			op += length;
		}
		return 0;
	}
	CONGA_INFL_HD static inline void stored(uint8_t *out, uint8_t *op, const uint8_t *from, uint32_t len)
	{
		(void) out;
		if (len)
			memcpy(op, from, len);
	}
};

// The decoder proper; `dec` is scratch (tables) owned by the calling thread.
template <typename Symbols>
CONGA_INFL_HD inline bool inflate_block_stream_t(Decoder &dec, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
	Bits b;
	b.ip = in;
	b.iend = in + in_len;
	uint8_t *op = out, *const oend = out + out_len;

	for (;;) {
		if (b.cnt < 0)
			return false;
		b.refill();
		const uint32_t final_block = b.take(1);
		const uint32_t type = b.take(2);
		if (b.cnt < 0)
			return false;
		const uint32_t *lit = dec.lit, *dist = dec.dist;
		if (type == 0) {
			// stored: back to the byte stream (the whole bytes still in the bit buffer are handed back)
			b.drop(b.cnt & 7);
			b.ip -= b.cnt >> 3;
			b.buf = 0;
			b.cnt = 0;
			if (b.iend - b.ip < 4)
				return false;
			const uint32_t len = (uint32_t) b.ip[0] | ((uint32_t) b.ip[1] << 8), nlen = (uint32_t) b.ip[2] | ((uint32_t) b.ip[3] << 8);
			b.ip += 4;
			if ((len ^ nlen) != 0xFFFFu || (size_t) (b.iend - b.ip) < len || (size_t) (oend - op) < len)
				return false;
			Symbols::stored(out, op, b.ip, len);
			op += len;
			b.ip += len;
			if (final_block)
				break;
			continue;
		} else if (type == 1) {
			if (!dec.fixed_ready) {
				uint8_t lens[288];
				for (int i = 0; i < 288; i++)
					lens[i] = (uint8_t) (i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
				uint8_t dl[32];
				memset(dl, 5, sizeof dl);
				if (!build_table(lens, 288, kKindLitLen, kLitBits, dec.fixed_lit, 1 << kLitBits)
						|| !build_table(dl, 32, kKindDist, kDistBits, dec.fixed_dist, 1 << kDistBits))
					return false;
				dec.fixed_ready = true;
			}
			lit = dec.fixed_lit;
			dist = dec.fixed_dist;
		} else if (type == 2) {
			const int n_lit = (int) b.take(5) + 257, n_dist = (int) b.take(5) + 1, n_pre = (int) b.take(4) + 4;
			if (n_lit > 286 || n_dist > 30)
				return false;
			const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
			uint8_t pre_lens[19] = {0};
			b.refill();
			for (int i = 0; i < n_pre; i++) {
				if (b.cnt < 3)
					b.refill();
				pre_lens[kPreOrder[i]] = (uint8_t) b.take(3);
			}
			if (b.cnt < 0 || !build_table(pre_lens, 19, kKindPre, kPreBits, dec.pre, kPreSize))
				return false;
			uint8_t lens[286 + 30 + 138];
			int k = 0;
			while (k < n_lit + n_dist) {
				b.refill();
				const uint32_t e = decode(b, dec.pre, kPreBits);
				if (!(e & kValid) || b.cnt < 0)
					return false;
				const int sym = (int) ((e >> 13) & 0x7FFFu);
				if (sym < 16) {
					lens[k++] = (uint8_t) sym;
					continue;
				}
				int rep;
				uint8_t val = 0;
				if (sym == 16) {
					if (k == 0)
						return false;
					val = lens[k - 1];
					rep = 3 + (int) b.take(2);
				} else if (sym == 17) {
					rep = 3 + (int) b.take(3);
				} else {
					rep = 11 + (int) b.take(7);
				}
				if (b.cnt < 0 || k + rep > n_lit + n_dist)
					return false;
				memset(lens + k, val, (size_t) rep);
				k += rep;
			}
			if (lens[256] == 0) // no end-of-block code
				return false;
			if (!build_table(lens, n_lit, kKindLitLen, kLitBits, dec.lit, kLitSize)
					|| !build_table(lens + n_lit, n_dist, kKindDist, kDistBits, dec.dist, kDistSize))
				return false;
		} else {
			return false;
		}

		if (Symbols::run(b, lit, dist, out, op, oend) != 0)
			return false;
		if (final_block)
			break;
	}
	return b.cnt >= 0 && op == oend;
}


CONGA_INFL_HD inline bool inflate_block_stream(Decoder &dec, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
	return inflate_block_stream_t<SequentialSymbols>(dec, in, in_len, out, out_len);
}

} // namespace inflate_core
} // namespace conga_host
