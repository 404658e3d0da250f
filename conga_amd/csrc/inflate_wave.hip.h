// inflate_wave.hip.h -- BGZF inflate for gfx950: one BGZF block per WAVE, sixty-four symbols' worth of stream per trip.
//
// The producer side of count_reads_bam (bam_data.c:192-221) is htslib's BGZF reader in the reference (bam_data.c:253-259,
// 293,201); with conga_reads_bgzf the compressed blocks are inflated here.  Written against RFC 1951 for a wave64
// machine, not derived from any CPU inflater:
//
// * A Huffman stream is serial only in WHERE the next symbol starts.  WHAT a symbol is, given its first bit, needs
//   nothing but the tables.  So every trip lane i decodes the whole symbol that would start at bit `ibit + i` of the
//   stream -- literal / end of block, or a length with its extra bits, its distance code and that one's extra bits: at
//   most 48 bits, every lane has a 64-bit view -- with its own look-ups in the wave's tables in LDS.  The sixty-four
//   results sit in registers, and the true chain of symbol starts (0, 0 + bits[0], ...) is then followed with
//   v_readlane and scalar code: no memory in the serial part at all.  A trip retires EVERY symbol on the chain: a prefix
//   sum over the bytes they produce gives the literals their places (one masked byte store) and the matches their
//   destinations; short matches are copied by their own lanes all at once, the others one after the other by all lanes.
//   One trip to LDS per ~8 symbols instead of one per symbol.
// * Tables: 16-bit entries, two levels (9-bit root for literals / lengths, 7-bit for distances), built by the wave for
//   every deflate block: code lengths -> per-length ballots give every symbol its canonical code without a serial
//   pass; the width of each second-level table follows from the canonical ranges, so there is no atomic anywhere.
//   An entry's low bits ARE the distance from a symbol's first bit to whatever follows its literal / length part -- the whole
//   codeword length, of a length symbol with its extra bits, 64 or more for what ends a chain (LitFormat / DistFormat below):
//   the vector unit is what bounds this kernel, and that is what sixty-four lanes compute for the eight that count.
// * The product's symbol loop (run_symbols_batched) works in two phases: a trip only finds where symbols start -- 26 vector
//   instructions for its look-ups, the chain walked two symbols a step --, and sixty-four starts at a time become output with
//   every lane on a real symbol; matches of up to 64 bytes are copied by their own lanes.  (run_symbols is round 2's loop, kept
//   for comparison: every trip decodes its sixty-four candidates completely.)
// * The symbol loop is a function of its own (run_symbols_batched_call): no spill, whatever the set-up around it keeps.
// * The output goes to HBM (L2) as it is produced.  A match reads what the wave wrote before: stores are waited for
//   (s_waitcnt vmcnt(0)) only when the match reaches into bytes stored since the last wait, and the read-back loads
//   bypass the L1 (it is write-through and may hold a line from before the store).
// * The block's CRC32 is checked by the wave as well: sixty-four contiguous pieces, one per lane, byte-wise with the table
//   in LDS, combined by multiplying with x^(8 * bytes behind the piece) mod P (the CRC is linear over GF(2)).
//
// 4.4 KB of LDS per wave (tables + builder scratch) and a 1 KB CRC table per workgroup of four waves: eight workgroups
// per CU, 32 waves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/conga_hip.h"

namespace conga {
namespace iw {

constexpr int kLitRoot = 9, kDistRoot = 7, kPreRoot = 7;
// zlib's enough.c proves 852 entries sufficient for 286 symbols behind a 9-bit root and 592 for 30 behind a 6-bit one;
// a wider root needs no more second-level space than a narrower one.  The builder refuses what does not fit.
constexpr int kLitCap = 864, kDistCap = 128 + 528;
constexpr int kMaxLens = 320;

// Table entries are 16 bits and made so that a lane gets from the stream's bits to "where does the symbol behind mine start"
// in as few instructions as possible -- sixty-four lanes do that every trip, for the eight that turn out to be symbols.
// An entry holds the WHOLE length of its codeword, also behind a second-level table -- of a length symbol with the extra bits
// that follow it added in; the base values of lengths and distances and a length's count of extra bits (needed only for the
// lanes that turn out to be symbols) come from two small tables afterwards.
struct LitFormat {
	// [4:0] codeword length n1 -- of a length symbol: PLUS its count of extra bits (<= 15 + 5), which is what a lane needs to find
	// the distance code behind it; the count itself comes with the base value (Luts::len); [6:5] class: 0 literal / length symbol,
	// 1 pointer to a second-level table, 2 end of block, 3 hole; [7] length symbol; [15:8] literal byte / index of the length
	// symbol (0..28) / a pointer's (offset behind the root) / 2.  So `entry & 0x5F` is the distance from a symbol's first bit to
	// whatever follows its literal / length part, and 64 or more exactly for the two kinds of entry that end a chain.
	enum : uint32_t { kSub = 0x20, kEob = 0x40, kHoleTag = 0x60 };
	// kEob | n1: end of block; kSub | bits indexing the second-level table ([4:0]: the field width v_bfe_u32 takes from a
	// register as it is) | (its offset behind the root / 2) << 8;
	// kHoleTag: no codeword leads here (or one of the two that must not occur)
	static constexpr uint32_t hole = kHoleTag | 1u; // (n1 = 1: whatever stands at a position, the symbol behind it starts further on)
	__device__ static uint32_t entry(uint32_t sym, uint32_t len)
	{
		if (sym < 256u)
			return len | (sym << 8);
		if (sym == 256u)
			return len | kEob;
		const uint32_t i = sym - 257u;
		if (i > 28u)
			return hole; // 286, 287: in the fixed code, never in valid data
		const uint32_t eb = (i < 8u || i == 28u) ? 0u : (i >> 2) - 1u;
		return (len + eb) | 0x80u | (i << 8);
	}
	// (second-level tables have 2^sb >= 2 entries each and follow one another behind the root: every offset is even)
	__device__ static uint32_t pointer(uint32_t sb, uint32_t rel) { return sb | kSub | ((rel >> 1) << 8); }
	__device__ static uint32_t pointer_start(uint32_t ptr) { return (ptr >> 8) << 1; }
	__device__ static uint32_t pointer_bits(uint32_t ptr) { return ptr & 15u; }
};
struct DistFormat {
	// [15:14] = 0: [4:0] codeword length + extra bits, [8:5] extra bits, [13:9] distance symbol (0..29)
	// [15:14] = 1: [4:0] bits indexing the second-level table, [13:5] (its offset behind the root) / 2;  [15:14] = 2: no codeword
	// (a pointer's low five bits are the field width v_bfe_u32 takes from a register: no masking on the way to the second level)
	static constexpr uint32_t hole = 0x8000u;
	__device__ static uint32_t entry(uint32_t sym, uint32_t len)
	{
		if (sym > 29u)
			return hole; // 30, 31: in the fixed code, never in valid data
		const uint32_t deb = sym < 4u ? 0u : (sym >> 1) - 1u;
		return (len + deb) | (deb << 5) | (sym << 9);
	}
	// (second-level tables have 2^sb >= 2 entries each and follow one another behind the root: every offset is even)
	__device__ static uint32_t pointer(uint32_t sb, uint32_t rel) { return sb | ((rel >> 1) << 5) | 0x4000u; }
	__device__ static uint32_t pointer_start(uint32_t ptr) { return ((ptr >> 5) & 511u) << 1; }
	__device__ static uint32_t pointer_bits(uint32_t ptr) { return ptr & 31u; }
};

struct alignas(16) WaveLds {
	uint16_t lit[kLitCap];
	uint16_t dist[kDistCap];
	uint16_t sorted[288 + 32]; // builder: symbols in canonical order (by length, then value); symbol loop: the staged symbols' bits (65 x 8 bytes; offset 3040: 8-byte aligned)
	uint8_t lens[kMaxLens];    // code lengths of the block being set up (literal/length alphabet, then distances)
	uint8_t pre[1 << kPreRoot]; // code-length code: [2:0] bits, [7:3] symbol -- 5 + 3 bits are enough for 19 symbols of <= 7 bits
};

#ifdef IW_PROF
// tools/inflate_prof.hip: where a wave's time goes (s_memtime ticks per phase, summed per wave; lane 0 adds them up)
enum { P_VIEW = 0, P_WALK, P_LITS, P_MATCH_WAIT, P_MATCH_COPY, P_TABLES, P_CRC, P_TRIPS, P_MATCHES, P_WAITS, P_SYMS, P_SLOW, P_SLOW_BYTES, P_SLOW_LONG, P_SLOW_OVERLAP, P_SLOW_LE16, P_SLOW_LE32, P_SLOW_LE64, P_SLOW_SAFE32, P_N };
__device__ unsigned long long g_prof[P_N];
#define IW_T0() unsigned long long t_prof_ = __builtin_readcyclecounter()
#define IW_LAP(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); prof[k] += n_ - t_prof_; t_prof_ = n_; } while (0)
#define IW_ADD(k, v) (prof[k] += (v))
#else
#define IW_T0() do { } while (0)
#define IW_LAP(k) do { } while (0)
#define IW_ADD(k, v) do { } while (0)
#endif

__device__ __forceinline__ uint32_t uni(uint32_t v)
{
	return (uint32_t) __builtin_amdgcn_readfirstlane((int) v);
}

// LDS traffic between the lanes of ONE wave: the hardware runs a wave's LDS instructions in order; this keeps the
// compiler from moving them across.
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t lane_id()
{
	return threadIdx.x & 63u;
}

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t shift)
{
	return __builtin_amdgcn_alignbit(hi, lo, shift); // ({hi, lo} >> shift[4:0])[31:0]
}

// the stream and the output are in global memory, and the types say so: the symbol loop may be a function of its own, where
// nothing else would tell the compiler (generic pointers become flat_ instructions, which count on both wait counters)
typedef __attribute__((address_space(1))) uint8_t *GBytes;
typedef __attribute__((address_space(1))) const uint8_t *GConstBytes;
typedef __attribute__((address_space(1))) const uint32_t *GWords;

// bytes this wave stored are read back from L2: loads that do not stop at the (write-through) L1
template <typename P> __device__ __forceinline__ uint32_t load_written_u8(P p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename P> __device__ __forceinline__ uint32_t load_written_u32(P p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// eight bytes at ANY address (global memory takes unaligned dword accesses), past the L1 like the loads above
template <typename P> __device__ __forceinline__ uint64_t load_written_u64_unaligned(P p)
{
	uint64_t v;
	asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	return v;
}

// the types a lane copies a match of up to 64 bytes with (run_symbols_batched: loads at any address, past the L1; the stores are the
// compiler's -- unaligned dword stores, as everywhere here)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) uint64_t *GU64;
typedef __attribute__((address_space(1))) u32x4 *GU128;

// ---- the stream, read uniformly (block headers, code lengths) ---------------------------------------------------------
// 64 consecutive dwords of the input live in one register across the wave; a field is two v_readlane and a funnel shift.
struct Window {
	const uint32_t *in32; // 4-byte aligned base of the block's data
	uint32_t d0;          // first dword held
	uint32_t w;           // lane i holds in32[d0 + i]
	__device__ __forceinline__ void load(uint32_t dword)
	{
		d0 = dword;
		w = in32[d0 + lane_id()];
	}
	// the next n <= 25 bits at bit offset `ibit` (uniform)
	__device__ __forceinline__ uint32_t peek(uint32_t ibit, uint32_t n)
	{
		uint32_t d = (ibit >> 5) - d0;
		if (d >= 63u) { // (also when the position moved backwards: never happens)
			load(ibit >> 5);
			d = 0;
		}
		const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) w, (int) d);
		const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) w, (int) d + 1);
		return alignbit(hi, lo, ibit & 31u) & ((1u << n) - 1u);
	}
};

// ---- tables --------------------------------------------------------------------------------------------------------
// Canonical Huffman code of lens[0..n) (RFC 1951 3.2.2) as a two-level table indexed by the next bits of the stream,
// least significant bit first.  kind_of(sym) -> entry without the bit count.  false: over-subscribed, incomplete beyond
// what zlib lets pass (a single one-bit code), or more second-level space than `cap` holds.
template <int ROOT, typename Fmt>
__device__ __forceinline__ bool build_table(const uint8_t *lens, int n, uint16_t *table, int cap, uint16_t *sorted, bool allow_incomplete)
{
	const uint32_t lane = lane_id();
	// how many symbols of each length: one ballot per length and 64 symbols
	uint32_t count[16], first[16], offs[16];
#pragma unroll
	for (int L = 0; L < 16; L++)
		count[L] = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + (int) lane;
		const uint32_t len = i < n ? lens[i] : 0u;
#pragma unroll
		for (int L = 1; L < 16; L++)
			count[L] += (uint32_t) __popcll(__ballot(len == (uint32_t) L));
	}
	int left = 1, max_len = 0;
	uint32_t code = 0, at = 0;
	count[0] = 0;
#pragma unroll
	for (int L = 1; L < 16; L++) {
		left = (left << 1) - (int) count[L];
		if (left < 0)
			return false;
		if (count[L])
			max_len = L;
		code = (code + count[L - 1]) << 1;
		first[L] = code; // first canonical code of this length
		offs[L] = at;    // its place in `sorted`
		at += count[L];
	}
	if (left > 0 && max_len != 0 && (!allow_incomplete || max_len != 1))
		return false;
	// canonical order: a symbol's rank among those of its length is the number of equal-length symbols in front of it
	uint32_t next[16];
#pragma unroll
	for (int L = 1; L < 16; L++)
		next[L] = offs[L];
	for (int base = 0; base < n; base += 64) {
		const int i = base + (int) lane;
		const uint32_t len = i < n ? lens[i] : 0u;
#pragma unroll
		for (int L = 1; L < 16; L++) {
			const unsigned long long m = __ballot(len == (uint32_t) L);
			if (len == (uint32_t) L)
				sorted[next[L] + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u))] = (uint16_t) i;
			next[L] += (uint32_t) __popcll(m);
		}
	}
	constexpr int kRootSize = 1 << ROOT;
	// second-level tables: root slot p (stream order = bit-reversed prefix P) needs one iff a code longer than ROOT starts
	// with P; the codes of length L are the range [first[L], first[L] + count[L]), so its prefixes are a range too
	uint32_t my_sb[(kRootSize + 63) / 64], my_start[(kRootSize + 63) / 64];
	uint32_t mine = 0;
#pragma unroll
	for (int k = 0; k < (kRootSize + 63) / 64; k++) {
		const uint32_t p = lane * (uint32_t) ((kRootSize + 63) / 64) + (uint32_t) k; // consecutive slots per lane: the scan below is in slot order
		const uint32_t P = __brev(p) >> (32 - ROOT);
		uint32_t sb = 0;
#pragma unroll
		for (int L = ROOT + 1; L < 16; L++)
			if (count[L] && P >= (first[L] >> (L - ROOT)) && P <= ((first[L] + count[L] - 1u) >> (L - ROOT)))
				sb = (uint32_t) (L - ROOT);
		if (p >= (uint32_t) kRootSize)
			sb = 0;
		my_sb[k] = sb;
		my_start[k] = mine;
		mine += sb ? 1u << sb : 0u;
	}
	uint32_t incl = mine; // inclusive scan over the lanes
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t v = (uint32_t) __shfl_up((int) incl, o, 64);
		if (lane >= (uint32_t) o)
			incl += v;
	}
	const uint32_t total = uni((uint32_t) __shfl((int) incl, 63, 64));
	if ((uint32_t) kRootSize + total > (uint32_t) cap)
		return false;
	wave_sync();
	for (uint32_t j = lane; j < (uint32_t) kRootSize + total; j += 64)
		table[j] = (uint16_t) Fmt::hole;
	wave_sync();
#pragma unroll
	for (int k = 0; k < (kRootSize + 63) / 64; k++) {
		const uint32_t p = lane * (uint32_t) ((kRootSize + 63) / 64) + (uint32_t) k;
		if (my_sb[k])
			table[p] = (uint16_t) Fmt::pointer(my_sb[k], (incl - mine) + my_start[k]);
	}
	wave_sync();
	// every symbol fills the slots its codeword (and any bits behind it) leads to
	const uint32_t n_coded = at;
	for (uint32_t r = lane; r < n_coded; r += 64) {
		const uint32_t sym = sorted[r];
		const uint32_t len = lens[sym];
		// canonical code of rank r: first[len] + (r - offs[len]); both per-length arrays are indexed without a scratch array
		uint32_t f = 0, o = 0;
#pragma unroll
		for (int L = 1; L < 16; L++)
			if (len == (uint32_t) L) {
				f = first[L];
				o = offs[L];
			}
		const uint32_t rev = __brev(f + (r - o)) >> (32u - len);
		const uint32_t e = Fmt::entry(sym, len);
		if (len <= (uint32_t) ROOT) {
			for (uint32_t j = rev; j < (uint32_t) kRootSize; j += 1u << len)
				table[j] = (uint16_t) e;
		} else {
			const uint32_t ptr = table[rev & (uint32_t) (kRootSize - 1)];
			const uint32_t start = (uint32_t) kRootSize + Fmt::pointer_start(ptr), sb = Fmt::pointer_bits(ptr), l2 = len - (uint32_t) ROOT;
			for (uint32_t j = rev >> ROOT; j < (1u << sb); j += 1u << l2)
				table[start + j] = (uint16_t) e;
		}
	}
	wave_sync();
	return true;
}

// ---- one deflate block's symbols ---------------------------------------------------------------------------------------
struct Stream {
	GWords in32;          // 4-byte aligned
	uint32_t ibit;        // next bit (from in32)
	uint32_t end_bit;     // first bit behind the block's data
	GBytes out;
	uint32_t opos, out_len;
	uint32_t safe_pos;    // every byte below is known to have reached L2
};

// Base value and number of extra bits of the length symbols 257..285 (index 0..28; 29, 30: the fixed code's 286, 287,
// refused) and of the distance symbols 0..29 (30, 31: refused) -- RFC 1951 3.2.5 -- computed, not tabulated by hand.
struct Luts {
	uint16_t len[32];  // [8:0] base value of length symbol 257 + i, [14:12] its count of extra bits
	uint16_t dist[32]; // base value of distance symbol i
};

__device__ __forceinline__ void fill_luts(Luts &l, uint32_t i /* 0..31 */)
{
	const uint32_t eb = (i < 8u || i >= 28u) ? 0u : (i >> 2) - 1u;
	const uint32_t lbase = i < 8u ? 3u + i : i == 28u ? 258u : 3u + ((4u + (i & 3u)) << eb);
	l.len[i] = (uint16_t) (lbase | (eb << 12));
	const uint32_t deb = i < 4u ? 0u : (i >> 1) - 1u;
	const uint32_t dbase = i < 4u ? 1u + i : 1u + ((2u + (i & 1u)) << deb);
	l.dist[i] = (uint16_t) dbase; // (<= 24 577; the entries of symbols that do not exist are never asked for: their codewords are holes)
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
	return v + (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, ROW_MASK, 0xF, false);
}

// inclusive prefix sum over the wave: row_shr 1/2/4/8 inside each 16-lane row, row_bcast:15 / :31 across the rows
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
	v = dpp_add<0x111, 0xF>(v);
	v = dpp_add<0x112, 0xF>(v);
	v = dpp_add<0x114, 0xF>(v);
	v = dpp_add<0x118, 0xF>(v);
	v = dpp_add<0x142, 0xA>(v);
	v = dpp_add<0x143, 0xC>(v);
	return v;
}

// 0: end-of-block symbol reached; -1: the stream is invalid
__device__ __forceinline__ int run_symbols(Stream &s, const WaveLds &t, const Luts &luts
#ifdef IW_PROF
		, unsigned long long *prof
#endif
)
{
	const uint32_t lane = lane_id();
	// the stream's state is the same in every lane: kept in scalar registers (readfirstlane tells the compiler)
	uint32_t ibit = uni(s.ibit), opos = uni(s.opos), safe_pos = uni(s.safe_pos);
	const uint32_t end_bit = uni(s.end_bit), out_len = uni(s.out_len);
	const GWords in32 = s.in32;
	auto leave = [&](int rc) {
		s.ibit = ibit;
		s.opos = opos;
		s.safe_pos = safe_pos;
		return rc;
	};
	// 64 consecutive dwords of the input sit in one register across the wave (lane i: dword d0 + i); a lane's view is
	// three ds_bpermute (the LDS crossbar, no memory) and two funnel shifts, and the register is refilled with one
	// coalesced load every ~1900 bits
	// (loaded and waited for in one piece of assembly: a load the compiler tracks would make it wait for everything in
	// flight at the top of every trip, because the register MAY have been refilled)
	auto load_window = [&](uint32_t dword) {
		uint32_t w;
		const GWords at = in32 + dword + lane;
		asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(at) : "memory");
		return w;
	};
	uint32_t d0 = uni(ibit >> 5);
	uint32_t wreg = load_window(d0);
	// n <= 8 bytes of v to out[to..): dword / short / byte pieces (unaligned dword stores are fine in global memory)
	// 3 <= n <= 8 bytes of v8 to out[to..): the first four and the last four as dwords (they may overlap: the same bytes), or a
	// short and a byte (unaligned dword stores are fine in global memory)
	auto store_pieces = [&](uint64_t v8, uint32_t to, uint32_t n) {
		const GBytes q = s.out + to;
		if (n >= 4u) {
			*(__attribute__((address_space(1))) uint32_t *) q = (uint32_t) v8;
			*(__attribute__((address_space(1))) uint32_t *) (q + (n - 4u)) = (uint32_t) (v8 >> (8u * (n - 4u)));
		} else {
			*(__attribute__((address_space(1))) uint16_t *) q = (uint16_t) v8;
			q[2] = (uint8_t) (v8 >> 16);
		}
	};
	for (;;) {
		IW_T0();
		if (ibit > end_bit)
			return leave(-1);
		uint32_t dd = (ibit >> 5) - d0;
		if (dd > 59u) {
			d0 = ibit >> 5;
			wreg = load_window(d0);
			dd = 0;
		}
		// this lane's 64-bit view of the stream from bit ibit + lane on
		const uint32_t b = (ibit & 31u) + lane;
		const uint32_t at = (dd + (b >> 5)) << 2;
		const uint32_t w0 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) at, (int) wreg);
		const uint32_t w1 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) at + 4, (int) wreg);
		const uint32_t w2 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) at + 8, (int) wreg);
		const uint32_t sh = b & 31u;
		const uint32_t lo = alignbit(w1, w0, sh), hi = alignbit(w2, w1, sh);
		// the literal / length code that would start here (second level read by every lane: some lane nearly always needs it.
		// A lane whose root entry is no pointer forms its second index from that entry's other fields: it reads some other
		// halfword of LDS -- of this wave's tables, of a neighbour's, or past the workgroup's allocation, where the hardware
		// returns 0 -- and drops it.  Deliberate: a clamp would be two more vector instructions per trip for nothing.)
		const uint32_t e1 = t.lit[lo & ((1u << kLitRoot) - 1u)];
		const bool sub1 = (e1 & 0xF0u) == LitFormat::kSub;
		const uint32_t e2 = t.lit[(1u << kLitRoot) + ((e1 >> 8) << 1) + __builtin_amdgcn_ubfe(lo, (uint32_t) kLitRoot, e1)];
		const uint32_t e = sub1 ? e2 : e1;
		const uint32_t n1eb = e & 31u; // the codeword's length, of a length symbol with its extra bits
		const bool is_match = (e & 0x80u) != 0u;
		// ... as a length: its extra bits, the distance code behind them and that one's extra bits (<= 15 + 13 bits from bit n1 + eb <= 20)
		const uint32_t r2 = alignbit(hi, lo, n1eb);
		const uint32_t d1 = t.dist[r2 & ((1u << kDistRoot) - 1u)];
		const bool subd = (d1 & 0xC000u) == 0x4000u;
		const uint32_t d2 = t.dist[(1u << kDistRoot) + (((d1 >> 5) & 511u) << 1) + __builtin_amdgcn_ubfe(r2, (uint32_t) kDistRoot, d1)];
		const uint32_t ed = subd ? d2 : d1;
		// base values: asked for now, needed behind the walk
		const uint32_t lb = luts.len[(e >> 8) & 31u];
		const uint32_t dbase = luts.dist[(ed >> 9) & 31u];
		// (never 0: a codeword has a length, and a hole is given one -- LitFormat::hole; the walk below must move)
		const unsigned long long is_match_m = __ballot(is_match);
		const uint32_t bits = is_match ? n1eb + (ed & 31u) : n1eb;
		const uint32_t nxt = lane + bits; // where the symbol behind this one starts
		IW_LAP(P_VIEW);

		// The chain of true symbol starts, from lane 0 on: J = "start of the next symbol" is doubled (J <- J o J) while the
		// lanes known to be starts mark the lane their J points at (ds_permute pushes a flag there): after round k the first
		// 2^(k+1) starts are known.  An end-of-block or invalid symbol ends the chain.
		const uint32_t cls = e & 0xF0u;
		const unsigned long long eob_m = __ballot(cls == LitFormat::kEob), hole_m = __ballot(cls == LitFormat::kHoleTag);
		// one v_readlane and four scalar instructions per symbol: the vector unit, which is what the waves of a SIMD
		// compete for, sees one instruction per symbol.  (Pointer doubling over the LDS crossbar -- J <- J o J while the known
		// starts push a flag to the lane their J names -- needs four rounds of nine vector instructions for the same.)
		unsigned long long chain = 0;
		{
			uint32_t cur = 0;
			do {
				chain |= 1ull << cur;
				cur = (uint32_t) __builtin_amdgcn_readlane((int) nxt, (int) cur);
			} while (cur < 64u);
		}
		// an end-of-block symbol or a hole ends the chain where it stands (the walk went on behind it: cut)
		{
			const unsigned long long stop = chain & (eob_m | hole_m);
			if (stop)
				chain &= (2ull << __builtin_ctzll(stop)) - 1ull;
		}
		const bool on_chain = __builtin_amdgcn_inverse_ballot_w64(chain);
		// a hole, or a length whose distance is one: the stream is invalid (a hole ends the chain, a bad distance is met on it)
		const unsigned long long match_m = chain & is_match_m;
		if ((chain & hole_m) | (match_m & __ballot((ed & 0x8000u) != 0u)))
			return leave(-1);
		const bool ends = (chain & eob_m) != 0ull;
		const bool is_lit = cls == 0u;
		const uint32_t val = e >> 8;
		const uint32_t deb = (ed >> 5) & 15u;
		const uint32_t dist = dbase + ((r2 >> ((ed & 31u) - deb)) & ((1u << deb) - 1u));
		const bool is_len = __builtin_amdgcn_inverse_ballot_w64(is_match_m); // (the predicate behind the walk: straight from the mask)
		const uint32_t eb = lb >> 12, lbase = lb & 511u;
		const uint32_t produced = is_len ? lbase + ((lo >> (n1eb - eb)) & ((1u << eb) - 1u)) : is_lit ? 1u : 0u;
		IW_LAP(P_WALK);
		// where every start's bytes go
		const uint32_t mine = on_chain ? produced : 0u;
		const uint32_t incl = wave_incl_scan(mine);
		const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
		const uint32_t last = 63u - (uint32_t) __builtin_clzll(chain);
		const uint32_t advance = (uint32_t) __builtin_amdgcn_readlane((int) nxt, (int) last);
		IW_ADD(P_TRIPS, 1);
		IW_ADD(P_SYMS, (uint32_t) __popcll(chain));
		if (opos + total > out_len)
			return leave(-1);
		if (on_chain && is_lit)
			s.out[opos + incl - 1u] = (uint8_t) val;
		IW_LAP(P_LITS);
		// Short matches whose source lies wholly in bytes known to be in L2 are copied by their own lanes, all at once: one
		// 8-byte load each (any alignment), then the bytes stored as dword / short / byte pieces.  At zlib's fast levels these
		// are most matches (3..8 bytes, found by hash anywhere in the 32 KB behind).
		unsigned long long mm = match_m;
		{
			const uint32_t to_l = opos + incl - produced, src_l = to_l - dist;
			// (dist <= to_l is tested on its own: a distance that reaches up to eight bytes in front of the output makes src_l wrap
			// and src_l + produced wrap back below safe_pos -- damaged streams do that, tests/test_gpu_inflate.py's fuzz found it;
			// the loop below refuses such a match)
			const bool fast = on_chain && is_len && produced <= 8u && dist >= produced && dist <= to_l && src_l + produced <= safe_pos;
			const unsigned long long fast_m = __ballot(fast);
			if (fast_m) {
				if (fast)
					store_pieces(load_written_u64_unaligned(s.out + src_l), to_l, produced);
				IW_ADD(P_MATCHES, (uint32_t) __popcll(fast_m));
				mm &= ~fast_m;
			}
		}
		IW_LAP(P_MATCH_COPY);
		while (mm) {
			const uint32_t i = (uint32_t) __builtin_ctzll(mm);
			mm &= mm - 1ull;
			const uint32_t len = (uint32_t) __builtin_amdgcn_readlane((int) produced, (int) i);
			const uint32_t d = (uint32_t) __builtin_amdgcn_readlane((int) dist, (int) i);
			const uint32_t to = opos + (uint32_t) __builtin_amdgcn_readlane((int) incl, (int) i) - len;
			if (d > to)
				return leave(-1);
			const uint32_t src = to - d;
			if (src + (d < len ? d : len) > safe_pos) { // reaches into bytes stored since the last wait
				asm volatile("" ::: "memory");
				__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the stores are acknowledged by L2
				asm volatile("" ::: "memory");
				safe_pos = to;
				IW_ADD(P_WAITS, 1);
			}
			IW_LAP(P_MATCH_WAIT);
			IW_ADD(P_MATCHES, 1);
			IW_ADD(P_SLOW, 1);
			IW_ADD(P_SLOW_BYTES, len);
			IW_ADD(P_SLOW_LONG, len > 8u ? 1 : 0);
			IW_ADD(P_SLOW_OVERLAP, d < len ? 1 : 0);
			IW_ADD(P_SLOW_LE16, len > 8u && len <= 16u ? 1 : 0);
			IW_ADD(P_SLOW_LE32, len > 16u && len <= 32u ? 1 : 0);
			IW_ADD(P_SLOW_LE64, len > 32u && len <= 64u ? 1 : 0);
			IW_ADD(P_SLOW_SAFE32, len > 8u && len <= 32u && d >= len && src + len <= opos ? 1 : 0);
			if (d >= len) {
				for (uint32_t k = lane; k < len; k += 64u)
					s.out[to + k] = (uint8_t) load_written_u8(s.out + src + k);
			} else { // the match overlaps itself: it repeats its first d bytes
				const float inv = 1.0f / (float) d;
				for (uint32_t k = lane; k < len; k += 64u) {
					uint32_t q = (uint32_t) ((float) k * inv);
					int32_t r = (int32_t) (k - q * d);
					if (r < 0)
						r += (int32_t) d;
					else if ((uint32_t) r >= d)
						r -= (int32_t) d;
					s.out[to + k] = (uint8_t) load_written_u8(s.out + src + (uint32_t) r);
				}
			}
			IW_LAP(P_MATCH_COPY);
		}
		opos += total;
		ibit += advance;
		if (ends)
			return leave(ibit > end_bit ? -1 : 0);
	}
}

// ---- the same loop in two phases (round 3) -----------------------------------------------------------------------------
// A trip of run_symbols() decodes sixty-four candidate symbols completely -- values, base tables, extra bits, the prefix sum
// of what they produce, the stores -- for the ~8 that turn out to lie on the chain: the vector unit, which bounds the kernel,
// spends seven eighths of that on lanes that are dropped.  Here a trip only finds out WHERE symbols start (the look-ups give
// every candidate its length in bits, the scalar walk follows the chain) and stages the sixty-four bits every start saw (LDS);
// once sixty-four are staged -- eight trips or so -- every lane takes ONE real symbol's bits back, decodes it for good and the
// whole wave produces output for sixty-four symbols at once.  Per trip ~50 vector instructions instead of 112, per batch ~100 more.
// Matches: everything stored before a batch is in L2 when it starts (one s_waitcnt per batch, long satisfied); a match of up to
// 64 bytes that does not repeat itself and whose source lies in front of the batch's own output is copied by its lane -- the
// loads of all of them in flight before one wait --; the others -- a match that reaches into the batch's own bytes, a longer
// one, one that repeats itself -- go one after the other behind a wait for the batch's stores.
__device__ __forceinline__ int run_symbols_batched(Stream &s, const WaveLds &t, const Luts &luts, uint32_t *stage /* LDS: sixty-five times eight bytes, 8-byte aligned */)
{
	const uint32_t lane = lane_id();
	uint32_t ibit = uni(s.ibit), opos = uni(s.opos), safe_pos = uni(s.safe_pos);
	const uint32_t end_bit = uni(s.end_bit), out_len = uni(s.out_len);
	const GWords in32 = s.in32;
	uint32_t staged = 0; // starts noted and not yet turned into output
	auto leave = [&](int rc) {
		s.ibit = ibit;
		s.opos = opos;
		s.safe_pos = safe_pos;
		return rc;
	};
	auto load_window = [&](uint32_t dword) {
		uint32_t w;
		const GWords at = in32 + dword + lane;
		asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(at) : "memory");
		return w;
	};
	auto store_pieces = [&](uint64_t v8, uint32_t to, uint32_t n) {
		const GBytes q = s.out + to;
		if (n >= 4u) {
			*(__attribute__((address_space(1))) uint32_t *) q = (uint32_t) v8;
			*(__attribute__((address_space(1))) uint32_t *) (q + (n - 4u)) = (uint32_t) (v8 >> (8u * (n - 4u)));
		} else {
			*(__attribute__((address_space(1))) uint16_t *) q = (uint16_t) v8;
			q[2] = (uint8_t) (v8 >> 16);
		}
	};
	// phase 2: the staged starts become output.  false: the stream is invalid (more output than the block holds, a match
	// that reaches in front of the output)
	auto flush = [&]() -> bool {
		const uint32_t n = staged;
		staged = 0;
		if (n == 0u)
			return true;
		wave_sync();
		const bool have = lane < n;
		// the symbol's sixty-four bits as the trip that found it saw them (48 are needed): staged in LDS, no trip to memory for them
		const uint64_t v = have ? reinterpret_cast<const uint64_t *>(stage)[lane] : 0ull;
		asm volatile("" ::: "memory");
		__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): everything stored before the batch has arrived (long since, as a rule)
		asm volatile("" ::: "memory");
		safe_pos = opos;
		const uint32_t lo = (uint32_t) v;
		const uint32_t e1 = t.lit[lo & ((1u << kLitRoot) - 1u)];
		const bool sub1 = (e1 & 0xF0u) == LitFormat::kSub;
		const uint32_t e2 = t.lit[(1u << kLitRoot) + ((e1 >> 8) << 1) + __builtin_amdgcn_ubfe(lo, (uint32_t) kLitRoot, e1)];
		const uint32_t e = sub1 ? e2 : e1;
		const uint32_t n1eb = e & 31u;
		const bool is_len = have && (e & 0x80u) != 0u;
		const uint32_t r2 = (uint32_t) (v >> n1eb);
		const uint32_t d1 = t.dist[r2 & ((1u << kDistRoot) - 1u)];
		const bool subd = (d1 & 0xC000u) == 0x4000u;
		const uint32_t d2 = t.dist[(1u << kDistRoot) + (((d1 >> 5) & 511u) << 1) + __builtin_amdgcn_ubfe(r2, (uint32_t) kDistRoot, d1)];
		const uint32_t ed = subd ? d2 : d1;
		const uint32_t lb = luts.len[(e >> 8) & 31u];
		const uint32_t dbase = luts.dist[(ed >> 9) & 31u];
		const uint32_t deb = (ed >> 5) & 15u;
		const uint32_t dist = dbase + ((r2 >> ((ed & 31u) - deb)) & ((1u << deb) - 1u));
		const uint32_t eb = lb >> 12, lbase = lb & 511u;
		const uint32_t produced = !have ? 0u : is_len ? lbase + ((lo >> (n1eb - eb)) & ((1u << eb) - 1u)) : 1u;
		const uint32_t incl = wave_incl_scan(produced);
		const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
		// more output than the block holds; a length whose distance code is a hole (asked here, of real symbols only)
		if (opos + total > out_len || __ballot(is_len && (ed & 0x8000u) != 0u) != 0ull)
			return false;
		if (have && !is_len)
			s.out[opos + incl - 1u] = (uint8_t) (e >> 8);
		unsigned long long mm = __ballot(is_len);
		{
			const uint32_t to_l = opos + incl - produced, src_l = to_l - dist;
			// (dist <= to_l on its own: see run_symbols)
			// a match that does not repeat itself and whose source lies in front of the batch's output is copied by its own lane
			const bool own = is_len && dist >= produced && dist <= to_l && src_l + produced <= safe_pos;
			// ... up to 64 bytes long (with run-structured qualities seven in ten of the output bytes come from matches of more than
			// eight): four sizes, each with loads of its own -- the first and the last 8, 16 or 32 bytes of the match, the pieces
			// overlap in the middle, nothing outside the match is touched -- and ALL of them in flight before the one wait: a
			// batch pays one trip to L2 for its matches, not one per size.
			const unsigned long long own_m = __builtin_amdgcn_ballot_w64(own && produced <= 64u);
			if (own_m) {
				const bool c8 = own && produced <= 8u, c16 = own && produced - 9u < 8u, c32 = own && produced - 17u < 16u, c64 = own && produced - 33u < 32u;
				const GBytes from = s.out + src_l, to = s.out + to_l;
				uint64_t q0, q1;
				u32x4 x0, x1, x2, x3;
				if (c8)
					asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=&v"(q0) : "v"(from) : "memory");
				if (c16)
					asm volatile("global_load_dwordx2 %0, %2, off sc0 sc1\n\tglobal_load_dwordx2 %1, %3, off sc0 sc1"
							: "=&v"(q0), "=&v"(q1) : "v"(from), "v"(from + (produced - 8u)) : "memory");
				if (c32)
					asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1"
							: "=&v"(x0), "=&v"(x1) : "v"(from), "v"(from + (produced - 16u)) : "memory");
				if (c64)
					asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc0 sc1\n\t"
							"global_load_dwordx4 %2, %5, off sc0 sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc0 sc1"
							: "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(from), "v"(from + (produced - 32u)) : "memory");
				// (the registers go through the wait: nothing that reads them can be moved in front of it)
				asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : : "memory");
				if (c8)
					store_pieces(q0, to_l, produced);
				if (c16) {
					*(GU64) to = q0;
					*(GU64) (to + (produced - 8u)) = q1;
				}
				if (c32) {
					*(GU128) to = x0;
					*(GU128) (to + (produced - 16u)) = x1;
				}
				if (c64) {
					*(GU128) to = x0;
					*(GU128) (to + 16u) = x1;
					*(GU128) (to + (produced - 32u)) = x2;
					*(GU128) (to + (produced - 16u)) = x3;
				}
				mm &= ~own_m;
			}
		}
		while (mm) {
			const uint32_t i = (uint32_t) __builtin_ctzll(mm);
			mm &= mm - 1ull;
			const uint32_t len = (uint32_t) __builtin_amdgcn_readlane((int) produced, (int) i);
			const uint32_t d = (uint32_t) __builtin_amdgcn_readlane((int) dist, (int) i);
			const uint32_t to = opos + (uint32_t) __builtin_amdgcn_readlane((int) incl, (int) i) - len;
			if (d > to)
				return false;
			const uint32_t src = to - d;
			if (src + (d < len ? d : len) > safe_pos) { // reaches into bytes stored since the last wait
				asm volatile("" ::: "memory");
				__builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the stores are acknowledged by L2
				asm volatile("" ::: "memory");
				safe_pos = to;
			}
			if (d >= len) {
				for (uint32_t k = lane; k < len; k += 64u)
					s.out[to + k] = (uint8_t) load_written_u8(s.out + src + k);
			} else { // the match overlaps itself: it repeats its first d bytes
				const float inv = 1.0f / (float) d;
				for (uint32_t k = lane; k < len; k += 64u) {
					uint32_t q = (uint32_t) ((float) k * inv);
					int32_t r = (int32_t) (k - q * d);
					if (r < 0)
						r += (int32_t) d;
					else if ((uint32_t) r >= d)
						r -= (int32_t) d;
					s.out[to + k] = (uint8_t) load_written_u8(s.out + src + (uint32_t) r);
				}
			}
		}
		opos += total;
		return true;
	};
	uint32_t d0 = uni(ibit >> 5);
	uint32_t wreg = load_window(d0);
	for (;;) {
		if (ibit > end_bit)
			return leave(-1);
		uint32_t dd = (ibit >> 5) - d0;
		if (dd > 59u) {
			d0 = ibit >> 5;
			wreg = load_window(d0);
			dd = 0;
		}
		// phase 1: how long is the symbol that would start at bit ibit + lane
		const uint32_t b = (ibit & 31u) + lane;
		const uint32_t at = (dd + (b >> 5)) << 2;
		const uint32_t w0 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) at, (int) wreg);
		const uint32_t w1 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) at + 4, (int) wreg);
		const uint32_t w2 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) at + 8, (int) wreg);
		const uint32_t sh = b & 31u;
		const uint32_t lo = alignbit(w1, w0, sh), hi = alignbit(w2, w1, sh); // (the upper half: for the rare second level below, and staged with the lower one)
		const uint32_t e1 = t.lit[lo & ((1u << kLitRoot) - 1u)];
		const bool sub1 = (e1 & 0xF0u) == LitFormat::kSub;
		const uint32_t e2 = t.lit[(1u << kLitRoot) + ((e1 >> 8) << 1) + __builtin_amdgcn_ubfe(lo, (uint32_t) kLitRoot, e1)];
		const uint32_t e = sub1 ? e2 : e1;
		const uint32_t is_len = (e >> 7) & 1u;
		const unsigned long long match_m = __builtin_amdgcn_ballot_w64(is_len != 0u); // the candidates that are length symbols
		// from the symbol's first bit to what follows its literal / length part; 64 or more for an end-of-block symbol and for a
		// hole (LitFormat): their "next start" lies outside the window, so the walk stops there by itself, and what stood there
		// is asked of that one lane afterwards -- no ballots, no cutting of masks.
		uint32_t bits = e & 0x5Fu;
		// (v_bfe_u32 takes its offset from the entry's low five bits as they are; <= 20 where it matters: seven bits from there lie in `lo`)
		const uint32_t d1 = t.dist[__builtin_amdgcn_ubfe(lo, e, (uint32_t) kDistRoot)];
		uint32_t dbits = d1 & 31u; // the distance code with its extra bits
		// (the second level of the distance code only in trips where a candidate length symbol leads to one: distance codes of
		// more than seven bits are the rare small distances; the view's upper half is fetched for those trips alone)
		const unsigned long long subd_m = match_m & __builtin_amdgcn_ballot_w64((d1 & 0xC000u) == 0x4000u);
		if (subd_m != 0ull) {
			const uint32_t r2 = alignbit(hi, lo, bits);
			const uint32_t d2 = t.dist[(1u << kDistRoot) + (((d1 >> 5) & 511u) << 1) + __builtin_amdgcn_ubfe(r2, (uint32_t) kDistRoot, d1)];
			dbits = __builtin_amdgcn_inverse_ballot_w64(subd_m) ? d2 & 31u : dbits;
		}
		bits += __umul24(dbits, is_len); // (one v_mad_u32_u24)
		const uint32_t nxt = min(lane + bits, 64u);
		// The chain of symbol starts by a scalar walk, two symbols a step: every lane also knows where the symbol after next starts
		// (one ds_bpermute), so a step is two v_readlane with the same lane select -- the four wait states between a select's
		// write and its use are paid once per two symbols -- and two s_bitset1.  A start "at 64" sets bit 0, which is set anyway.
		unsigned long long chain = 1ull;
		{
			const uint32_t nxt2 = max((uint32_t) __builtin_amdgcn_ds_bpermute((int) (nxt << 2), (int) nxt), nxt); // (nxt = 64 reads lane 0: dropped by the max)
			uint32_t cur = 0;
			do {
				const uint32_t n_a = (uint32_t) __builtin_amdgcn_readlane((int) nxt, (int) cur);
				const uint32_t n_b = (uint32_t) __builtin_amdgcn_readlane((int) nxt2, (int) cur);
				asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(n_a));
				asm("s_bitset1_b64 %0, %1" : "+s"(chain) : "s"(n_b));
				cur = n_b;
			} while (cur < 64u);
		}
		const uint32_t last = 63u - (uint32_t) __builtin_clzll(chain);
		const uint32_t bits_last = (uint32_t) __builtin_amdgcn_readlane((int) bits, (int) last);
		const uint32_t ends = bits_last >> 6; // 1: an end-of-block symbol or a hole stands there (a symbol has fewer than 64 bits)
		if (ends && ((uint32_t) __builtin_amdgcn_readlane((int) e, (int) last) & 0xF0u) == LitFormat::kHoleTag)
			return leave(-1); // no codeword leads here (or one that must not occur)
		// every start notes its bit position; the end-of-block symbol, last on its chain, lands in the slot behind the others
		// and is not counted (the stage has room for sixty-five)
		const uint32_t n_new = (uint32_t) __popcll(chain) - ends;
		if (staged + n_new > 64u && !flush())
			return leave(-1);
		if (__builtin_amdgcn_inverse_ballot_w64(chain))
			reinterpret_cast<uint64_t *>(stage)[staged + __builtin_amdgcn_mbcnt_hi((uint32_t) (chain >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) chain, 0u))]
					= (uint64_t) lo | ((uint64_t) hi << 32);
		staged += n_new;
		ibit += last + (bits_last & 63u);
		if (ends) {
			if (!flush())
				return leave(-1);
			return leave(ibit > end_bit ? -1 : 0);
		}
	}
}

#ifndef IW_PROF
// The symbol loop as a function of its own: its registers are then allocated for the loop alone, not together with
// everything a block's set-up keeps alive around it.  Arguments and results travel in registers; the tables are named by
// their LDS addresses, so that the accesses stay ds_ instructions.
struct SymbolsOut {
	uint32_t ibit, opos, safe_pos;
	int rc;
};
typedef __attribute__((address_space(3))) const WaveLds *LdsTables;
typedef __attribute__((address_space(3))) const Luts *LdsLuts;

__device__ __attribute__((noinline)) SymbolsOut run_symbols_call(GWords in32, GBytes out, uint32_t ibit, uint32_t end_bit, uint32_t opos,
		uint32_t out_len, uint32_t safe_pos, uint32_t tables_at, uint32_t luts_at)
{
	Stream s;
	const uint64_t in_u = (uint64_t) uni((uint32_t) (uintptr_t) in32) | ((uint64_t) uni((uint32_t) ((uintptr_t) in32 >> 32)) << 32);
	const uint64_t out_u = (uint64_t) uni((uint32_t) (uintptr_t) out) | ((uint64_t) uni((uint32_t) ((uintptr_t) out >> 32)) << 32);
	s.in32 = (GWords) in_u;
	s.out = (GBytes) out_u;
	s.ibit = ibit;
	s.end_bit = end_bit;
	s.opos = opos;
	s.out_len = out_len;
	s.safe_pos = safe_pos;
	const WaveLds &t = *(const WaveLds *) (LdsTables) (uintptr_t) uni(tables_at);
	const Luts &luts = *(const Luts *) (LdsLuts) (uintptr_t) uni(luts_at);
	SymbolsOut o;
	o.rc = run_symbols(s, t, luts);
	o.ibit = s.ibit;
	o.opos = s.opos;
	o.safe_pos = s.safe_pos;
	return o;
}

__device__ __attribute__((noinline)) SymbolsOut run_symbols_batched_call(GWords in32, GBytes out, uint32_t ibit, uint32_t end_bit, uint32_t opos,
		uint32_t out_len, uint32_t safe_pos, uint32_t tables_at, uint32_t luts_at)
{
	Stream s;
	const uint64_t in_u = (uint64_t) uni((uint32_t) (uintptr_t) in32) | ((uint64_t) uni((uint32_t) ((uintptr_t) in32 >> 32)) << 32);
	const uint64_t out_u = (uint64_t) uni((uint32_t) (uintptr_t) out) | ((uint64_t) uni((uint32_t) ((uintptr_t) out >> 32)) << 32);
	s.in32 = (GWords) in_u;
	s.out = (GBytes) out_u;
	s.ibit = ibit;
	s.end_bit = end_bit;
	s.opos = opos;
	s.out_len = out_len;
	s.safe_pos = safe_pos;
	WaveLds &t = *(WaveLds *) (__attribute__((address_space(3))) WaveLds *) (uintptr_t) uni(tables_at);
	const Luts &luts = *(const Luts *) (LdsLuts) (uintptr_t) uni(luts_at);
	SymbolsOut o;
	o.rc = run_symbols_batched(s, t, luts, reinterpret_cast<uint32_t *>(t.sorted));
	o.ibit = s.ibit;
	o.opos = s.opos;
	o.safe_pos = s.safe_pos;
	return o;
}

#endif

// ---- CRC-32 (reflected, polynomial 0xEDB88320) over GF(2) ---------------------------------------------------------------
// a(x) * b(x) mod P in the reflected representation (bit 31 = x^0)
__device__ __forceinline__ uint32_t gf_mul(uint32_t a, uint32_t b)
{
	uint32_t p = 0;
#pragma unroll 8
	for (int k = 0; k < 32; k++) {
		p ^= b & (0u - ((a >> (31 - k)) & 1u));
		b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1u)));
	}
	return p;
}

// x^(8 n) mod P; x2n[k] = x^(2^k) mod P
__device__ __forceinline__ uint32_t gf_x8n(uint32_t n_bytes, const uint32_t *x2n)
{
	uint32_t p = 0x80000000u; // 1
	uint32_t n = n_bytes;
	for (int k = 3; n; k++, n >>= 1)
		if (n & 1u)
			p = gf_mul(x2n[k], p);
	return p;
}

// CRC32 of out[0..n) by the whole wave (every lane gets the result)
__device__ __forceinline__ uint32_t wave_crc32(const uint8_t *out, uint32_t n, const uint32_t *crc_table /* LDS */, const uint32_t *x2n)
{
	const uint32_t lane = lane_id();
	const uint32_t piece = ((n + 63u) / 64u + 3u) & ~3u;
	const uint32_t lo = min(lane * piece, n), hi = min(lo + piece, n);
	uint32_t c = lane == 0u ? 0xFFFFFFFFu : 0u; // the register's start value travels with the first piece
	uint32_t i = lo;
	for (; i < hi && ((uintptr_t) (out + i) & 3u); i++)
		c = crc_table[(c ^ load_written_u8(out + i)) & 0xFFu] ^ (c >> 8);
	for (; i + 4u <= hi; i += 4u) {
		const uint32_t w = load_written_u32(reinterpret_cast<const uint32_t *>(out + i));
		c ^= w;
#pragma unroll
		for (int k = 0; k < 4; k++)
			c = crc_table[c & 0xFFu] ^ (c >> 8);
	}
	for (; i < hi; i++)
		c = crc_table[(c ^ load_written_u8(out + i)) & 0xFFu] ^ (c >> 8);
	// shift every piece's register over the bytes behind it and add up
	c = gf_mul(c, gf_x8n(n - hi, x2n));
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		c ^= (uint32_t) __shfl_xor((int) c, o, 64);
	return c ^ 0xFFFFFFFFu;
}

enum { kStatusOk = 0, kStatusRefused = 1, kStatusCrc = 2 };

// One whole BGZF block (a raw deflate stream of one or more deflate blocks) by one wave.  BATCHED: run_symbols_batched.
template <bool BATCHED> __device__ __forceinline__ int inflate_block(WaveLds &t, const Luts &luts, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len
#ifdef IW_PROF
		, unsigned long long *prof
#endif
)
{
	const uint32_t lane = lane_id();
	Stream s;
	const uint32_t mis = (uint32_t) ((uintptr_t) in & 3u);
	s.in32 = (GWords) (in - mis);
	s.ibit = mis * 8u;
	s.end_bit = (mis + in_len) * 8u;
	s.out = (GBytes) out;
	s.opos = 0;
	s.out_len = out_len;
	s.safe_pos = 0;
	Window win;
	win.in32 = reinterpret_cast<const uint32_t *>(in - mis);
	win.load(0);
	for (;;) {
		if (s.ibit + 3u > s.end_bit)
			return kStatusRefused;
		const uint32_t hdr = win.peek(s.ibit, 3);
		s.ibit += 3;
		const uint32_t final_block = hdr & 1u, type = hdr >> 1;
		if (type == 0u) {
			// stored: LEN / NLEN at the next byte boundary, then LEN bytes as they are
			s.ibit = (s.ibit + 7u) & ~7u;
			if (s.ibit + 32u > s.end_bit)
				return kStatusRefused;
			const uint32_t ln = win.peek(s.ibit, 16), nl = win.peek(s.ibit + 16u, 16);
			s.ibit += 32u;
			if ((ln ^ nl) != 0xFFFFu || s.ibit + 8u * ln > s.end_bit || s.opos + ln > s.out_len)
				return kStatusRefused;
			const GConstBytes from = (GConstBytes) s.in32 + (s.ibit >> 3);
			for (uint32_t k = lane; k < ln; k += 64u)
				s.out[s.opos + k] = from[k];
			s.opos += ln;
			s.ibit += 8u * ln;
		} else if (type == 1u || type == 2u) {
			int n_lit, n_dist;
			if (type == 1u) {
				n_lit = 288;
				n_dist = 32; // (30 and 31 never occur in valid data: refused when met)
				for (uint32_t i = lane; i < 288u + 32u; i += 64u)
					t.lens[i] = (uint8_t) (i < 144u ? 8 : i < 256u ? 9 : i < 280u ? 7 : i < 288u ? 8 : 5);
				wave_sync();
			} else {
				if (s.ibit + 14u > s.end_bit)
					return kStatusRefused;
				const uint32_t h = win.peek(s.ibit, 14);
				s.ibit += 14u;
				n_lit = (int) (h & 31u) + 257;
				n_dist = (int) ((h >> 5) & 31u) + 1;
				const int n_pre = (int) (h >> 10) + 4;
				if (n_lit > 286 || n_dist > 30)
					return kStatusRefused;
				// the code-length code: 3 bits each, in the order of RFC 1951 3.2.7
				const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
				if (lane < 19u)
					t.lens[lane] = 0;
				wave_sync();
				for (int i = 0; i < n_pre; i++) {
					const uint32_t x = win.peek(s.ibit, 3);
					s.ibit += 3u;
					if (lane == 0u)
						t.lens[kOrder[i]] = (uint8_t) x;
				}
				wave_sync();
				if (s.ibit > s.end_bit)
					return kStatusRefused;
				// its table: 7-bit root, no second level (no code is longer than 7 bits); byte entries
				{
					const uint32_t len = lane < 19u ? t.lens[lane] : 0u;
					uint32_t cnt[8], fst[8];
					int left = 1;
					uint32_t code = 0;
					cnt[0] = 0;
					bool any = false;
#pragma unroll
					for (int L = 1; L < 8; L++) {
						cnt[L] = (uint32_t) __popcll(__ballot(len == (uint32_t) L));
						left = (left << 1) - (int) cnt[L];
						code = (code + cnt[L - 1]) << 1;
						fst[L] = code;
						any = any || cnt[L];
					}
					if (left != 0 || !any) // the code-length code has to be complete (zlib's rule)
						return kStatusRefused;
					for (uint32_t j = lane; j < (1u << kPreRoot); j += 64u)
						t.pre[j] = 0;
					wave_sync();
					uint32_t rank = 0, f = 0;
#pragma unroll
					for (int L = 1; L < 8; L++) {
						const unsigned long long m = __ballot(len == (uint32_t) L);
						if (len == (uint32_t) L) {
							rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
							f = fst[L];
						}
					}
					if (len) {
						const uint32_t rev = __brev(f + rank) >> (32u - len);
						for (uint32_t j = rev; j < (1u << kPreRoot); j += 1u << len)
							t.pre[j] = (uint8_t) (len | (lane << 3));
					}
					wave_sync();
				}
				// the code lengths of both alphabets, run-length coded with that code (uniform, serial: a few hundred symbols)
				int k = 0;
				const int n_all = n_lit + n_dist;
				uint32_t prev = 0;
				while (k < n_all) {
					if (s.ibit > s.end_bit)
						return kStatusRefused;
					const uint32_t bits14 = win.peek(s.ibit, 14);
					const uint32_t pe = uni(t.pre[bits14 & ((1u << kPreRoot) - 1u)]);
					const uint32_t pl = pe & 7u, sym = pe >> 3;
					if (pl == 0u)
						return kStatusRefused;
					s.ibit += pl;
					if (sym < 16u) {
						if (lane == 0u)
							t.lens[k] = (uint8_t) sym;
						prev = sym;
						k++;
						continue;
					}
					const uint32_t x = bits14 >> pl;
					uint32_t rep, val = 0;
					if (sym == 16u) {
						if (k == 0)
							return kStatusRefused;
						val = prev;
						rep = 3u + (x & 3u);
						s.ibit += 2u;
					} else if (sym == 17u) {
						rep = 3u + (x & 7u);
						s.ibit += 3u;
					} else {
						rep = 11u + (x & 127u);
						s.ibit += 7u;
					}
					if (k + (int) rep > n_all)
						return kStatusRefused;
					for (uint32_t j = lane; j < rep; j += 64u)
						t.lens[k + (int) j] = (uint8_t) val;
					prev = val;
					k += (int) rep;
				}
				wave_sync();
				if (s.ibit > s.end_bit || t.lens[256] == 0) // (no end-of-block code)
					return kStatusRefused;
			}
			IW_T0();
			if (!build_table<kLitRoot, LitFormat>(t.lens, n_lit, t.lit, kLitCap, t.sorted, true))
				return kStatusRefused;
			if (!build_table<kDistRoot, DistFormat>(t.lens + n_lit, n_dist, t.dist, kDistCap, t.sorted, true))
				return kStatusRefused;
			IW_LAP(P_TABLES);
#ifdef IW_PROF
			if (run_symbols(s, t, luts, prof) != 0)
				return kStatusRefused;
#else
			{
				const SymbolsOut o = BATCHED ? run_symbols_batched_call(s.in32, s.out, s.ibit, s.end_bit, s.opos, s.out_len, s.safe_pos,
						(uint32_t) (uintptr_t) (LdsTables) &t, (uint32_t) (uintptr_t) (LdsLuts) &luts)
						: run_symbols_call(s.in32, s.out, s.ibit, s.end_bit, s.opos, s.out_len, s.safe_pos,
						(uint32_t) (uintptr_t) (LdsTables) &t, (uint32_t) (uintptr_t) (LdsLuts) &luts);
				s.ibit = o.ibit;
				s.opos = o.opos;
				s.safe_pos = o.safe_pos;
				if (o.rc != 0)
					return kStatusRefused;
			}
#endif
			win.d0 = 0xFFFFFF00u; // (the position moved on behind the window's back: force a reload)
		} else
			return kStatusRefused;
		if (final_block)
			break;
	}
	return (s.ibit <= s.end_bit && s.opos == s.out_len) ? kStatusOk : kStatusRefused;
}

constexpr int kWavesPerGroup = 4;

// status[b]: kStatusOk / kStatusRefused / kStatusCrc.  `ticket` (a zeroed counter, or NULL): waves take the next block off it -- a
// wave that is through early takes more, and the launch ends a block's time after its mean instead of with the wave that drew the
// ten dearest of 79 086 (a 1x genome is 9.65 rounds of the machine's 8 192 waves: dealt round robin, the last round is two thirds
// full); NULL: blocks round robin over the waves (round 2-3's order).
template <bool BATCHED> __global__ __launch_bounds__(64 * kWavesPerGroup, 8) void bgzf_inflate_wave_kernel(uint32_t n_blocks, const uint8_t *__restrict__ bytes,
		const conga_bgzf_block *__restrict__ blocks, const uint64_t *__restrict__ out_off, uint8_t *out,
		const uint32_t *__restrict__ crc_table, const uint32_t *__restrict__ x2n, uint8_t *__restrict__ status, uint32_t *ticket = nullptr)
{
	__shared__ WaveLds lds[kWavesPerGroup];
	__shared__ uint32_t s_crc[256];
	__shared__ uint32_t s_x2n[32];
	__shared__ Luts s_luts;
	if (threadIdx.x < 32)
		fill_luts(s_luts, threadIdx.x);
	for (int i = threadIdx.x; i < 256; i += blockDim.x)
		s_crc[i] = crc_table[i];
	if (threadIdx.x < 32)
		s_x2n[threadIdx.x] = x2n[threadIdx.x];
	__syncthreads();
	const uint32_t wave = uni(threadIdx.x >> 6); // (uniform by construction; said so, everything per block stays in scalar registers)
	const uint32_t n_waves = gridDim.x * kWavesPerGroup;
	auto next_block = [&](uint32_t b) -> uint32_t { // (every wave reaches b >= n_blocks: the counter only grows)
		if (!ticket)
			return b + n_waves;
		uint32_t t = 0;
		if (lane_id() == 0u)
			t = atomicAdd(ticket, 1u);
		return uni(t);
	};
	for (uint32_t b = ticket ? next_block(0) : blockIdx.x * kWavesPerGroup + wave; b < n_blocks; b = next_block(b)) {
		const conga_bgzf_block bl = blocks[b];
		uint8_t *dst = out + out_off[b];
#ifdef IW_PROF
		unsigned long long prof[P_N] = {};
		int st = inflate_block<false>(lds[wave], s_luts, bytes + bl.data_off, bl.data_len, dst, bl.inflated_len, prof);
		IW_T0();
#else
		int st = inflate_block<BATCHED>(lds[wave], s_luts, bytes + bl.data_off, bl.data_len, dst, bl.inflated_len);
#endif
		if (st == kStatusOk) {
			asm volatile("" ::: "memory");
			__builtin_amdgcn_s_waitcnt(0x0F70); // every store of the block has reached L2
			asm volatile("" ::: "memory");
			if (wave_crc32(dst, bl.inflated_len, s_crc, s_x2n) != bl.crc32)
				st = kStatusCrc;
		}
		IW_LAP(P_CRC);
#ifdef IW_PROF
		if (lane_id() == 0u)
			for (int k = 0; k < P_N; k++)
				atomicAdd(&g_prof[k], prof[k]);
#endif
		if (lane_id() == 0u)
			status[b] = (uint8_t) st;
		wave_sync();
	}
}

} // namespace iw
} // namespace conga
