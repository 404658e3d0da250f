// split_map.hip.h -- the split-read evidence path (--rp with --dups) for gfx950.
//
//   reference (file:line)                                   here
//   readReferenceSeq            common.c:423-463            ref_pack_kernel: the chromosome as 4-bit codes, upper-cased on the way
//   build_hash_table / init_hash_table  split_read.c:357-460   kmer_key_kernel -> a stable sort by 10-mer -> kmer_bounds_kernel
//   count_reads_bam's gate      bam_data.c:205-207          split_map_kernel, first lines
//   find_split_reads            split_read.c:206-354        split_map_kernel: one LANE per half-read element
//   almostPerfect_match_seq_ref split_read.c:75-204         ... a binary search in the seed's bucket, Hamming distance on packed codes
//   read_SplitReads / determine_SvType  bam_data.c:29-154   ... the lane's own mappings
//   count_ReadPairs             likelihood.c:41-94          ... a row is counted against the chromosome's SVs by the whole wave
//
// Round 2's kernel gave a read to a WAVE (64 lanes scan a bucket of ~60-240 positions of which ~1 lies within SR_LOOKAHEAD of
// the read, compare 50 bases, hash forty letters with ballots): 775 vector + 650 scalar instructions and five dependent trips
// to memory per read, 34 spilled registers.  Nearly all of that was the price of finding ONE position in an unordered bucket.
// build_hash_table fills a bucket in increasing position (split_read.c:394-440), so here the buckets ARE sorted, the positions
// within SR_LOOKAHEAD of a read are one binary search away, and what is left per half read is a few hundred scalar-looking
// operations -- which a lane does by itself: thirty-two reads per wave instead of one.
//
// Everything is compared in the 4-bit codes BAM itself uses for the letters find_split_reads can produce (A 1, C 2, G 4, T 8,
// N 15; split_read.c:262-273): the read's sequence is used as it lies in the record, the reference is packed once per layout.
// Two letters are equal exactly when their codes are; a code the reference leaves unset (anything else than 1 2 4 8 15) is N
// (documented deviation); a letter of the reference outside ACGTN gets code 0, which equals no code of a read.
// Complementing a letter reverses the four bits of its code (A 0001 <-> T 1000, C 0010 <-> G 0100, N 1111), so the reverse
// complement of a packed string is its bits in reverse order: one v_bfrev_b32 per eight bases.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"
#include "split_geom.h"

namespace conga {

constexpr int kKmerLen = 10;              // HASHKMERLEN (split_read.h:17)
constexpr int kKmerBuckets = 1 << 20;     // 4^10
constexpr uint32_t kKmerInvalid = 1u << 20; // sort key of a position whose 10-mer holds a letter outside ACGT: behind every bucket
constexpr int kMaxSrHit = 50000;          // MAX_SR_HIT (split_read.h:12)
constexpr int kMaxMapping = 100;          // MAX_MAPPING (split_read.h:13)
constexpr int kSrLookahead = 100000;      // SR_LOOKAHEAD (split_read.c:6)
constexpr int kSoftclipWindow = 50;       // SOFTCLIP_WRONGMAP_WINDOW (bam_data.h:14)
constexpr int kWrongmapWindow = 100;      // WRONGMAP_WINDOW (likelihood.h:17)
constexpr int kWrongmapWindowDel = 5000;  // WRONGMAP_WINDOW_DEL (likelihood.h:18)
constexpr int kSrMaxHalf = 512;           // char str[512] in find_split_reads (split_read.c:211)
constexpr int kSplitUnitReads = 128;      // reads per work unit: one 256-thread workgroup, a lane per (read, element)
constexpr int64_t kRefPadBases = 2048;    // code-0 bases behind every chromosome: a comparison may run past its end (a mismatch each)

// One chromosome's share of the split-read launch.
struct SplitSlot {
	int64_t sr_off, n_sr; // its records: staged form, in pos / mapq / flag / l_qseq / data_off; in place, in rec_off
	int64_t refn_off;     // its packed reference, in dwords
	int64_t L;
	int64_t kpos_off;     // its 10-mer index, in positions
	int32_t kidx;         // which offset table (4^10 + 2 entries each)
	int32_t sat_off, n_sat;
	int32_t iv0, n_dels, n_dups;
	int32_t slot;         // chromosome of the batch (its Small block takes the counters)
	uint32_t unit0;       // first work unit
	int32_t inplace;      // records lie in the inflated BAM stream (conga_reads_bgzf), found through rec_off
	int32_t pad;
	int64_t pres_off;     // its {solo, echo} bits, in uint2 (32 positions each)
};

// ---- the packed reference ------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t ref_code(uint8_t c)
{
	c = (c >= 'a' && c <= 'z') ? (uint8_t) (c - 32) : c; // readReferenceSeq upper-cases every base (common.c:449)
	return c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 4u : c == 'T' ? 8u : c == 'N' ? 15u : 0u;
}

// text[0, L) -> dwords of eight codes, base 8j in the top four bits of dword j; code 0 from L on (n_words covers the pad)
__global__ __launch_bounds__(256) void ref_pack_kernel(const uint8_t *__restrict__ text, int64_t L, uint32_t *__restrict__ out, int64_t n_words)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	for (int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; j < n_words; j += stride) {
		uint32_t w = 0;
		const int64_t i0 = j * 8;
		if (i0 + 8 <= L) {
			uint2 v;
			__builtin_memcpy(&v, text + i0, 8);
#pragma unroll
			for (int b = 0; b < 4; b++)
				w = (w << 4) | ref_code((uint8_t) (v.x >> (8 * b)));
#pragma unroll
			for (int b = 0; b < 4; b++)
				w = (w << 4) | ref_code((uint8_t) (v.y >> (8 * b)));
		} else {
			for (int b = 0; b < 8; b++)
				w = (w << 4) | (i0 + b < L ? ref_code(text[i0 + b]) : 0u);
		}
		out[j] = w;
	}
}

// bases [x, x + 8) of a packed text, first base in the top four bits
__device__ __forceinline__ uint32_t funnel_codes(uint32_t a, uint32_t b, uint32_t shift_bits)
{
	return shift_bits ? (a << shift_bits) | (b >> (32u - shift_bits)) : a;
}

// ---- codes of a read ---------------------------------------------------------------------------------------------------

// a code with exactly one bit set is a letter; every other code is N (split_read.c:262-273 leaves the char unset)
__device__ __forceinline__ uint32_t bits_per_code(uint32_t w)
{
	const uint32_t m = 0x11111111u;
	return (w & m) + ((w >> 1) & m) + ((w >> 2) & m) + ((w >> 3) & m);
}

__device__ __forceinline__ uint32_t norm_codes(uint32_t w)
{
	uint32_t t = bits_per_code(w) ^ 0x11111111u; // a zero nibble where exactly one bit is set
	t |= t >> 1;
	t |= t >> 2;
	t &= 0x11111111u;
	return w | (t * 15u);
}

// bases [s, s + 8) of the read whose packed sequence starts at `sq` (bam_get_seq: base 2i in the high half of byte i), s >= 0;
// reads up to nine bytes from sq + s / 2 on (behind a record's sequence lie its qualities: the callers keep 16 bytes of slack
// behind the last record)
__device__ __forceinline__ uint32_t read_codes(const uint8_t *sq, int s)
{
	uint2 v;
	__builtin_memcpy(&v, sq + (s >> 1), 8);
	const uint32_t u = __builtin_bswap32(v.x), w = __builtin_bswap32(v.y);
	return norm_codes((s & 1) ? (u << 4) | (w >> 28) : u);
}

// bases [k, k + 8) of the reverse complement of the read's bases [from, from + n): the bits of bases
// [from + n - 8 - k, from + n - k) in reverse order.  What lies in front of the half (or of the read) lands behind base n.
__device__ __forceinline__ uint32_t revcomp_codes(const uint8_t *sq, int from, int n, int k)
{
	const int s = from + n - 8 - k;
	const uint32_t w = s >= 0 ? read_codes(sq, s) : read_codes(sq, 0) >> (uint32_t) (-4 * s);
	return __brev(w);
}

// hash_function_ref (split_read.c:37-49) of the ten bases at the top of w0:w1: two bits per base, (letter & 6) >> 1
// (A 0, C 1, T 2, G 3), first base in the highest bits; -1 unless all ten are ACGT (is_kmer_valid, split_read.c:60-73)
__device__ __forceinline__ uint32_t squeeze_codes(uint32_t w)
{
	const uint32_t m = 0x11111111u;
	// code 1 -> 0, 2 -> 1, 8 -> 2, 4 -> 3: low bit = bit 1 | bit 2, high bit = bit 2 | bit 3
	uint32_t c = (((w >> 1) | (w >> 2)) & m) | ((((w >> 2) | (w >> 3)) & m) << 1);
	c = (c | (c >> 2)) & 0x0F0F0F0Fu;
	c = (c | (c >> 4)) & 0x00FF00FFu;
	c = (c | (c >> 8)) & 0x0000FFFFu;
	return c;
}

__device__ __forceinline__ int seed_hash(uint32_t w0, uint32_t w1)
{
	if (bits_per_code(w0) != 0x11111111u || (bits_per_code(w1) >> 24) != 0x11u)
		return -1;
	return (int) ((squeeze_codes(w0) << 4) | (squeeze_codes(w1) >> 12));
}

// how many of the first `valid` (1..8) codes of two words differ
__device__ __forceinline__ int codes_differ(uint32_t x, int valid)
{
	x |= x >> 1;
	x |= x >> 2;
	x &= 0x11111111u;
	if (valid < 8)
		x &= ~(0xFFFFFFFFu >> (4 * valid));
	return __popc(x);
}

// ---- the 10-mer index ------------------------------------------------------------------------------------------------
// build_hash_table (split_read.c:357-442) visits every position whose 10-mer is ACGT-only, in increasing order, and appends
// it to the 10-mer's bucket.  Here: a sort key per position (its hash; kKmerInvalid for the others), a STABLE sort of the
// positions by that key (kmer_sort.hip), and the table of where each key's run starts.  init_hash_table's rule -- a bucket
// with MAX_SR_HIT positions or more is dropped (split_read.c:450-457) -- is applied where a bucket is looked up.

// one thread per dword of the packed reference: the keys of its eight positions
__global__ __launch_bounds__(256) void kmer_key_kernel(const uint32_t *__restrict__ refn, int64_t L, uint32_t *__restrict__ keys)
{
	const int64_t n_words = (L + 7) >> 3;
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	for (int64_t j = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; j < n_words; j += stride) {
		const uint32_t a = refn[j], b = refn[j + 1], c = refn[j + 2]; // (the pad behind the chromosome is there to be read)
#pragma unroll
		for (int t = 0; t < 8; t++) {
			const int64_t i = j * 8 + t;
			if (i >= L)
				break;
			const int h = seed_hash(funnel_codes(a, b, 4u * t), funnel_codes(b, c, 4u * t)); // (code 0 behind L: never valid)
			keys[i] = h < 0 ? kKmerInvalid : (uint32_t) h;
		}
	}
}

// offset[k] = first index of sorted[] whose key is >= k, for k = 0 .. kKmerBuckets + 1 (offset[kKmerBuckets + 1] = n)
__global__ __launch_bounds__(256) void kmer_bounds_kernel(const uint32_t *__restrict__ sorted, int64_t n, uint32_t *__restrict__ offset)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
		const uint32_t lo = i == 0 ? 0u : sorted[i - 1] + 1u;                      // keys in front of this index are all < lo
		const uint32_t hi = i == n ? (uint32_t) kKmerBuckets + 1u : sorted[i];   // the key that starts here (or: the end)
		for (uint32_t k = lo; k <= hi; k++) // (empty unless a run starts at i; every key is written exactly once)
			offset[k] = (uint32_t) i;
	}
}

// ---- what a position knows about its own 10-mer -------------------------------------------------------------------
// A half read asks its seed's bucket for the positions within SR_LOOKAHEAD of its anchor (split_read.c:118,166).  The forward
// seed finds the read's own place there and, five times in six, nothing else; the reverse seed finds nothing five times in six.
// Each of those answers used to cost the bucket's bounds (a line of a 4 MB table) and a line of the bucket: two fetches from
// beyond L2 that say "nothing here" -- 4.4 of the kernel's 7 line fetches per element (profiles/split_map_traffic.json).
// But a half read that matches the reference where it lies asks about 10-mers OF the reference, and what the bucket would
// answer is a property of that place, known when the index is built.  The half lies at `own` (the read's second half at
// pos + l / 2, its first half at pos) and is looked for around the OTHER half's place, the anchor: |own - anchor| = l / 2.
// Two bits per position:
//   solo[p]  the 10-mer at p is in a bucket that init_hash_table keeps (fewer than MAX_SR_HIT positions, split_read.c:450-457)
//            and has no namesake q != p with |q - p| < SR_LOOKAHEAD + 1024: a forward seed that IS the reference's ten bases
//            at own and finds solo[own] set has own for its only candidate (|c - anchor| < SR_LOOKAHEAD implies
//            |c - own| < SR_LOOKAHEAD + l / 2) -- no bucket is looked at;
//   echo[q]  the reverse complement of the 10-mer at q occurs, in a kept bucket, at some p with |p - q| < SR_LOOKAHEAD + 1024:
//            a reverse seed that is the reverse complement of the reference's ten bases at q = own + n - 10 (the half's last
//            ten) and finds echo[q] clear has no candidate (|c - anchor| < SR_LOOKAHEAD implies |c - q| < SR_LOOKAHEAD + l / 2
//            + n, and a read is at most 1 022 bases).
// Both are read next to the read's own place -- the same lines for neighbouring reads, L2 hits -- and both are exact in the
// direction they are used: whatever they do not settle (a seed with a mismatch, a 10-mer with a namesake nearby: one element in
// five) goes to the bucket as before.  One uint2 {solo, echo} per 32 positions: a quarter of a byte per base.
constexpr int kOwnSlack = 1024; // (2 * kSrMaxHalf: a read's length bounds how far own and its last ten bases lie from the anchor)

// the reference's ten bases at c are the ten at the top of w0 : w1 (codes of an ACGT-only seed: seed_hash(w0, w1) >= 0)
__device__ __forceinline__ bool ref_ten_equal(const uint32_t *refn, int64_t c, uint32_t w0, uint32_t w1)
{
	const uint32_t *rw = refn + (c >> 3);
	const uint32_t sh = ((uint32_t) c & 7u) * 4u;
	const uint32_t a = rw[0], b = rw[1], d = rw[2];
	return funnel_codes(a, b, sh) == w0 && ((funnel_codes(b, d, sh) ^ w1) >> 24) == 0u;
}

// ---- mapping, pairing, counting ------------------------------------------------------------------------------------

struct SplitMapArgs {
	// staged records (conga_split_reads_commit): the fields of bam1_t the path touches
	const int32_t *pos;
	const uint8_t *mapq;
	const uint16_t *flag;
	const int32_t *l_qseq;
	const uint64_t *data_off; // per read: packed 4-bit sequence ((l + 1) / 2 bytes) followed by l quality bytes
	const uint8_t *data;
	// records in place (conga_reads_bgzf): where each kept record starts in the inflated BAM stream
	const uint64_t *rec_off;
	const uint8_t *stream;
	// per layout
	const uint32_t *refn;
	const int32_t *sat_start; // sorted, disjoint
	const int32_t *sat_end;
	const uint32_t *offset;
	const int32_t *positions;
	const uint2 *pres;
	uint32_t flags; // measurement switches (engine_knobs.h): 1 = do not ask the solo / echo bits, 2 = units in plain grid-stride order
	const int32_t *iv_start;
	const int32_t *iv_end;
	int32_t *support; // [n_iv] rp (dups) / border_rp (dels)
	const SplitSlot *slots;
	int32_t n_slots;
	uint32_t n_units;
	Small *small;
	int32_t mq_threshold, min_read_length;
};

// sonic_is_satellite(chr, a, b): any satellite interval overlapping [a, b).  Sorted and disjoint (conga_satellites merges
// them): the first interval that ends behind a either starts in front of b or nothing does.
__device__ __forceinline__ bool is_satellite_lane(const int32_t *ss, const int32_t *se, int n, int64_t a, int64_t b)
{
	int lo = 0, hi = n;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if ((int64_t) se[mid] <= a)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo < n && (int64_t) ss[lo] < b;
}

// hammingDistance (common.c:278-287) between the reference at `c` and the half read [from, from + n), or its reverse
// complement; a base behind the chromosome's end is a mismatch (the reference reads past its buffer there)
__device__ __forceinline__ int half_distance(const uint32_t *refn, int c, const uint8_t *sq, int from, int n, bool rev)
{
	const uint32_t *rw = refn + (c >> 3);
	const uint32_t sh = ((uint32_t) c & 7u) * 4u;
	uint32_t a = rw[0];
	int d = 0;
	for (int k = 0; k < n; k += 8) {
		const uint32_t b = rw[(k >> 3) + 1];
		const uint32_t r = funnel_codes(a, b, sh);
		a = b;
		const uint32_t h = rev ? revcomp_codes(sq, from, n, k) : read_codes(sq, from + k);
		d += codes_differ(r ^ h, min(8, n - k));
	}
	return d;
}

// ---- the same comparison with WIDE loads.  A lane's loads are what this kernel is made of (the machine takes ~250 G divergent
// 4-byte probes a second from L2 and ~55 G from beyond it, tools/gathercal.hip; a half read compared dword by dword is fifteen of
// them per candidate): 56 bases of either side are 28 bytes -- two 16-byte loads give eight dwords, a funnel shift the seven
// words of eight bases each.
__device__ __forceinline__ void load_dwords8(const void *p, uint32_t r[8])
{
	uint4 a, b;
	__builtin_memcpy(&a, p, 16);
	__builtin_memcpy(&b, static_cast<const uint8_t *>(p) + 16, 16);
	r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
}

// bases [s, s + 56) of the read as seven words of codes (read_codes' rules; up to 32 bytes from sq + s / 2 on are read: the
// callers' buffers end with that much slack)
__device__ __forceinline__ void read_words7(const uint8_t *sq, int s, uint32_t h[7])
{
	uint32_t u[8];
	load_dwords8(sq + (s >> 1), u);
#pragma unroll
	for (int j = 0; j < 8; j++)
		u[j] = __builtin_bswap32(u[j]);
	const bool odd = (s & 1) != 0;
#pragma unroll
	for (int j = 0; j < 7; j++)
		h[j] = norm_codes(odd ? (u[j] << 4) | (u[j + 1] >> 28) : u[j]);
}

// Hamming distance of the half read [from, from + n) -- h0 = its first 56 bases, read_words7(sq, from) -- and the reference at c
__device__ __forceinline__ int half_distance_fwd(const uint32_t *refn, int c, const uint8_t *sq, int from, int n, const uint32_t h0[7])
{
	int d = 0;
	for (int k0 = 0; k0 < n; k0 += 56) {
		uint32_t r[8], hk[7];
		load_dwords8(refn + ((c + k0) >> 3), r);
		const uint32_t sh = ((uint32_t) (c + k0) & 7u) * 4u;
		if (k0)
			read_words7(sq, from + k0, hk);
		const int m = n - k0;
#pragma unroll
		for (int j = 0; j < 7; j++)
			if (8 * j < m)
				d += codes_differ(funnel_codes(r[j], r[j + 1], sh) ^ (k0 ? hk[j] : h0[j]), min(8, m - 8 * j));
	}
	return d;
}

// ... and of its reverse complement and the reference at c: base k of the reverse complement is the complement of the half's
// base n - 1 - k, so the half's base i meets the complement of reference base c + n - 1 - i -- the half's words as they are
// against the reference's words from the window's END backwards, bits reversed.  (The last step's load begins 56 bases in
// front of the end of the (n - 1) % 56 + 1 bases it looks at: a window whose c is not that far from the text's first base goes
// to the caller's dword-by-dword form.)
__device__ __forceinline__ int half_distance_rev(const uint32_t *refn, int c, const uint8_t *sq, int from, int n, const uint32_t h0[7])
{
	int d = 0;
	for (int k0 = 0; k0 < n; k0 += 56) {
		const int t = sr_rev_step_base(c, n, k0); // word i of the load below: the eight bases from t + 8 i on; the half's word j meets word 6 - j
		uint32_t r[8], hk[7];
		load_dwords8(refn + (t >> 3), r);
		const uint32_t sh = ((uint32_t) t & 7u) * 4u;
		if (k0)
			read_words7(sq, from + k0, hk);
		const int m = n - k0;
#pragma unroll
		for (int j = 0; j < 7; j++)
			if (8 * j < m)
				d += codes_differ(__brev(funnel_codes(r[6 - j], r[7 - j], sh)) ^ (k0 ? hk[j] : h0[j]), min(8, m - 8 * j));
	}
	return d;
}

// The seed's bucket, or nothing (a seed with a letter outside ACGT; a bucket that init_hash_table dropped), from its first
// position p > anchor - SR_LOOKAHEAD on: [k0, b1) of `positions`.  The caller walks it while p < anchor + SR_LOOKAHEAD
// (abs(p - anchor) < SR_LOOKAHEAD, split_read.c:118,166): a read's window holds a position or two.
// A bucket's positions are spread over the chromosome, so where the window begins in it can be guessed (inv_len = 1 / L):
// eight positions around the guess -- two 16-byte loads, one cache line as a rule -- usually hold the answer; when they do not,
// a binary search of the side they point to does.  The answer never depends on the guess.
__device__ __forceinline__ void bucket_from(const uint32_t *offset, const int32_t *positions, int hash, int64_t anchor, float inv_len, uint32_t &k0,
		uint32_t &b1)
{
	k0 = b1 = 0;
	if (hash < 0)
		return;
	uint2 b;
	__builtin_memcpy(&b, offset + hash, 8);
	const uint32_t size = b.y - b.x;
	if (size >= (uint32_t) kMaxSrHit)
		return;
	const int64_t lo_pos = anchor - (kSrLookahead - 1);
	uint32_t lo = b.x, hi = b.y;
	if (size > 0 && lo_pos > 0) { // (lo_pos <= 0: the bucket's first position is the answer)
		uint32_t w0 = b.x;
		if (size > 8u) {
			const uint32_t guess = (uint32_t) fminf((float) lo_pos * inv_len * (float) size, (float) (size - 1u));
			w0 = b.x + min(max(guess, 4u) - 4u, size - 8u);
		}
		uint32_t w[8]; // (behind a chromosome's last bucket: the next chromosome's positions, or the buffer's slack)
		load_dwords8(positions + w0, w);
		const uint32_t in_bucket = min(size, 8u);
		uint32_t below = 0;
#pragma unroll
		for (int i = 0; i < 8; i++)
			below += ((uint32_t) i < in_bucket && (int64_t) (int32_t) w[i] < lo_pos) ? 1u : 0u;
		if (below == 0u)
			hi = w0; // (w0 itself qualifies, or is the bucket's end)
		else if (below == in_bucket && w0 + in_bucket < b.y)
			lo = w0 + in_bucket;
		else
			lo = hi = w0 + below;
	} else
		hi = lo;
	while (lo < hi) { // first position >= lo_pos
		const uint32_t mid = (lo + hi) >> 1;
		if ((int64_t) positions[mid] < lo_pos)
			lo = mid + 1;
		else
			hi = mid;
	}
	k0 = lo;
	b1 = b.y;
}

// solo: a thread per bucket walks its positions (they are one behind the other, in increasing position); the bits can be made at
// any time from what stays resident (offset, positions) -- the engine makes them before the SECOND split-read launch on an index, so
// that a single sample never pays for them (engine_compute.hip.h)
__global__ __launch_bounds__(256) void kmer_solo_kernel(const int32_t *__restrict__ positions, const uint32_t *__restrict__ offset, uint2 *__restrict__ bits)
{
	const uint32_t stride = gridDim.x * blockDim.x;
	for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < (uint32_t) kKmerBuckets; h += stride) {
		const uint32_t b0 = offset[h], b1 = offset[h + 1];
		if (b1 == b0 || b1 - b0 >= (uint32_t) kMaxSrHit)
			continue;
		int32_t prev = positions[b0];
		bool prev_far = true;
		for (uint32_t k = b0; k < b1; k++) {
			const int32_t p = prev;
			const bool last = k + 1 == b1;
			const int32_t next = last ? 0 : positions[k + 1];
			const bool next_far = last || next - p >= kSrLookahead + kOwnSlack;
			if (prev_far && next_far)
				atomicOr(&bits[p >> 5].x, 1u << (p & 31));
			prev_far = next_far;
			prev = next;
		}
	}
}

// echo: over the positions of the reference.  Where the window begins in the bucket is guessed like a read's (bucket_from): a line of
// bounds, a line of the bucket
__global__ __launch_bounds__(256) void kmer_echo_kernel(const uint32_t *__restrict__ refn, int64_t L, const uint32_t *__restrict__ offset,
		const int32_t *__restrict__ positions, uint2 *__restrict__ bits)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	const float inv_len = 1.0f / (float) max(L, (int64_t) 1);
	for (int64_t q = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; q + kKmerLen <= L; q += stride) {
		const uint32_t *rw = refn + (q >> 3);
		const uint32_t sh = ((uint32_t) q & 7u) * 4u;
		const uint32_t a = rw[0], b = rw[1], c = rw[2];
		const uint32_t w0 = funnel_codes(a, b, sh), w1 = funnel_codes(b, c, sh) & 0xFF000000u; // bases q .. q + 7, q + 8 .. q + 9
		// the reverse complement of the ten: base j of it is the complement of base 9 - j -- the twelve nibbles' bits in reverse
		// order, the two of w1 first
		const uint32_t r0 = __brev(w0), r1 = __brev(w1);      // r1: bases 9, 8 complemented in its low byte; r0: bases 7 .. 0
		const uint32_t v0 = (r1 << 24) | (r0 >> 8), v1 = r0 << 24;
		const int h = seed_hash(v0, v1);
		uint32_t k0 = 0, b1 = 0;
		// (bucket_from's window begins at anchor - (SR_LOOKAHEAD - 1): an anchor kOwnSlack in front of q makes it q - (LA + slack - 1))
		bucket_from(offset, positions, h, q - kOwnSlack, inv_len, k0, b1);
		if (k0 < b1 && (int64_t) positions[k0] <= q + (kSrLookahead + kOwnSlack - 1))
			atomicOr(&bits[q >> 5].y, 1u << (q & 31));
	}
}

// determine_SvType's geometry for a forward mapping `pm` of the element anchored at `anchor` (bam_data.c:40-61): the two
// pieces in position order must not overlap.  -> left end / right start of the row before the +-50 of bam_data.c:112-127.
__device__ __forceinline__ bool pair_geometry(int64_t anchor, int64_t pm, int l, int64_t &pos1_2, int64_t &pos2_1)
{
	const int length_split = l / 2, length_read = l - length_split;
	if (anchor < pm) {
		pos1_2 = anchor + length_read;
		pos2_1 = pm;
	} else if (pm < anchor) {
		pos1_2 = pm + length_split;
		pos2_1 = anchor;
	} else
		return false; // (pos1_/pos2_ unset in the reference)
	return pos1_2 < pos2_1;
}

__global__ __launch_bounds__(256) void split_map_kernel(SplitMapArgs g)
{
	const int lane = threadIdx.x & (kWave - 1);
	const int e = threadIdx.x & 1; // element 1 (e == 0) maps bases [l/2, l) at anchor pos; element 2 maps [0, l/2) at pos + l/2
	uint32_t n_elem = 0, n_map = 0, n_del = 0, n_dup = 0;
	int cur = 0;
	auto flush = [&](int s) {
		const int we = wave_sum_i32((int) n_elem), wm = wave_sum_i32((int) n_map), wd = wave_sum_i32((int) n_del), wu = wave_sum_i32((int) n_dup);
		if (lane == 0) {
			unsigned long long *cnt = g.small[g.slots[s].slot].counters;
			if (we)
				atomicAdd(&cnt[CNT_SR_ELEMENTS], (unsigned long long) we);
			if (wm)
				atomicAdd(&cnt[CNT_SR_MAPPINGS], (unsigned long long) wm);
			if (wd)
				atomicAdd(&cnt[CNT_SR_DEL_ROWS], (unsigned long long) wd);
			if (wu)
				atomicAdd(&cnt[CNT_SR_DUP_ROWS], (unsigned long long) wu);
		}
		n_elem = n_map = n_del = n_dup = 0;
	};
	// Workgroup w runs on XCD w % 8 (the dispatcher deals them round robin), and every XCD has an L2 of its own: XCD x takes the
	// runs of `per` consecutive units numbered x, x + 8, x + 16 ..., its workgroups side by side inside a run -- neighbouring reads,
	// the same lines of the reference and of its solo / echo bits, in ONE L2 instead of a copy in each of eight.
	const uint32_t per = gridDim.x >> 3, xcd = blockIdx.x & 7u, seat = blockIdx.x >> 3;
	const bool plain_order = (g.flags & 2u) != 0, ask_presence = (g.flags & 1u) == 0;
	for (uint32_t round = 0; plain_order || seat < per; round++) {
		const uint64_t u64 = plain_order ? (uint64_t) round * gridDim.x + blockIdx.x : ((uint64_t) round * 8u + xcd) * per + seat;
		if (u64 >= g.n_units)
			break;
		const uint32_t u = (uint32_t) u64;
		if (cur + 1 < g.n_slots && u >= g.slots[cur + 1].unit0) { // (the same for the whole workgroup)
			flush(cur);
			do
				cur++;
			while (cur + 1 < g.n_slots && u >= g.slots[cur + 1].unit0);
		}
		const SplitSlot &sl = g.slots[cur];
		const int64_t r = (int64_t) (u - sl.unit0) * kSplitUnitReads + (threadIdx.x >> 1);
		const int64_t L = sl.L;
		const uint32_t *refn = g.refn + sl.refn_off;
		const int32_t *sat_s = g.sat_start + sl.sat_off, *sat_e = g.sat_end + sl.sat_off;
		const uint32_t *offset = g.offset + (int64_t) sl.kidx * (kKmerBuckets + 2);
		const int32_t *positions = g.positions + sl.kpos_off;
		const uint2 *pres = g.pres + sl.pres_off;
		const float inv_len = 1.0f / (float) max(L, (int64_t) 1);

		// ---- the record, and the gate of count_reads_bam (bam_data.c:205-207) and of find_split_reads (split_read.c:216)
		bool alive = r < sl.n_sr;
		int p = 0, l = 0;
		const uint8_t *sq = g.data;
		if (alive) {
			int q, fl;
			if (sl.inplace) {
				const uint8_t *rec = g.stream + g.rec_off[sl.sr_off + r];
				uint32_t block_size;
				uint4 hd; // pos | l_read_name, mapq, bin | n_cigar_op, flag | l_seq
				__builtin_memcpy(&block_size, rec, 4);
				__builtin_memcpy(&hd, rec + 8, 16);
				p = (int) hd.x;
				q = (int) ((hd.y >> 8) & 0xFFu);
				fl = (int) (hd.z >> 16);
				l = (int) hd.w;
				const uint32_t seq_at = 36u + (hd.y & 0xFFu) + 4u * (hd.z & 0xFFFFu);
				sq = rec + seq_at;
				// (a record whose fields do not fit its own size is no read to split: the walk has vouched for block_size only)
				if (l < 0 || l > 2 * kSrMaxHalf - 2 || (uint64_t) seq_at + (uint64_t) ((l + 1) / 2 + l) > (uint64_t) block_size + 4u)
					alive = false;
			} else {
				const int64_t i = sl.sr_off + r;
				p = g.pos[i];
				q = (int) g.mapq[i];
				fl = (int) g.flag[i];
				l = g.l_qseq[i];
				sq = g.data + g.data_off[i];
			}
			if (!(q > g.mq_threshold) || !(l > g.min_read_length) || (fl & (0x100 | 0x800 | 0x400 | 0x200)) != 0)
				alive = false;
			if (p <= 0 || l > 2 * kSrMaxHalf - 2 || l < 2) // (a negative position is no record of this chromosome)
				alive = false;
		}
		if (alive && is_satellite_lane(sat_s, sat_e, sl.n_sat, p, (int64_t) p + 20))
			alive = false;
		const int half = l / 2;
		const int from = e == 0 ? half : 0, n = e == 0 ? l - half : half;
		const int64_t anchor = e == 0 ? (int64_t) p : (int64_t) p + half;

		// ---- mean base quality of the mapped half against the threshold (split_read.c:238-251,307-320).  The accumulator is a
		// float that is NOT reset between the two elements: element 1's sum starts at 0 and stays an integer below 2^24 -- exact
		// in any order --, element 2's starts from element 1's mean and is added in the reference's order.  A mean of
		// qualities is never negative, so with a threshold of 0 or below (the default is -1) nothing is dropped here.
		if (g.mq_threshold > 0 && alive) {
			const uint8_t *qq = sq + (l + 1) / 2;
			int isum = 0;
			for (int i = half; i < l; i++)
				isum += (int) qq[i];
			float avg = (float) isum / (float) (l - half);
			bool ok = !((int) floorf(avg) < g.mq_threshold);
			if (ok && e == 1) { // element 2 exists only behind element 1
				for (int i = 0; i < half; i++)
					avg = avg + (float) qq[i];
				avg = avg / (float) half;
				ok = !((int) floorf(avg) < g.mq_threshold);
			}
			alive = ok;
		}
		if (alive)
			n_elem++;

		// ---- almostPerfect_match_seq_ref (split_read.c:75-204): forward seed = the half's first ten bases, reverse seed = the
		// first ten of its reverse complement; hits within SR_LOOKAHEAD of the anchor and dist_max of the reference
		uint32_t f0 = 0, f1 = 0; // the forward seed's bucket from the window's first position on
		const int64_t hi_pos = anchor + kSrLookahead;
		int size = 0, row_candidates = 0;
		const int dist_max = (int) (0.05 * (double) n);
		uint32_t h0[7] = {0, 0, 0, 0, 0, 0, 0}; // the half's first 56 bases
		if (alive && n >= kKmerLen) {
			read_words7(sq, from, h0);
			const int hf = seed_hash(h0[0], h0[1]);
			int cf = 0;
			const int64_t own = e == 0 ? (int64_t) p + half : (int64_t) p; // where this half lies if the read is the reference's
			if (hf >= 0 && ask_presence && own + kKmerLen <= L && (pres[own >> 5].x >> (own & 31) & 1u) != 0u
					&& ref_ten_equal(refn, own, h0[0], h0[1])) {
				// the half's own place is its bucket's only position in the window (and cannot make a row: the pieces abut, pair_geometry)
				if (half_distance_fwd(refn, (int) own, sq, from, n, h0) <= dist_max)
					cf = 1;
			} else
				bucket_from(offset, positions, hf, anchor, inv_len, f0, f1);
			for (uint32_t k = f0; k < f1 && cf < kMaxMapping; k++) { // (a hundred hits or more: the element is dropped whatever follows)
				const int c = positions[k];
				if ((int64_t) c >= hi_pos)
					break;
				if (half_distance_fwd(refn, c, sq, from, n, h0) <= dist_max) {
					cf++;
					int64_t a1, a2;
					if (c > 0 && pair_geometry(anchor, c, l, a1, a2))
						row_candidates++;
				}
			}
			size = cf;
			if (cf < kMaxMapping) {
				uint32_t r0 = 0, r1 = 0;
				const int hr = seed_hash(revcomp_codes(sq, from, n, 0), revcomp_codes(sq, from, n, 8));
				// the reverse seed is the reverse complement of the half's last ten bases: when those are the reference's at
				// q = own + n - 10 and echo[q] is clear, its bucket holds nothing within the window
				const int64_t q = own + n - kKmerLen;
				const bool nothing = hr >= 0 && ask_presence && q + kKmerLen <= L && (pres[q >> 5].y >> (q & 31) & 1u) == 0u
						&& ref_ten_equal(refn, q, read_codes(sq, from + n - kKmerLen), read_codes(sq, from + n - kKmerLen + 8));
				if (!nothing)
					bucket_from(offset, positions, hr, anchor, inv_len, r0, r1);
				for (uint32_t k = r0; k < r1 && size <= kMaxMapping; k++) {
					const int c = positions[k];
					if ((int64_t) c >= hi_pos)
						break;
					// (half_distance_rev's LAST step looks at the window's first (n - 1) % 56 + 1 bases and loads the 56 that end
					// with them: those begin in front of base 0 of the reference text unless c is that far in.  `c + n >= 64` was the
					// test until tests/soak.py --bam-rp (seed 82, case 11) met a half of more than 56 bases within 56 of chromosome
					// 1's first base -- two dwords in front of the buffer, on a page that was not there)
					// (split_geom.h holds the test and the load's arithmetic: tests/test_split_geom.py walks every (c, n) on the host)
					const int d = sr_rev_wide_ok(c, n) ? half_distance_rev(refn, c, sq, from, n, h0) : half_distance(refn, c, sq, from, n, true);
					if (d <= dist_max)
						size++;
				}
			}
		}
		const bool mapped = size > 0 && size < kMaxMapping;
		if (mapped)
			n_map += (uint32_t) size;

		// ---- read_SplitReads / determine_SvType (bam_data.c:29-154) for the forward mappings that can make a row -- nearly
		// every mapping is the read's own locus, which cannot (the pieces abut or overlap) --, then count_ReadPairs
		// (likelihood.c:41-94) by the whole wave for each row
		bool more = mapped && row_candidates > 0 && 60 / size > g.mq_threshold && anchor < L
				&& !is_satellite_lane(sat_s, sat_e, sl.n_sat, anchor, anchor + 1);
		uint32_t k = f0;
		while (__any(more)) {
			bool row = false;
			bool is_del = false;
			int64_t left_end = 0, right_start = 0;
			while (more && !row) {
				if (k >= f1) {
					more = false;
					break;
				}
				const int c = positions[k++];
				if ((int64_t) c >= hi_pos) {
					more = false;
					break;
				}
				int64_t a1, a2;
				if (c > 0 && pair_geometry(anchor, c, l, a1, a2) && half_distance_fwd(refn, c, sq, from, n, h0) <= dist_max
						&& !is_satellite_lane(sat_s, sat_e, sl.n_sat, c, (int64_t) c + 1)) {
					row = true;
					is_del = (anchor < c && e == 0) || (anchor > c && e == 1);
					left_end = a1 - kSoftclipWindow;
					right_start = a2 + kSoftclipWindow;
				}
			}
			if (row) {
				if (is_del)
					n_del++;
				else
					n_dup++;
			}
			unsigned long long rows = __ballot(row);
			while (rows) { // (the same in every lane)
				const int src = __builtin_ctzll(rows);
				rows &= rows - 1ull;
				const bool del = __shfl((int) is_del, src, kWave) != 0;
				const int64_t le = ((int64_t) __shfl((int) (left_end >> 32), src, kWave) << 32) | (uint32_t) __shfl((int) left_end, src, kWave);
				const int64_t rs = ((int64_t) __shfl((int) (right_start >> 32), src, kWave) << 32) | (uint32_t) __shfl((int) right_start, src, kWave);
				if (del) {
					for (int i = lane; i < sl.n_dels; i += kWave) {
						const int64_t s0 = g.iv_start[sl.iv0 + i], e0 = g.iv_end[sl.iv0 + i];
						if (le <= s0 + kWrongmapWindow && le >= s0 - kWrongmapWindowDel && rs >= e0 - kWrongmapWindow && rs <= e0 + kWrongmapWindowDel)
							atomicAdd(&g.support[sl.iv0 + i], 1);
					}
				} else {
					for (int i = lane; i < sl.n_dups; i += kWave) {
						const int iv = sl.iv0 + sl.n_dels + i;
						const int64_t lo = (int64_t) g.iv_start[iv] - kWrongmapWindowDel, hi = (int64_t) g.iv_end[iv] + kWrongmapWindowDel;
						if (le >= lo && le <= hi && rs <= hi && rs >= lo)
							atomicAdd(&g.support[iv], 1);
					}
				}
			}
		}
	}
	if (g.n_slots > 0)
		flush(cur);
}

} // namespace conga
