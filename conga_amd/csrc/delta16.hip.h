// delta16.hip.h -- read positions handed over as 4- to 16-bit differences (conga_sample_reads_packed / _d16).
//
// The step of a cohort is the copy of the sample's tuples over PCIe (bench.py: step_bound), and a position-sorted sample's
// positions (bam1_core_t.pos in the order sam_itr_next yields them, bam_data.c:201-213) are a 32-bit number each only because
// nobody subtracted: at 1x two neighbours are ~100 bases apart.  The producer sends pos[i] - pos[i - 1] in W bits; whatever
// does not fit -- the first read of a chromosome, a gap of 2^W - 1 bases or more, a position in front of its predecessor -- is
// sent as all ones plus an entry (index, position) of a short exception list.  Here the differences become positions again:
// a segmented inclusive scan (an exception restarts the sum) in four launches -- where every chunk's exceptions begin in the list,
// per-chunk aggregates, one workgroup's scan over them, the chunks' local scans with their carry -- writing the int32 array every kernel of the path reads.  25.6 M reads:
// 51 MB (16 bits) or 32 MB (10 bits) over the link instead of 102, and ~70 us of launches (6 + 21 + 15 + 27: the differences are read
// twice, the positions written once) that hide under the next sample's copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace conga {

constexpr int kDeltaChunk = 2048; // differences per workgroup: 256 threads x 8 (one 16-byte load each)
// The differences are W bits wide (W = 4 .. 16: the producer picks what its coverage needs -- at 1x two neighbours are ~100 bases
// apart: 10 bits hold all but one difference in ten thousand; 9 bits would be fewer bytes, but six exceptions in a thousand reads
// are a search of the list each, and the expansion then takes longer than the copy it hides under; at 30x six bits do), packed
// little-endian: difference i occupies bits
// [i * W, (i + 1) * W) of the byte stream.  Eight of them are W whole bytes: a thread's share starts on a byte.  All ones = exception.

struct SegVal { // a run's sum and whether an exception (an absolute position) lies inside it
	int32_t v;
	uint32_t f;
};

__device__ __forceinline__ SegVal seg_combine(SegVal a, SegVal b) // a in front of b
{
	SegVal r;
	r.v = b.f ? b.v : a.v + b.v;
	r.f = a.f | b.f;
	return r;
}

// position of exception `i`.  The list is sorted by index (the host checks that, and that it holds the first read of every chromosome);
// that EVERY all-ones value of the stream has its entry is the producer's contract (include/conga_hip.h; conga_pack_positions keeps
// it by construction) and is not checked: an all-ones value without an entry takes the next entry's position, or 0 behind the last
// one -- memory-safe, positions wrong (as a rule out of order, which the engine's order check then reports).  Looked for
// among the exceptions of the read's own chunk, [lo, hi) of the list (delta_esc_rank_kernel) -- a dozen entries where
// the whole list has hundreds of thousands, and a narrow width makes one read in a hundred an exception
__device__ __forceinline__ int32_t escape_value(const uint32_t *esc_index, const int32_t *esc_pos, uint32_t n_esc, uint32_t lo, uint32_t hi, uint32_t i)
{
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if (esc_index[mid] < i)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo < n_esc ? esc_pos[lo] : 0;
}

// the thread's eight elements as segmented values, and their inclusive scan in place; -> the thread's aggregate
// (the stream has 16 bytes of slack behind its last difference: one 16-byte load whatever W is)
template <int W> __device__ __forceinline__ SegVal delta_thread_scan(const uint8_t *delta, uint64_t n, uint64_t first, const uint32_t *esc_index,
		const int32_t *esc_pos, uint32_t n_esc, uint32_t esc_lo, uint32_t esc_hi, SegVal e[8])
{
	constexpr uint32_t kEscape = (1u << W) - 1u;
	uint64_t raw[2] = {0, 0};
	if (first < n)
		__builtin_memcpy(raw, delta + (first >> 3) * W, 16);
	SegVal run = {0, 0u};
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const int bit = k * W; // (known at compile time once the loop is unrolled)
		const uint64_t lo = raw[bit >> 6] >> (bit & 63);
		const uint64_t hi = ((bit & 63) + W > 64) ? raw[1] << (64 - (bit & 63)) : 0ull;
		const uint32_t d = (uint32_t) (lo | hi) & kEscape;
		SegVal x;
		x.f = (d == kEscape && first + k < n) ? 1u : 0u;
		x.v = x.f ? escape_value(esc_index, esc_pos, n_esc, esc_lo, esc_hi, (uint32_t) (first + k)) : (first + k < n ? (int32_t) d : 0);
		run = seg_combine(run, x);
		e[k] = run;
	}
	return run;
}

// segmented scan of one value per thread over a workgroup of W waves; -> this thread's EXCLUSIVE prefix, the total in *all
template <int W> __device__ __forceinline__ SegVal delta_block_exclusive(SegVal mine, SegVal *s_wave /* [W] */, SegVal *all)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	SegVal inc = mine;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		SegVal p;
		p.v = __shfl_up(inc.v, o, 64);
		p.f = (uint32_t) __shfl_up((int) inc.f, o, 64);
		if (lane >= o)
			inc = seg_combine(p, inc);
	}
	if (lane == 63)
		s_wave[wv] = inc;
	__syncthreads();
	SegVal before = {0, 0u}; // of the waves in front of this one
	for (int k = 0; k < wv; k++)
		before = seg_combine(before, s_wave[k]);
	if (all) {
		SegVal t = before;
		for (int k = wv; k < W; k++)
			t = seg_combine(t, s_wave[k]);
		*all = t;
	}
	SegVal ex;
	ex.v = __shfl_up(inc.v, 1, 64);
	ex.f = (uint32_t) __shfl_up((int) inc.f, 1, 64);
	if (lane == 0)
		ex = SegVal{0, 0u};
	return seg_combine(before, ex);
}

// launch 0: rank[c] = how many exceptions lie in front of chunk c (c = 0 .. n_chunks: the last entry is n_esc)
__global__ __launch_bounds__(256) void delta_esc_rank_kernel(const uint32_t *__restrict__ esc_index, uint32_t n_esc, uint32_t n_chunks, uint32_t *__restrict__ rank)
{
	const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c > n_chunks)
		return;
	const uint64_t first = (uint64_t) c * kDeltaChunk;
	uint32_t lo = 0, hi = n_esc;
	while (lo < hi) {
		const uint32_t mid = (lo + hi) >> 1;
		if ((uint64_t) esc_index[mid] < first)
			lo = mid + 1;
		else
			hi = mid;
	}
	rank[c] = lo;
}

// launch 1: every chunk's aggregate
template <int W> __global__ __launch_bounds__(256) void delta_aggregate_kernel(const uint8_t *__restrict__ delta, uint64_t n, const uint32_t *__restrict__ esc_index,
		const int32_t *__restrict__ esc_pos, uint32_t n_esc, const uint32_t *__restrict__ rank, int2 *__restrict__ agg)
{
	__shared__ SegVal s_wave[4];
	SegVal e[8], all;
	const uint64_t first = (uint64_t) blockIdx.x * kDeltaChunk + (uint64_t) threadIdx.x * 8;
	const SegVal mine = delta_thread_scan<W>(delta, n, first, esc_index, esc_pos, n_esc, rank[blockIdx.x], rank[blockIdx.x + 1], e);
	(void) delta_block_exclusive<4>(mine, s_wave, &all);
	if (threadIdx.x == 0)
		agg[blockIdx.x] = make_int2(all.v, (int) all.f);
}

// launch 2 (one workgroup): exclusive scan over the chunks' aggregates -> what each chunk starts from.  Every thread takes a run of
// consecutive chunks (13 of a 1x genome's 12 500), folds it, ONE scan over the 1 024 runs gives each its start, and the run is walked
// once more to write the carries: one barrier (a scan per 1 024 chunks, thirteen of them one after the other, was 27 us of the
// expansion's 82 -- as long as the launch that writes the positions, profiles/r04n_kernel_stats.csv; 70 us in all now).
__global__ __launch_bounds__(1024) void delta_carry_kernel(const int2 *__restrict__ agg, uint32_t n_chunks, int32_t *__restrict__ carry)
{
	__shared__ SegVal s_wave[16];
	const uint32_t per = (n_chunks + 1023u) / 1024u;
	const uint32_t c0 = min(threadIdx.x * per, n_chunks), c1 = min(c0 + per, n_chunks);
	SegVal mine = {0, 0u};
	// (eight aggregates -- one 64-byte line of the thread's own -- asked for before the first is used: a run is two trips to L2, not thirteen)
	for (uint32_t c = c0; c < c1; c += 8) {
		int2 a[8];
#pragma unroll
		for (int j = 0; j < 8; j++)
			a[j] = c + j < c1 ? agg[c + j] : make_int2(0, 0);
#pragma unroll
		for (int j = 0; j < 8; j++)
			mine = seg_combine(mine, SegVal{a[j].x, (uint32_t) a[j].y});
	}
	SegVal running = delta_block_exclusive<16>(mine, s_wave, nullptr); // of the runs in front of this one
	for (uint32_t c = c0; c < c1; c += 8) {
		int2 a[8];
#pragma unroll
		for (int j = 0; j < 8; j++)
			a[j] = c + j < c1 ? agg[c + j] : make_int2(0, 0);
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (c + j < c1)
				carry[c + j] = running.v;
			running = seg_combine(running, SegVal{a[j].x, (uint32_t) a[j].y});
		}
	}
}

// launch 3: the positions
template <int W> __global__ __launch_bounds__(256) void delta_expand_kernel(const uint8_t *__restrict__ delta, uint64_t n, const uint32_t *__restrict__ esc_index,
		const int32_t *__restrict__ esc_pos, uint32_t n_esc, const uint32_t *__restrict__ rank, const int32_t *__restrict__ carry, int32_t *__restrict__ pos)
{
	__shared__ SegVal s_wave[4];
	SegVal e[8];
	const uint64_t first = (uint64_t) blockIdx.x * kDeltaChunk + (uint64_t) threadIdx.x * 8;
	const SegVal mine = delta_thread_scan<W>(delta, n, first, esc_index, esc_pos, n_esc, rank[blockIdx.x], rank[blockIdx.x + 1], e);
	SegVal before = delta_block_exclusive<4>(mine, s_wave, nullptr);
	before = seg_combine(SegVal{carry[blockIdx.x], 0u}, before);
	int32_t out[8];
#pragma unroll
	for (int k = 0; k < 8; k++)
		out[k] = seg_combine(before, e[k]).v;
	if (first + 8 <= n) {
		*reinterpret_cast<int4 *>(pos + first) = make_int4(out[0], out[1], out[2], out[3]);
		*reinterpret_cast<int4 *>(pos + first + 4) = make_int4(out[4], out[5], out[6], out[7]);
	} else
		for (int k = 0; k < 8; k++)
			if (first + k < n)
				pos[first + k] = out[k];
}

} // namespace conga
