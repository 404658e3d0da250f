// kernels.hip.h -- gfx950 (CDNA4, wave64) kernels of the read-depth / likelihood path.
//
// All kernels are HBM-bound integer/byte work, searches or short scalar chains; none is a contraction, so no
// MFMA is used.  Every kernel works on a BATCH of chromosomes ("slots") in one launch: a whole
// sample is a handful of launches, not a handful per chromosome.
//
// Two formulations of the same arithmetic:
//   tuple / row space (default)  tuple_pass_kernel (ingest_tuples + interval_count + interval_map_rows) and
//                                interval_chain_kernel (+ scoring): read_depth[] and mappability[] are never built
//   dense (the reference's)      ingest_kernel, depth_tile_kernel, paint_*, interval_reduce_kernel, interval_chain_kernel,
//                                interval_score_kernel: CONGA_FLAG_MATERIALIZE_DEPTH, unsorted reads / rows, possible wraps
//
// Layout in HBM; every per-chromosome array is a region of one concatenated buffer,
// located through the Slot table:
//   pos   int32[N], mapq uint8[N]   read tuples in BAM order (sorted by pos inside a slot)
//   rd    int16[sum L]              bam_info.read_depth          (common.h:91)   dense formulation only
//   map   float[sum L]              bam_info.mappability         (common.h:92)   painted tracks only
//   gc_*  uint8[sum n_win]          rounded GC% per `step`-base window
//   small Small[n_slots]            status, counters, GC histogram, expected_read_depth[101]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/conga_hip.h"
#include "serial_f32.h"

namespace conga {

constexpr int kWave = 64;
constexpr int kGcBins = 101; // read_distribution.c:51-52

constexpr uint32_t kStatusUnsorted = 1u; // Small.status bits
constexpr uint32_t kStatusWrapRisk = 2u; // some base of this chromosome may hold more than 32767 read starts (tuple pass)
// The tuple pass looks for runs of equal positions the way the commit path does on the host (conga_api.hip:
// note_equal_runs): the thread that holds tuples 4..7 of a 1024-tuple chunk compares tuple 3 with the one kWrapProbe
// further on.  A run of 32768 equal positions [j, j + 32768) contains a whole chunk [i, i + 1024) with i <= j + 1023,
// and i + 3 + kWrapProbe <= j + 31746 lies inside the run too: the pair is equal and the chunk lies inside one
// chromosome (the plain path).  Conservative: runs from 30721 on may be flagged.
constexpr uint32_t kWrapProbe = 30720;
enum { CNT_COUNTED = 0, CNT_OUT_OF_RANGE, CNT_SR_ELEMENTS, CNT_SR_MAPPINGS, CNT_SR_DEL_ROWS, CNT_SR_DUP_ROWS, CNT_N };

// One chromosome of the batch.  Offsets are in elements of the respective concatenated buffer.
struct Slot {
	int64_t L;        // chromosome length (sonic->chromosome_lengths[chr_index])
	int64_t rd_off;   // first element in rd / map (multiple of 8)
	int64_t read_off; // first tuple in pos / mapq
	int64_t n_reads;
	int64_t gc_off;   // first byte in gc_hist / gc_like (multiple of 16)
	int64_t n_win;
	int64_t tile0;    // first global depth tile
	int64_t n_tiles;
	int64_t tidx_off; // first entry in tile_start (n_tiles + 1 entries per slot)
};

// Small per-chromosome block, read back after every compute.
struct Small {
	uint32_t status;
	uint32_t pad;
	unsigned long long counters[CNT_N];
	unsigned long long hist_sum[kGcBins];   // rd_per_gc_unfiltered (read_distribution.c:52)
	unsigned long long hist_bases[kGcBins]; // window_per_gc (read_distribution.c:51)
	float E[kGcBins];                       // expected_read_depth (common.h:94)
	float pad2;
};

// -------------------------------------------------------------------------------------------
// wave64 helpers (DPP-backed shuffles)
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
	for (int o = kWave / 2; o > 0; o >>= 1)
		v += __shfl_down(v, o, kWave);
	return v;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
	for (int o = kWave / 2; o > 0; o >>= 1)
		v += __shfl_down(v, o, kWave);
	return v;
}

// last slot whose first key is <= x; keys are non-decreasing
template <typename KeyOf> __device__ __forceinline__ int find_slot(int n_slots, int64_t x, KeyOf key)
{
	int lo = 0, hi = n_slots; // invariant: key(lo) <= x < key(hi)
	while (hi - lo > 1) {
		const int mid = (lo + hi) >> 1;
		if (key(mid) <= x)
			lo = mid;
		else
			hi = mid;
	}
	return lo;
}

// -------------------------------------------------------------------------------------------
// K0 ingest: one pass over every read tuple of the batch.
//   * flags tuples that break the position order inside a chromosome (the tile index below needs
//     sorted input, which is what sam_itr_next over an indexed BAM yields: bam_data.c:201,293);
//   * counts tuples outside [0, L) (the reference would write out of bounds: bam_data.c:213);
//   * builds tile_first[t] = index (into the concatenated tuple arrays) of the first tuple whose
//     position falls in GLOBAL depth tile t or later, so the depth kernel needs no search: the tuples
//     of tile t are [tile_first[t], tile_first[t + 1]).  Chromosomes are concatenated in begin order
//     and sorted inside, so the global tile number never decreases along the tuple array.  Entries
//     past the last tuple's tile keep the memset value 0xFFFFFFFF (= "n_total").  Out-of-range
//     tuples are filed under their chromosome's first / last tile, where the depth kernel's range
//     check drops them.
// -------------------------------------------------------------------------------------------
// p / tile_len in 32-bit arithmetic: a float estimate is at most 1 off (relative error 2^-23 on a quotient
// below 2^21), and one multiply-subtract settles it.
__device__ __forceinline__ uint32_t div_tile(uint32_t p, uint32_t tile_len, float inv_tile_len)
{
	uint32_t q = (uint32_t) ((float) p * inv_tile_len);
	const int32_t r = (int32_t) (p - q * tile_len);
	if (r < 0)
		q--;
	else if ((uint32_t) r >= tile_len)
		q++;
	return q;
}

struct IngestSlot { // the fields of a Slot this kernel needs, in 32 bits
	uint32_t r0, r1;   // tuple index range
	int32_t L;
	uint32_t tile0, last_tile;
};

__device__ __forceinline__ IngestSlot ingest_slot(const Slot &sl)
{
	IngestSlot o;
	o.r0 = (uint32_t) sl.read_off;
	o.r1 = (uint32_t) (sl.read_off + sl.n_reads);
	o.L = (int32_t) sl.L;
	o.tile0 = (uint32_t) sl.tile0;
	o.last_tile = (uint32_t) (sl.tile0 + sl.n_tiles - 1);
	return o;
}

__device__ __forceinline__ uint32_t global_tile_of(const IngestSlot &sl, int32_t p, uint32_t tile_len, float inv_tile_len)
{
	if (p < 0)
		return sl.tile0;
	if (p >= sl.L)
		return sl.last_tile;
	return sl.tile0 + div_tile((uint32_t) p, tile_len, inv_tile_len);
}

__global__ __launch_bounds__(256) void ingest_kernel(const int32_t *__restrict__ pos, int64_t n_total64,
		const Slot *__restrict__ slots, int n_slots, int32_t tile_len_, uint32_t *__restrict__ tile_first,
		Small *__restrict__ small)
{
	const uint32_t n_total = (uint32_t) n_total64; // < 2^32 - 16 (conga_reads_commit)
	const uint32_t tile_len = (uint32_t) tile_len_;
	const float inv_T = 1.0f / (float) tile_len;
	const uint64_t stride = (uint64_t) gridDim.x * blockDim.x * 4;
	int s = -1;
	IngestSlot sl = {1, 0, 0, 0, 0}; // empty range: the first tuple refreshes it
	// four tuples (one 16-byte load) per lane per step: the kernel is bound by bytes in flight
	for (uint64_t i0 = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n_total; i0 += stride) {
		int32_t p4[4];
		if (i0 + 4 <= n_total) {
			const int4 q = *reinterpret_cast<const int4 *>(pos + i0);
			p4[0] = q.x;
			p4[1] = q.y;
			p4[2] = q.z;
			p4[3] = q.w;
		} else {
			for (int e = 0; e < 4; e++)
				p4[e] = (i0 + e < n_total) ? pos[i0 + e] : 0;
		}
		int32_t prev = (i0 > 0) ? pos[i0 - 1] : 0;
		uint32_t t_prev = 0;
		bool t_prev_known = false; // t_prev is the global tile of `prev`, which lies in the same chromosome
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const uint32_t i = (uint32_t) i0 + e;
			if (i >= n_total)
				break;
			if (i < sl.r0 || i >= sl.r1) {
				s = find_slot(n_slots, (int64_t) i, [&](int k) { return slots[k].read_off; });
				sl = ingest_slot(slots[s]);
				t_prev_known = false;
			}
			const int32_t p = p4[e];
			if (p < 0 || p >= sl.L)
				atomicAdd(&small[s].counters[CNT_OUT_OF_RANGE], 1ull);
			const uint32_t t_cur = global_tile_of(sl, p, tile_len, inv_T);
			int64_t first_fill;
			if (i == 0)
				first_fill = 0;
			else {
				if (i > sl.r0) { // `prev` is in the same chromosome
					if (p < prev)
						atomicOr(&small[s].status, kStatusUnsorted);
					if (!t_prev_known)
						t_prev = global_tile_of(sl, prev, tile_len, inv_T);
				} else {
					// first tuple of this chromosome: the previous tuple belongs to an earlier one
					const int sp = find_slot(n_slots, (int64_t) i - 1, [&](int k) { return slots[k].read_off; });
					t_prev = global_tile_of(ingest_slot(slots[sp]), prev, tile_len, inv_T);
				}
				first_fill = (int64_t) t_prev + 1;
			}
			for (int64_t t = first_fill; t <= (int64_t) t_cur; t++)
				tile_first[t] = i;
			prev = p;
			t_prev = t_cur;
			t_prev_known = true;
		}
	}
}

// -------------------------------------------------------------------------------------------
// Tuple-space formulation (default whenever it is provably identical to the dense one).
//
// read_depth[i] is "how many kept reads start at i", so as long as no position collects more than 32767 reads
// (no `short` wrap -- conga_reads_commit checks that on the host and falls back to the dense kernels otherwise)
//   rd_per_gc[g]            = number of kept reads whose start lies in a window of GC bin g
//   sum(read_depth[a .. b)) = number of kept reads with a <= pos < b
// and neither needs read_depth[L] in HBM: the sample is 5 bytes per READ of traffic instead of 4+ bytes per BASE.
//
// K0' ingest_tuples: one pass over the tuples.  Each workgroup owns a contiguous run of 1024-tuple chunks and
// keeps the next chunk's loads in flight while it works on the current one.
//   * the checks of K0 (order inside a chromosome, range);
//   * kept = in range and mapq > threshold (bam_data.c:205): counted, and added to the GC bin of its window in a
//     workgroup-private LDS histogram (32 copies, copy c at odd stride 101 words, so one wave's atomics on one
//     bin land in 32 different banks); one global atomic per non-empty bin when the workgroup ends or moves on
//     to another chromosome;
// A chunk that straddles chromosomes (or is cut short by the end of the batch) is taken once per chromosome with
// the other tuples masked out.
// -------------------------------------------------------------------------------------------
constexpr int kTupleBlock = 256;
constexpr int kTupleChunk = kTupleBlock * 4; // tuples per workgroup step: one 16-byte position load per lane
constexpr int kHistCopies = 32;

struct TupleArgs {
	const int32_t *pos;
	const uint8_t *mapq;
	uint32_t n_total;
	const Slot *slots;
	int n_slots;
	const uint8_t *gc_hist;
	int32_t step;
	int32_t mq_threshold;
	Small *small;
	uint32_t n_chunks;          // ceil(n_total / kTupleChunk)
	uint32_t chunks_per_block;  // consecutive chunks per workgroup
	const struct TupleBlockHome *block_home; // per workgroup: the chromosome of its first chunk, looked up by the host
};

struct TupleSlot {
	uint32_t r0, r1; // tuple index range
	int32_t L;
	uint32_t gc_off;
};

struct TupleBlockHome { // 32 bytes: one load at workgroup start instead of a binary search over the Slot table
	TupleSlot sl;
	int32_t slot; // -1: the first chunk is not inside one chromosome
	int32_t pad[3];
};

__device__ __forceinline__ TupleSlot tuple_slot(const Slot &sl)
{
	TupleSlot o;
	o.r0 = (uint32_t) sl.read_off;
	o.r1 = (uint32_t) (sl.read_off + sl.n_reads);
	o.L = (int32_t) sl.L;
	o.gc_off = (uint32_t) sl.gc_off;
	return o;
}

struct TupleRegs { // one chunk's share of a lane: four tuples and, for lane 0 of a wave, the tuple in front of them
	int4 q;
	uint32_t mq;
	int32_t pv;
};

// ALL: the threshold is below 0, every read counts (cmdline.c:188-194): the MAPQ bytes are neither read nor tested.
template <bool ALL> __device__ __forceinline__ TupleRegs load_tuples(const TupleArgs &a, uint32_t chunk, uint32_t n_chunks)
{
	TupleRegs r;
	r.q = make_int4(0, 0, 0, 0);
	r.mq = 0;
	r.pv = 0;
	const uint32_t i0 = chunk * (uint32_t) kTupleChunk + threadIdx.x * 4;
	if (chunk >= n_chunks || i0 >= a.n_total)
		return r;
	// (with a threshold below 0 every MAPQ passes `qual > mq_threshold`, bam_data.c:205: the bytes are not read -- they may not
	// even have been sent, conga_sample_reads)
	constexpr bool want_mq = !ALL;
	if (i0 + 4 <= a.n_total) {
		r.q = *reinterpret_cast<const int4 *>(a.pos + i0);
		if (want_mq)
			r.mq = *reinterpret_cast<const uint32_t *>(a.mapq + i0);
	} else { // the lane that holds the ragged end of the batch: element by element, zeros behind the last tuple
		int32_t t[4] = {0, 0, 0, 0};
		for (uint32_t e = 0; i0 + e < a.n_total; e++) {
			t[e] = a.pos[i0 + e];
			if (want_mq)
				r.mq |= (uint32_t) a.mapq[i0 + e] << (8 * e);
		}
		r.q = make_int4(t[0], t[1], t[2], t[3]);
	}
	if ((threadIdx.x & (kWave - 1)) == 0 && i0 > 0)
		r.pv = a.pos[i0 - 1];
	else if (threadIdx.x == 1 && i0 - 1 + kWrapProbe < a.n_total)
		r.pv = a.pos[i0 - 1 + kWrapProbe]; // the wrap probe's far end (this lane's `pv` is otherwise unused)
	return r;
}

// A chunk that lies inside one chromosome (all but ~21 of 28,000): no per-tuple chromosome lookup, four tuples
// per lane handled without a branch except for the rare tile boundary / out-of-range tuple.
struct GcRegs { // stage 1 -> stage 2 of a chunk: the GC bins of a lane's four tuples (loads in flight) and which of them count
	int g[4];
	uint32_t kmask;
};

// Stage 1: checks, read filter, window index, and the four GC-byte loads -- issued, not waited for.
// MASKED: only the tuples with index in [lo, hi) belong to chromosome `home` (a chunk that straddles two).
template <bool MASKED, bool ALL> __device__ __forceinline__ GcRegs ingest_chunk_inside(const TupleArgs &a, const TupleSlot &sl, int home,
		uint32_t base, const TupleRegs &r, float inv_step, int &kept, uint32_t lo = 0, uint32_t hi = 0)
{
	const int lane = threadIdx.x & (kWave - 1);
	const uint32_t step = (uint32_t) a.step;
	const uint32_t i0 = base + threadIdx.x * 4;
	const int32_t p[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
	int32_t prev = __shfl_up(p[3], 1, kWave); // the neighbour lane holds the tuple in front of this lane's four
	if (lane == 0)
		prev = r.pv;
	bool mine[4] = {true, true, true, true};
	if (MASKED) {
#pragma unroll
		for (int e = 0; e < 4; e++)
			mine[e] = i0 + e >= lo && i0 + e < hi;
	}
	// a tuple is compared with its predecessor only when that one belongs to the same chromosome
	bool unsorted = mine[0] & (i0 != sl.r0) & (p[0] < prev);
#pragma unroll
	for (int e = 1; e < 4; e++)
		unsorted |= mine[e] & (!MASKED | (i0 + e != sl.r0)) & (p[e] < p[e - 1]);
	if (unsorted)
		atomicOr(&a.small[home].status, kStatusUnsorted);
	// wrap probe (kWrapProbe above): thread 1 holds tuple base + 3 in `prev` and tuple base + 3 + kWrapProbe in r.pv
	if (!MASKED && threadIdx.x == 1 && (uint64_t) base + 3u + kWrapProbe < (uint64_t) sl.r1 && prev == r.pv)
		atomicOr(&a.small[home].status, kStatusWrapRisk);
	bool in[4];
	int n_out = 0;
#pragma unroll
	for (int e = 0; e < 4; e++) {
		in[e] = mine[e] & ((uint32_t) p[e] < (uint32_t) sl.L);
		n_out += (mine[e] & !in[e]) ? 1 : 0;
	}
	if (n_out)
		atomicAdd(&a.small[home].counters[CNT_OUT_OF_RANGE], (unsigned long long) n_out);
	uint32_t w[4];
	GcRegs out;
	out.kmask = 0;
#pragma unroll
	for (int e = 0; e < 4; e++) {
		const bool k = ALL ? in[e] : in[e] && (int) ((r.mq >> (8 * e)) & 0xFFu) > a.mq_threshold;
		// the usual 100-base window: one multiply-high (p < 2^31: floor(p / 100) = (p * 0x51EB851F) >> 37)
		const uint32_t q = (step == 100u) ? (__umulhi((uint32_t) p[e], 0x51EB851Fu) >> 5)
				: (step == 1u) ? (uint32_t) p[e] : div_tile((uint32_t) p[e], step, inv_step);
		w[e] = k ? q : 0u;
		out.kmask |= (k ? 1u : 0u) << e;
	}
	kept += __popc(out.kmask);
	const uint8_t *gc = a.gc_hist + sl.gc_off; // uniform base, 32-bit lane offsets
#pragma unroll
	for (int e = 0; e < 4; e++)
		out.g[e] = gc[w[e]];
	return out;
}

// Stage 2, one workgroup step later: the histogram adds.
__device__ __forceinline__ void ingest_chunk_count(const GcRegs &r, uint32_t *my_hist)
{
#pragma unroll
	for (int e = 0; e < 4; e++)
		if ((r.kmask >> e) & 1u)
			atomicAdd(&my_hist[r.g[e]], 1u);
}

template <bool ALL> __device__ __forceinline__ void ingest_tuples_body(const TupleArgs &a, uint32_t block, uint32_t *hist, uint32_t &kept_block)
{
	const uint32_t c0 = block * a.chunks_per_block;
	const uint32_t c1 = min(c0 + a.chunks_per_block, a.n_chunks);
	const TupleBlockHome bh = a.block_home[block]; // in flight with the tuple loads below
	// software pipeline over the workgroup's chunks: tuples two chunks ahead, GC bytes one chunk ahead
	TupleRegs t1 = load_tuples<ALL>(a, c0, c1), t2 = load_tuples<ALL>(a, c0 + 1, c1); // in flight while the histogram is cleared
	for (int k = threadIdx.x; k < kHistCopies * kGcBins; k += kTupleBlock)
		hist[k] = 0;
	if (threadIdx.x == 0)
		kept_block = 0;
	__syncthreads();

	const int lane = threadIdx.x & (kWave - 1);
	const float inv_step = 1.0f / (float) a.step;
	uint32_t *const my_hist = hist + (threadIdx.x & (kHistCopies - 1)) * kGcBins;
	int home = bh.slot; // chromosome the LDS histogram belongs to
	TupleSlot hs = bh.sl; // (an empty range when slot is -1)
	int kept = 0;
	GcRegs pend = {{0, 0, 0, 0}, 0u}; // stage-1 result of the previous chunk (kmask 0: nothing pending)

	auto flush = [&]() { // workgroup-uniform
		kept = wave_sum_i32(kept);
		if (lane == 0 && kept)
			atomicAdd(&kept_block, (uint32_t) kept);
		kept = 0;
		__syncthreads();
		if (threadIdx.x < kGcBins) {
			uint32_t sum = 0;
#pragma unroll 8
			for (int c = 0; c < kHistCopies; c++) {
				sum += hist[c * kGcBins + threadIdx.x];
				hist[c * kGcBins + threadIdx.x] = 0;
			}
			if (sum)
				atomicAdd(&a.small[home].hist_sum[threadIdx.x], (unsigned long long) sum);
		}
		if (threadIdx.x == 128 && kept_block) {
			atomicAdd(&a.small[home].counters[CNT_COUNTED], (unsigned long long) kept_block);
			kept_block = 0;
		}
		__syncthreads();
	};

	for (uint32_t c = c0; c < c1; c++) {
		const TupleRegs cur = t1;
		t1 = t2;
		t2 = load_tuples<ALL>(a, c + 2, c1);
		const uint32_t base = c * (uint32_t) kTupleChunk;
		if (!(base >= hs.r0 && (uint64_t) base + kTupleChunk <= (uint64_t) hs.r1)) { // chunk not inside `home`
			ingest_chunk_count(pend, my_hist); // the previous chunk still belongs to the old chromosome
			pend.kmask = 0;
			if (home >= 0)
				flush(); // (keeps `home`: the walk below starts from it)
			// One masked pass per chromosome the chunk touches (usually one or two; the ragged last chunk of the batch is
			// simply cut at n_total); the last one stays `home`.  Chromosomes follow each other in the tuple array, so
			// the walk continues from the previous home instead of searching.
			const uint32_t chunk_end = min(base + (uint32_t) kTupleChunk, a.n_total);
			uint32_t from = base;
			if (home < 0)
				home = find_slot(a.n_slots, (int64_t) from, [&](int k) { return a.slots[k].read_off; });
			for (;;) {
				hs = tuple_slot(a.slots[home]);
				while (from >= hs.r1) // next chromosome that holds tuples (never runs off the table: from < n_total)
					hs = tuple_slot(a.slots[++home]);
				const uint32_t to = min(chunk_end, hs.r1);
				if (from == base && to == base + (uint32_t) kTupleChunk)
					break; // the whole chunk lies in this chromosome after all: the plain path below
				const GcRegs g = ingest_chunk_inside<true, ALL>(a, hs, home, base, cur, inv_step, kept, from, to);
				ingest_chunk_count(g, my_hist);
				from = to;
				if (from >= chunk_end)
					break;
				flush();
			}
			if (from != base)
				continue; // handled chromosome by chromosome
		}
		const GcRegs g = ingest_chunk_inside<false, ALL>(a, hs, home, base, cur, inv_step, kept); // GC loads of this chunk go out ...
		ingest_chunk_count(pend, my_hist);                                                  // ... before the previous chunk's are used
		pend = g;
	}
	ingest_chunk_count(pend, my_hist);
	if (home >= 0)
		flush();
}

template <bool ALL> __global__ __launch_bounds__(kTupleBlock, 8) void ingest_tuples_kernel(TupleArgs a)
{
	__shared__ uint32_t hist[kHistCopies * kGcBins];
	__shared__ uint32_t kept_block;
	ingest_tuples_body<ALL>(a, blockIdx.x, hist, kept_block);
}

// -------------------------------------------------------------------------------------------
// K4' interval_count: observed_rd_sv of the tuple-space formulation (likelihood.c:111-114 without read_depth).
// One lane per reduce item [lo, lo + len) (<= 16384 bases of one interval): two interleaved binary searches over
// the chromosome's sorted positions give the index range of the tuples that start inside it -- with the default
// threshold (-1: every read counts, cmdline.c:188-194) the difference IS the sum -- otherwise the wave walks the
// 64 ranges of its lanes one after the other and counts the MAPQ bytes above the threshold.  Reads only the
// tuples, so it runs beside ingest_tuples on the second stream.  Integer, order-free.
// -------------------------------------------------------------------------------------------
struct CountArgs {
	const int32_t *pos;
	const uint8_t *mapq;
	const int32_t *item_slot; // the item's chromosome: its tuple index range comes from the Slot table, which is
	const Slot *slots;        // the only thing that changes when another sample's reads are put behind the same layout
	const int32_t *item_lo;  // first base, chromosome coordinates
	const int32_t *item_len;
	const int32_t *item_iv;
	int64_t n_items;
	int32_t mq_threshold;
	int32_t *observed; // [n_iv], zeroed before launch
};

__device__ __forceinline__ void interval_count_body(const CountArgs &a, int64_t block)
{
	const int lane = threadIdx.x & (kWave - 1);
	const int64_t item = block * blockDim.x + threadIdx.x;
	const bool have = item < a.n_items;
	int32_t lo = 0, hi = 0;
	uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0; // first tuple with pos >= lo lies in [a0, a1], with pos >= hi in [b0, b1]
	if (have) {
		lo = a.item_lo[item];
		hi = lo + a.item_len[item];
		const Slot &sl = a.slots[a.item_slot[item]];
		a0 = b0 = (uint32_t) sl.read_off;
		a1 = b1 = (uint32_t) (sl.read_off + sl.n_reads);
	}
	while (__any(a0 < a1 || b0 < b1)) {
		const uint32_t ma = a0 + ((a1 - a0) >> 1), mb = b0 + ((b1 - b0) >> 1);
		const int32_t pa = (a0 < a1) ? a.pos[ma] : 0;
		const int32_t pb = (b0 < b1) ? a.pos[mb] : 0;
		if (a0 < a1) {
			if (pa < lo)
				a0 = ma + 1;
			else
				a1 = ma;
		}
		if (b0 < b1) {
			if (pb < hi)
				b0 = mb + 1;
			else
				b1 = mb;
		}
	}
	int cnt = (int) (b0 - a0); // tuples with lo <= pos < hi
	if (a.mq_threshold >= 0) { // some reads may be filtered out: look at the MAPQ bytes
		cnt = 0;
		for (int i = 0; i < kWave; i++) {
			const uint32_t u = (uint32_t) __builtin_amdgcn_readlane((int) a0, i);
			const uint32_t v = (uint32_t) __builtin_amdgcn_readlane((int) b0, i);
			int total = 0; // wave-uniform
			for (uint32_t j0 = u; j0 < v; j0 += kWave) {
				const uint32_t j = j0 + lane;
				const bool kept = j < v && (int) a.mapq[j] > a.mq_threshold;
				total += (int) __popcll(__ballot(kept));
			}
			if (lane == i)
				cnt = total;
		}
	}
	if (have && cnt)
		atomicAdd(&a.observed[a.item_iv[item]], cnt);
}

__global__ __launch_bounds__(256) void interval_count_kernel(CountArgs a)
{
	interval_count_body(a, (int64_t) blockIdx.x);
}

// -------------------------------------------------------------------------------------------
// K3' interval_map_rows: the mappability sum of loop C (likelihood.c:113,121) without painting mappability[L].
// For a track whose rows are sorted and at most abutting (what conga_mappability() detects: the bedGraph layout of
// README.md:77-88) the painted value of base x is that of the LAST row with start <= x, if its end >= x
// (svs.c:363-371: end inclusive, later rows overwrite), i.e. row k covers [start_k, min(end_k, start_{k+1} - 1)].
// An item's sum is therefore sum_k val_k * |cover_k  intersected with  [lo, hi)|: float x integer products are exact in
// double, and whenever the per-base partial sums of the reference are exact (k-mer-track values) so is this.
// One lane per item finds its row range with two interleaved binary searches; the wave then walks the 64 ranges
// of its lanes (coalesced row loads, fixed-shape double reduction).  ~3 M rows touched for a 1000G-sized call set
// instead of 11.5 GB painted and 2 GB read back.
// -------------------------------------------------------------------------------------------
struct MapRowsArgs {
	const int32_t *row_start; // concatenated per-chromosome row arrays
	const int32_t *row_end;
	const float *row_val;
	const uint32_t *item_row0; // row range of the item's chromosome (indices into the arrays above)
	const uint32_t *item_row1;
	const uint32_t *row_tile;    // per chromosome: first row (chromosome-local) that starts in 1024-base tile t or later
	const uint32_t *item_rt_off; // where the item's chromosome begins in row_tile
	const int32_t *item_lo;
	const int32_t *item_len;
	const uint8_t *item_has_map; // 1: summed here, 2: summed from the painted track by interval_reduce, 0: no track
	int64_t n_items;
	double *map_part; // [n_items]
};

// items a workgroup of 256 lanes takes: one per 16-lane row
constexpr int kMapRowsItemsPerBlock = 16;

__device__ __forceinline__ void interval_map_rows_body(const MapRowsArgs &a, int64_t block)
{
	// One 16-lane DPP row per item, no traffic between rows: every lane of the row runs the (short) searches, then
	// the row walks the item's rows with stride 16 and sums inside the row.  Many short independent waves hide the
	// memory latency that a wave walking 64 items one after the other could not.
	const int gl = threadIdx.x & 15;
	const int64_t item = block * kMapRowsItemsPerBlock + (threadIdx.x >> 4);
	if (item >= a.n_items || a.item_has_map[item] != 1)
		return; // uniform inside the row (a row never waits for another one below)
	const int32_t lo = a.item_lo[item], hi = lo + a.item_len[item];
	const uint32_t r0 = a.item_row0[item], r1 = a.item_row1[item];
	// the per-tile row index of the track (built once per layout for the painter) narrows both searches to the rows
	// of one 1024-base tile: ~3 probes instead of ~20 over a million rows
	const uint32_t *rt = a.row_tile + a.item_rt_off[item];
	const uint32_t m = r1 - r0;
	const uint32_t ta = (uint32_t) lo >> 10, tb = (uint32_t) hi >> 10;
	uint32_t a0 = r0 + min(rt[ta], m), a1 = r0 + min(rt[ta + 1], m); // first row with start > lo
	uint32_t b0 = r0 + min(rt[tb], m), b1 = r0 + min(rt[tb + 1], m); // first row with start >= hi
	while (a0 < a1 || b0 < b1) {
		const uint32_t ma = a0 + ((a1 - a0) >> 1), mb = b0 + ((b1 - b0) >> 1);
		const int32_t sa = (a0 < a1) ? a.row_start[ma] : 0;
		const int32_t sb = (b0 < b1) ? a.row_start[mb] : 0;
		if (a0 < a1) {
			if (sa <= lo)
				a0 = ma + 1;
			else
				a1 = ma;
		}
		if (b0 < b1) {
			if (sb < hi)
				b0 = mb + 1;
			else
				b1 = mb;
		}
	}
	const uint32_t u = (a0 > r0) ? a0 - 1 : r0; // the last row that starts at or before lo may still cover it
	const uint32_t v = b0;
	double acc = 0.0;
	// eight strided rows per lane are requested before the first is used (128 rows per pass: the whole item in one
	// memory round trip for a 100-mer track); longer row ranges take more passes
	for (uint32_t j0 = u + gl; j0 < v; j0 += 128) {
		int32_t rs[8], re[8], rn[8];
		float rv[8];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const uint32_t j = j0 + 16 * q;
			const bool in = j < v;
			rs[q] = in ? a.row_start[j] : 0;
			re[q] = in ? a.row_end[j] : -1;
			rn[q] = (in && j + 1 < r1) ? a.row_start[j + 1] : INT32_MAX;
			rv[q] = in ? a.row_val[j] : 0.0f;
		}
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const int32_t eff = (re[q] < rn[q] - 1) ? re[q] : rn[q] - 1;
			const int32_t from = (rs[q] > lo) ? rs[q] : lo;
			const int32_t to = (eff < hi - 1) ? eff : hi - 1;
			if (j0 + 16 * q < v && to >= from)
				acc += (double) rv[q] * (double) (to - from + 1);
		}
	}
	// sum over the 16 lanes of the row: row_shr 1, 2, 4, 8 leave the total in the row's last lane (lanes shifted in
	// from outside the row add +0.0)
#pragma unroll
	for (int sh = 0; sh < 4; sh++) {
		const uint64_t bits = __double_as_longlong(acc);
		const uint32_t lo32 = (uint32_t) (sh == 0 ? __builtin_amdgcn_update_dpp(0, (int) (uint32_t) bits, 0x111, 0xF, 0xF, true)
				: sh == 1 ? __builtin_amdgcn_update_dpp(0, (int) (uint32_t) bits, 0x112, 0xF, 0xF, true)
				: sh == 2 ? __builtin_amdgcn_update_dpp(0, (int) (uint32_t) bits, 0x114, 0xF, 0xF, true)
				          : __builtin_amdgcn_update_dpp(0, (int) (uint32_t) bits, 0x118, 0xF, 0xF, true));
		const uint32_t hi32 = (uint32_t) (sh == 0 ? __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (bits >> 32), 0x111, 0xF, 0xF, true)
				: sh == 1 ? __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (bits >> 32), 0x112, 0xF, 0xF, true)
				: sh == 2 ? __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (bits >> 32), 0x114, 0xF, 0xF, true)
				          : __builtin_amdgcn_update_dpp(0, (int) (uint32_t) (bits >> 32), 0x118, 0xF, 0xF, true));
		acc += __longlong_as_double((long long) (((uint64_t) hi32 << 32) | lo32));
	}
	if (gl == 15)
		a.map_part[item] = acc;
}

__global__ __launch_bounds__(256) void interval_map_rows_kernel(MapRowsArgs a)
{
	interval_map_rows_body(a, (int64_t) blockIdx.x);
}

// K0' and K4' in one launch (the step is launch-gap-bound: every dependent launch costs ~10 us on this stack, and a
// second stream's event dependency costs more than it hides): the first count_blocks workgroups search and count,
// the rest stream the tuples.  The two halves touch disjoint outputs and only read the tuples.
template <bool MAP_ROWS, bool ALL> __global__ __launch_bounds__(kTupleBlock, 8) void tuple_pass_kernel(TupleArgs a, CountArgs c,
		int count_blocks, MapRowsArgs m, int map_blocks, int ingest_blocks)
{
	__shared__ uint32_t hist[kHistCopies * kGcBins];
	__shared__ uint32_t kept_block;
	// The ingest workgroups come first: their grid is one resident wave of workgroups, and a single one left waiting
	// for a slot behind the (short) search workgroups would stretch the launch by a whole workgroup lifetime.
	const int b = (int) blockIdx.x;
	if (b < ingest_blocks)
		ingest_tuples_body<ALL>(a, (uint32_t) b, hist, kept_block);
	else if (b < ingest_blocks + count_blocks)
		interval_count_body(c, (int64_t) (b - ingest_blocks));
	else if (MAP_ROWS)
		interval_map_rows_body(m, (int64_t) (b - ingest_blocks - count_blocks));
}

// -------------------------------------------------------------------------------------------
// K1 + K2 depth_tile: builds read_depth and the GC-stratified depth sums in one pass.
//
// Replaces init_rd_per_chr's memset (read_distribution.c:16-17), the increments of
// count_reads_bam (bam_data.c:205-215) and the depth side of both loops of calc_mean_per_chr
// (read_distribution.c:33-37,63-73).
//
// Every WAVE is an independent worker (no workgroup barrier anywhere): it owns a contiguous run of
// 2048-position tiles (4 KiB of int16, 4 KiB-aligned in HBM; tiles are NOT aligned to GC windows -- a
// window that straddles two tiles adds both parts into the same bin) and a private LDS area.  Per tile it zeroes 16-bit counters
// in LDS (two per dword -- the layout of the int16 output), adds the tile's tuples with LDS
// atomics, then streams the tile out ONCE with one ds_read_b128 + one 16-byte global store per lane
// (1 KiB per wave-instruction; read_depth is written exactly once and never read back for the
// histogram).  Per-window sums feed a private 101-bin histogram that is flushed with one global
// atomic per non-empty bin when the wave moves to another chromosome.  LDS operations of one wave
// execute in order, so the phases need no barrier; 28 such waves per CU hide each other's
// tile-index / tuple / GC-byte load latencies.  All accumulators are integers, so the result does
// not depend on the order of the atomics.
// `short` semantics: a counter wraps modulo 2^16 exactly like read_depth[pos]++ on a short
// (gcc); the carry of a wrapping low half into its neighbour is undone on the spot.
// -------------------------------------------------------------------------------------------
constexpr int kDepthBlock = 256;
constexpr int kDepthWaves = kDepthBlock / kWave;
constexpr int kDepthMaxTile = 2048; // positions per wave tile: 4 KiB of packed LDS counters = 4 KiB-aligned stores
constexpr int kDepthMaxWin = 64;    // GC windows (or parts of windows) a tile may touch
constexpr int kDepthTilesPerBlock = 32; // tiles per workgroup (128 KiB of read_depth, one histogram flush); 32-64 measured best

// One workgroup = one contiguous range of tiles inside ONE chromosome (table built on the host).
struct DepthBlock {
	int32_t slot;
	int32_t n_tiles;
	int64_t first_tile; // global tile index
};

struct DepthArgs {
	const int32_t *pos;
	const uint8_t *mapq;
	const uint32_t *tile_first; // [total_tiles + 1], global tuple indices (ingest_kernel)
	int16_t *rd;
	const uint8_t *gc_hist;
	const Slot *slots;
	const DepthBlock *blocks;
	Small *small;
	int32_t step;
	uint32_t step_magic; // floor(2^32 / step) + 1: j / step == umulhi(j, magic) for j * step < 2^32
	int32_t tile_len;    // positions per tile: a multiple of 8, NOT of step -- a window may straddle two tiles,
	                     // both parts then add into the same GC bin
	int32_t mq_threshold;
	uint32_t n_total;    // tuples in the batch
	int64_t total_tiles;
};

constexpr int kDepthPrefetch = 2; // tuples per lane fetched one tile ahead (128 per 2048-base tile = 6x coverage)

__global__ __launch_bounds__(kDepthBlock) void depth_tile_kernel(DepthArgs a)
{
	__shared__ __attribute__((aligned(16))) uint32_t cnt2_all[kDepthWaves][kDepthMaxTile / 2];
	__shared__ int32_t wsum_all[kDepthWaves][kDepthMaxWin + 8];
	__shared__ unsigned long long h_sum[kGcBins]; // shared by the workgroup's waves (LDS atomics)
	__shared__ unsigned int h_counted;

	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
	const int lane = threadIdx.x & (kWave - 1);
	uint32_t *cnt2 = cnt2_all[wv];
	int32_t *wsum = wsum_all[wv];

	for (int g = threadIdx.x; g < kGcBins; g += kDepthBlock)
		h_sum[g] = 0;
	if (threadIdx.x == 0)
		h_counted = 0;
	__syncthreads();

	const DepthBlock blk = a.blocks[blockIdx.x];
	const int s = blk.slot;
	const Slot sl = a.slots[s];
	const int T = a.tile_len;
	const bool skip = (a.small[s].status & kStatusUnsorted) != 0; // then this chromosome's tile index is meaningless
	const int64_t g_end = blk.first_tile + blk.n_tiles;
	unsigned int counted = 0;

	auto first_of = [&](int64_t t) -> uint32_t {
		const uint32_t v = a.tile_first[(t < a.total_tiles) ? t : a.total_tiles];
		return (v == 0xFFFFFFFFu) ? a.n_total : v;
	};
	auto gc_of = [&](int64_t gt_) -> uint32_t {
		// GC byte of the lane-th window touched by global tile gt_ (lanes past the tile's last window are unused)
		int64_t wg = (uint32_t) ((gt_ - sl.tile0) * T) / (uint32_t) a.step + lane;
		if (wg >= sl.n_win)
			wg = sl.n_win - 1;
		return a.gc_hist[sl.gc_off + wg];
	};
	auto load_tuples = [&](uint32_t lo, uint32_t hi, int32_t *p_out, int *q_out) {
#pragma unroll
		for (int k = 0; k < kDepthPrefetch; k++) {
			const uint32_t i = lo + k * kWave + lane;
			p_out[k] = (i < hi) ? a.pos[i] : INT32_MIN; // INT32_MIN: no tuple (fails the range check)
			q_out[k] = (i < hi) ? (int) a.mapq[i] : 0;
		}
	};

	// The waves of the workgroup alternate tiles of its range (tile = first + wave, + 4, + 8, ...), and workgroups
	// are dispatched in order over the genome: the chip writes one advancing front of read_depth rather than
	// thousands of separate streams (tools/membw.hip: 5.9 vs 5.6 TB/s with pure stores).
	// Software pipeline: tile index two of this wave's tiles ahead, tuples and GC bytes one ahead.
	int64_t gt = blk.first_tile + wv;
	if (gt < g_end) {
		uint32_t c_lo = first_of(gt), c_hi = first_of(gt + 1);
		uint32_t n_lo = first_of(gt + kDepthWaves), n_hi = first_of(gt + kDepthWaves + 1);
		int32_t c_pos[kDepthPrefetch];
		int c_mq[kDepthPrefetch];
		load_tuples(c_lo, c_hi, c_pos, c_mq);
		uint32_t c_gc = gc_of(gt);

		for (; gt < g_end; gt += kDepthWaves) {
			const bool have_next = gt + kDepthWaves < g_end;
			const uint32_t nn_lo = first_of(gt + 2 * kDepthWaves), nn_hi = first_of(gt + 2 * kDepthWaves + 1);
			int32_t n_pos[kDepthPrefetch];
			int n_mq[kDepthPrefetch];
			load_tuples(n_lo, have_next ? n_hi : n_lo, n_pos, n_mq);
			const uint32_t n_gc = have_next ? gc_of(gt + kDepthWaves) : 0;

			// ---- tile gt
			const int64_t tile = gt - sl.tile0;
			const int64_t base = tile * T;
			const int len = (int) ((sl.L - base < T) ? (sl.L - base) : T);
			// offset of the tile's first base inside its GC window (one 32-bit division per tile)
			const int r0 = (int) ((uint32_t) base - ((uint32_t) base / (uint32_t) a.step) * (uint32_t) a.step);

			for (int j = lane * 4; j < T / 2; j += kWave * 4)
				*reinterpret_cast<uint4 *>(&cnt2[j]) = make_uint4(0, 0, 0, 0);
			wsum[lane] = 0; // kDepthMaxWin == kWave
			__builtin_amdgcn_wave_barrier();

			auto add_tuple = [&](int32_t pp, int mq) {
				const int64_t p = (int64_t) pp - base;
				if (p >= 0 && p < len && mq > a.mq_threshold) {
					if (p & 1)
						atomicAdd(&cnt2[p >> 1], 0x10000u);
					else {
						const uint32_t old = atomicAdd(&cnt2[p >> 1], 1u);
						if ((old & 0xFFFFu) == 0xFFFFu)
							atomicSub(&cnt2[p >> 1], 0x10000u); // the low short wrapped: undo its carry
					}
					counted++;
				}
			};
			if (!skip) {
#pragma unroll
				for (int k = 0; k < kDepthPrefetch; k++)
					add_tuple(c_pos[k], c_mq[k]); // lanes past c_hi carry INT32_MIN and fail the range check
				for (uint32_t i = c_lo + kDepthPrefetch * kWave + lane; i < c_hi; i += kWave) // deep tiles: direct loads
					add_tuple(a.pos[i], (int) a.mapq[i]);
			}
			__builtin_amdgcn_wave_barrier();

			// stream the tile out: 8 positions (16 bytes) per lane per step; the slot's region is padded to whole
			// tiles and positions >= len hold zeros, so the last store may run over len
			int16_t *out = a.rd + sl.rd_off + base;
			for (int j = lane * 8; j < len; j += kWave * 8) {
				const uint4 q = *reinterpret_cast<const uint4 *>(&cnt2[j >> 1]);
				*reinterpret_cast<uint4 *>(out + j) = q;
				if ((q.x | q.y | q.z | q.w) != 0u) { // sparse: most 8-base groups hold no read start
					const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
					const int t0 = j + r0; // position relative to the start of the tile's first window
					int w = (a.step == 1) ? t0 : (int) __umulhi((uint32_t) t0, a.step_magic);
					int r = t0 - w * a.step;
					int acc = 0;
#pragma unroll
					for (int e = 0; e < 8; e++) {
						const uint32_t half = (e & 1) ? (wd[e >> 1] >> 16) : (wd[e >> 1] & 0xFFFFu);
						acc += (int) (int16_t) half;
						if (++r == a.step) {
							if (acc)
								atomicAdd(&wsum[w], acc);
							acc = 0;
							r = 0;
							w++;
						}
					}
					if (acc)
						atomicAdd(&wsum[w], acc);
				}
			}
			__builtin_amdgcn_wave_barrier();

			// per-window (or window-part) depth sums -> GC bins (read_distribution.c:70-71)
			const int nw = (a.step == 1) ? len : (int) __umulhi((uint32_t) (len + r0 + a.step - 1), a.step_magic);
			if (lane < nw) {
				const int sw = wsum[lane];
				if (sw && c_gc < (uint32_t) kGcBins)
					atomicAdd(&h_sum[c_gc], (unsigned long long) (long long) sw);
			}
			__builtin_amdgcn_wave_barrier();

			// ---- rotate the pipeline
			c_lo = n_lo;
			c_hi = n_hi;
			n_lo = nn_lo;
			n_hi = nn_hi;
#pragma unroll
			for (int k = 0; k < kDepthPrefetch; k++) {
				c_pos[k] = n_pos[k];
				c_mq[k] = n_mq[k];
			}
			c_gc = n_gc;
		}
	}
	const int w = wave_sum_i32((int) counted);
	if (lane == 0 && w)
		atomicAdd(&h_counted, (unsigned int) w);
	__syncthreads(); // the only workgroup barrier: every wave's tiles are in the shared histogram
	for (int g = threadIdx.x; g < kGcBins; g += kDepthBlock)
		if (h_sum[g])
			atomicAdd(&a.small[s].hist_sum[g], h_sum[g]);
	if (threadIdx.x == 0 && h_counted)
		atomicAdd(&a.small[s].counters[CNT_COUNTED], (unsigned long long) h_counted);
}

// window_per_gc (read_distribution.c:72) depends on the annotation only: one thread per GC window,
// run once per layout (not per compute).
__global__ __launch_bounds__(256) void gc_bases_kernel(const uint8_t *__restrict__ gc_hist,
		const Slot *__restrict__ slots, int n_slots, int32_t step, unsigned long long *__restrict__ bases /* [n_slots][101] */)
{
	__shared__ unsigned int h[kGcBins];
	const int s = blockIdx.y;
	const Slot sl = slots[s];
	for (int g = threadIdx.x; g < kGcBins; g += blockDim.x)
		h[g] = 0;
	__syncthreads();
	const int64_t n_w = (sl.L + step - 1) / step;
	for (int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; w < n_w; w += (int64_t) gridDim.x * blockDim.x) {
		const int64_t b = w * step;
		const int nb = (int) ((sl.L - b < step) ? (sl.L - b) : step);
		const int g = gc_hist[sl.gc_off + ((w < sl.n_win) ? w : sl.n_win - 1)];
		if (g < kGcBins)
			atomicAdd(&h[g], (unsigned int) nb);
	}
	__syncthreads();
	for (int g = threadIdx.x; g < kGcBins; g += blockDim.x)
		if (h[g])
			atomicAdd(&bases[(int64_t) s * kGcBins + g], (unsigned long long) h[g]);
}

// Depth for unsorted input (CONGA_FLAG_READS_UNSORTED): read_depth is zeroed by hipMemsetAsync and
// incremented with global atomics on the containing 32-bit word.  One launch per chromosome.
__global__ __launch_bounds__(256) void depth_atomic_kernel(const int32_t *__restrict__ pos,
		const uint8_t *__restrict__ mapq, int64_t n, int64_t L, int32_t mq_threshold, int16_t *rd,
		unsigned long long *counters)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	unsigned int counted = 0;
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const int64_t p = pos[i];
		if (p < 0 || p >= L) {
			atomicAdd(&counters[CNT_OUT_OF_RANGE], 1ull);
			continue;
		}
		if ((int) mapq[i] <= mq_threshold)
			continue;
		unsigned int *word = reinterpret_cast<unsigned int *>(rd) + (p >> 1);
		if (p & 1)
			atomicAdd(word, 0x10000u);
		else {
			const unsigned int old = atomicAdd(word, 1u);
			if ((old & 0xFFFFu) == 0xFFFFu)
				atomicSub(word, 0x10000u); // undo the carry: the low short wrapped to 0
		}
		counted++;
	}
	int w = wave_sum_i32((int) counted);
	if ((threadIdx.x & (kWave - 1)) == 0 && w)
		atomicAdd(&counters[CNT_COUNTED], (unsigned long long) w);
}

// GC histogram from a finished read_depth (used after depth_atomic_kernel only).
__global__ __launch_bounds__(256) void gc_hist_kernel(const int16_t *__restrict__ rd, int64_t L,
		const uint8_t *__restrict__ gc_hist, int64_t n_win, int32_t step, unsigned long long *hist_sum,
		unsigned long long *hist_bases)
{
	__shared__ unsigned long long h_sum[kGcBins];
	__shared__ unsigned int h_bases[kGcBins];
	for (int g = threadIdx.x; g < kGcBins; g += blockDim.x) {
		h_sum[g] = 0;
		h_bases[g] = 0;
	}
	__syncthreads();
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	const int64_t n_w = (L + step - 1) / step;
	for (int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; w < n_w; w += stride) {
		const int64_t b = w * step;
		const int nb = (int) ((L - b < step) ? (L - b) : step);
		long long s = 0;
		for (int e = 0; e < nb; e++)
			s += rd[b + e];
		const int g = gc_hist[(w < n_win) ? w : n_win - 1];
		if (g < kGcBins) {
			if (s)
				atomicAdd(&h_sum[g], (unsigned long long) s);
			atomicAdd(&h_bases[g], (unsigned int) nb);
		}
	}
	__syncthreads();
	for (int g = threadIdx.x; g < kGcBins; g += blockDim.x) {
		if (h_sum[g])
			atomicAdd(&hist_sum[g], h_sum[g]);
		if (h_bases[g])
			atomicAdd(&hist_bases[g], (unsigned long long) h_bases[g]);
	}
}

// -------------------------------------------------------------------------------------------
// K2b expected_table (one workgroup per chromosome):
// expected_read_depth[g] = (float) rd_per_gc[g] / window_per_gc[g] (read_distribution.c:75-83):
// float(long) / float(int) in single precision, [0] forced to 0, NaN / +-inf -> 0.  Integer ->
// float goes through double (exact below 2^53, then one rounding), which equals the correctly
// rounded direct conversion.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ float expected_value(unsigned long long sum, unsigned long long bases, int g)
{
	if (g == 0)
		return 0.0f;
	const float num = (float) (double) (long long) sum;
	const float den = (float) (double) (int) bases;
	const float e = num / den;
	return (isnan(e) || isinf(e)) ? 0.0f : e;
}

__device__ __forceinline__ void expected_table_body(Small *__restrict__ small, const unsigned long long *__restrict__ bases,
		Small *__restrict__ host_small, int slot)
{
	Small &sm = small[slot];
	const int g = threadIdx.x;
	if (host_small) { // pinned host copy of the block, written by the kernel instead of a separate copy
		Small &hs = host_small[slot];
		if (g == 0) {
			hs.status = sm.status;
			hs.pad = 0;
			hs.pad2 = 0.0f;
		}
		if (g < CNT_N)
			hs.counters[g] = sm.counters[g];
	}
	if (g >= kGcBins)
		return;
	if (bases)
		sm.hist_bases[g] = bases[(int64_t) slot * kGcBins + g];
	const float e = expected_value(sm.hist_sum[g], bases ? bases[(int64_t) slot * kGcBins + g] : sm.hist_bases[g], g);
	sm.E[g] = e;
	if (host_small) {
		Small &hs = host_small[slot];
		hs.hist_sum[g] = sm.hist_sum[g];
		hs.hist_bases[g] = sm.hist_bases[g];
		hs.E[g] = e;
	}
}

__global__ void expected_table_kernel(Small *__restrict__ small, const unsigned long long *__restrict__ bases,
		Small *__restrict__ host_small)
{
	expected_table_body(small, bases, host_small, (int) blockIdx.x);
}

// -------------------------------------------------------------------------------------------
// K3 mappability paint (one launch per chromosome that has a track).  Semantics
// (svs.c:363-371): rows in file order, END INCLUSIVE, a later row overwrites an earlier one.
//
// paint_sorted_kernel: rows sorted by start with row k+1 starting at or after row k's end (the
// bedGraph-like layout of README.md:77-88; abutting rows share one base, which the later row
// wins).  A base x is covered by the LAST row with start <= x, if that row's end >= x.  One pass,
// every float written exactly once, no memset.
// paint_winner_kernel / paint_resolve_kernel: any row order -- atomicMax of the row index per
// base, then map[i] = val[winner[i]].
// -------------------------------------------------------------------------------------------
// row_tile_first[t] = first row whose start lies in paint tile t or later (rows sorted by start); entries past
// the last row keep the memset value 0xFFFFFFFF (= "m").  Built once per layout.
__global__ __launch_bounds__(256) void row_tile_index_kernel(const int32_t *__restrict__ start, int64_t m,
		int32_t tile_len, int64_t n_tiles, uint32_t *__restrict__ row_tile_first)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	for (int64_t k = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; k < m; k += stride) {
		int64_t t_cur = (start[k] < 0) ? 0 : (int64_t) start[k] / tile_len;
		int64_t t_prev = (k == 0) ? -1 : ((start[k - 1] < 0) ? 0 : (int64_t) start[k - 1] / tile_len);
		if (t_cur > n_tiles)
			t_cur = n_tiles;
		if (t_prev > n_tiles)
			t_prev = n_tiles;
		for (int64_t t = t_prev + 1; t <= t_cur; t++)
			row_tile_first[t] = (uint32_t) k;
	}
}

// One wave paints one 1024-base tile: row starts are marked in LDS (atomicMax of the row index, so the later
// of two rows with equal starts wins), a running maximum over the marks gives every base the LAST row that
// starts at or before it, and that row covers the base iff its end >= base (rows abut at most, so no earlier
// row can reach further).  16 consecutive bases per lane; a DPP max-scan carries the running row across lanes.
constexpr int kPaintTile = 1024;

__device__ __forceinline__ int wave_excl_scan_max_i32(int v, int carry)
{
	// inclusive max-scan through DPP, then shift by one lane
#define CONGA_DPP_MAX(ctrl, rows) { const int t = __builtin_amdgcn_update_dpp(INT32_MIN, v, ctrl, rows, 0xF, false); v = (t > v) ? t : v; }
	CONGA_DPP_MAX(0x111, 0xF)
	CONGA_DPP_MAX(0x112, 0xF)
	CONGA_DPP_MAX(0x114, 0xF)
	CONGA_DPP_MAX(0x118, 0xF)
	CONGA_DPP_MAX(0x142, 0xA)
	CONGA_DPP_MAX(0x143, 0xC)
#undef CONGA_DPP_MAX
	int up = __shfl_up(v, 1, kWave);
	if ((threadIdx.x & (kWave - 1)) == 0)
		up = INT32_MIN;
	return (up > carry) ? up : carry;
}

constexpr int kPaintTilesPerBlock = 32; // 128 KiB of floats per workgroup, workgroups dispatched in genome order

__global__ __launch_bounds__(256) void paint_sorted_kernel(const int32_t *__restrict__ start,
		const int32_t *__restrict__ end, const float *__restrict__ val, int64_t m,
		const uint32_t *__restrict__ row_tile_first, float *__restrict__ map, int64_t L)
{
	__shared__ __attribute__((aligned(16))) int32_t mark_all[4][kPaintTile];
	__shared__ int32_t row_end_all[4][kWave];
	__shared__ float row_val_all[4][kWave];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
	int32_t *mark = mark_all[wv];
	int32_t *row_end = row_end_all[wv];
	float *row_val = row_val_all[wv];
	const int64_t n_tiles = (L + kPaintTile - 1) / kPaintTile;
	const int64_t t_begin = (int64_t) blockIdx.x * kPaintTilesPerBlock;
	const int64_t t_end = (t_begin + kPaintTilesPerBlock < n_tiles) ? t_begin + kPaintTilesPerBlock : n_tiles;

	auto first_row = [&](int64_t t) -> uint32_t {
		const uint32_t v = row_tile_first[(t < n_tiles) ? t : n_tiles];
		return (v == 0xFFFFFFFFu) ? (uint32_t) m : v;
	};
	// the rows a tile can need: the last one that starts before it (k0 - 1) and those that start inside it;
	// one row per lane is fetched a tile ahead (tiles with more than 63 starting rows fall back to direct loads)
	struct RowRegs {
		int32_t s, e;
		float v;
	};
	auto load_row = [&](uint32_t k0, uint32_t k1) -> RowRegs {
		RowRegs r = {INT32_MIN, -1, 0.0f};
		const int64_t k = (int64_t) k0 - 1 + lane;
		if (k >= 0 && k < (int64_t) k1) {
			r.s = start[k];
			r.e = end[k];
			r.v = val[k];
		}
		return r;
	};

	// the four waves of the workgroup alternate tiles; tile index two of this wave's tiles ahead, rows one ahead
	int64_t tile = t_begin + wv;
	if (tile >= t_end)
		return;
	uint32_t c_k0 = first_row(tile), c_k1 = first_row(tile + 1);
	uint32_t n_k0 = first_row(tile + 4), n_k1 = first_row(tile + 5);
	RowRegs c_row = load_row(c_k0, c_k1);
	for (; tile < t_end; tile += 4) {
		const bool have_next = tile + 4 < t_end;
		const uint32_t nn_k0 = first_row(tile + 8), nn_k1 = first_row(tile + 9);
		RowRegs n_row = {INT32_MIN, -1, 0.0f};
		if (have_next)
			n_row = load_row(n_k0, n_k1);

		const int64_t base = tile * kPaintTile;
		const uint32_t k0 = c_k0, k1 = c_k1;
		const bool in_lds = (int64_t) k1 - ((int64_t) k0 - 1) <= kWave; // every needed row sits in a lane
		for (int j = lane * 4; j < kPaintTile; j += kWave * 4)
			*reinterpret_cast<int4 *>(&mark[j]) = make_int4(-1, -1, -1, -1);
		row_end[lane] = c_row.e;
		row_val[lane] = c_row.v;
		__builtin_amdgcn_wave_barrier();
		if (in_lds) {
			// lane l holds row k0 - 1 + l; lane 0's row starts before the tile and only seeds the carry
			const int64_t k = (int64_t) k0 - 1 + lane;
			if (lane > 0 && k < (int64_t) k1) {
				int64_t x = (int64_t) c_row.s - base;
				if (x < 0)
					x = 0; // a row that starts before base 0 is in force from the first base
				if (x < kPaintTile)
					atomicMax(&mark[x], (int32_t) k);
			}
		} else {
			for (uint32_t k = k0 + lane; k < k1; k += kWave) {
				int64_t x = (int64_t) start[k] - base;
				if (x < 0)
					x = 0;
				if (x < kPaintTile)
					atomicMax(&mark[x], (int32_t) k);
			}
		}
		__builtin_amdgcn_wave_barrier();
		// the row in force when the tile begins: the last row that starts before it
		int carry = (int) k0 - 1;
		const int row0 = (int) k0 - 1; // row held by lane 0
		// four passes of 256 bases: lane l owns bases [4 (64 q + l), +4), so every store instruction of the
		// wave writes 1 KiB contiguously
#pragma unroll
		for (int q = 0; q < kPaintTile / 256; q++) {
			const int seg = q * kWave + lane;
			const int4 mk = *reinterpret_cast<const int4 *>(&mark[seg * 4]);
			int lane_max = (mk.x > mk.y) ? mk.x : mk.y;
			lane_max = (mk.z > lane_max) ? mk.z : lane_max;
			lane_max = (mk.w > lane_max) ? mk.w : lane_max;
			int cur = wave_excl_scan_max_i32(lane_max, carry);
			const int m4[4] = {mk.x, mk.y, mk.z, mk.w};
			int cached = -2;
			int64_t c_end = -1;
			float c_val = 0.0f, out[4];
			const int64_t x0 = base + seg * 4;
#pragma unroll
			for (int e = 0; e < 4; e++) {
				cur = (m4[e] > cur) ? m4[e] : cur;
				if (cur != cached) {
					cached = cur;
					if (cur >= 0) {
						if (in_lds) {
							c_end = row_end[cur - row0];
							c_val = row_val[cur - row0];
						} else {
							c_end = end[cur];
							c_val = val[cur];
						}
					}
				}
				out[e] = (cur >= 0 && c_end >= x0 + e) ? c_val : 0.0f;
			}
			if (x0 + 4 <= L)
				*reinterpret_cast<float4 *>(map + x0) = make_float4(out[0], out[1], out[2], out[3]);
			else
				for (int e = 0; e < 4 && x0 + e < L; e++)
					map[x0 + e] = out[e];
			// the row in force after this pass = the running row of the last lane
			carry = __builtin_amdgcn_readlane(cur, kWave - 1);
		}
		__builtin_amdgcn_wave_barrier();
		c_k0 = n_k0;
		c_k1 = n_k1;
		n_k0 = nn_k0;
		n_k1 = nn_k1;
		c_row = n_row;
	}
}

__global__ __launch_bounds__(256) void paint_winner_kernel(const int32_t *__restrict__ start,
		const int32_t *__restrict__ end, int64_t m, int32_t *__restrict__ winner, int64_t L)
{
	// one wave per row; lanes stride over the row's bases
	const int lane = threadIdx.x & (kWave - 1);
	const int64_t wave = (int64_t) blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane((int) (threadIdx.x / kWave));
	const int64_t n_waves = (int64_t) gridDim.x * blockDim.x / kWave;
	for (int64_t r = wave; r < m; r += n_waves) {
		int64_t s = start[r], e = end[r];
		if (s < 0)
			s = 0;
		if (e > L - 1)
			e = L - 1;
		for (int64_t x = s + lane; x <= e; x += kWave)
			atomicMax(&winner[x], (int32_t) r);
	}
}

__global__ __launch_bounds__(256) void paint_resolve_kernel(const int32_t *__restrict__ winner,
		const float *__restrict__ val, float *__restrict__ map, int64_t L)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < L; i += stride) {
		const int32_t w = winner[i];
		map[i] = (w >= 0) ? val[w] : 0.0f;
	}
}

// -------------------------------------------------------------------------------------------
// K4 (memory side) interval_reduce: observed_rd += read_depth[i] (int) and
// mappability_score += mappability[i] (double) over [start, end) (likelihood.c:120-123).
//
// Intervals are cut on the host into work items of at most kItemLen bases; one wave per item,
// 16-byte loads per lane (8 depths / 4 mappability floats), wave reduction through shuffles.
// The integer sum is order-free (atomicAdd into observed[iv]).  The double sum is kept
// deterministic: each item writes its partial to map_part[item] and the scoring kernel adds an
// interval's partials in item order.  Whenever every partial sum is exactly representable
// (mappability values with few mantissa bits -- the usual 1, 0.5, 0.25 ... of a k-mer track) the
// result equals the reference's serial sum bit for bit; otherwise it differs by rounding only.
// -------------------------------------------------------------------------------------------
constexpr int kItemLen = 16384;

struct ReduceArgs {
	const int16_t *rd;
	const float *map; // concatenated like rd; only read for items whose slot has a track
	const int64_t *item_off;   // global element offset of the item's first base
	const int32_t *item_len;
	const int32_t *item_iv;
	const uint8_t *item_has_map;
	int64_t n_items;
	int32_t *observed;  // [n_iv], zeroed before launch
	double *map_part;   // [n_items]
};

__global__ __launch_bounds__(256) void interval_reduce_kernel(ReduceArgs a)
{
	const int lane = threadIdx.x & (kWave - 1);
	// (the wave's item: the same in all its lanes -- said so, its bounds are scalars and their loads scalar loads)
	const int64_t item = (int64_t) blockIdx.x * (blockDim.x / kWave) + __builtin_amdgcn_readfirstlane((int) (threadIdx.x / kWave));
	if (item >= a.n_items)
		return; // wave-uniform
	const int64_t s = a.item_off[item], e = s + a.item_len[item];

	// ---- depth: int16, 8 per lane (skipped when the tuple-space formulation counts reads instead)
	int acc = 0;
	if (a.rd) {
		int64_t head = (s + 7) & ~(int64_t) 7; // first 16-byte aligned index
		if (head > e)
			head = e;
		const int64_t body_end = head + ((e - head) & ~(int64_t) 7);
		for (int64_t i = s + lane; i < head; i += kWave)
			acc += a.rd[i];
		for (int64_t i = head + (int64_t) lane * 8; i < body_end; i += kWave * 8) {
			const uint4 q = *reinterpret_cast<const uint4 *>(a.rd + i);
			const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
			for (int k = 0; k < 4; k++) {
				acc += (int) (int16_t) (w[k] & 0xFFFFu);
				acc += (int) (int16_t) (w[k] >> 16);
			}
		}
		for (int64_t i = body_end + lane; i < e; i += kWave)
			acc += a.rd[i];
	}
	acc = wave_sum_i32(acc);
	if (lane == 0 && acc)
		atomicAdd(&a.observed[a.item_iv[item]], acc);

	// ---- mappability: float, 4 per lane, summed in double
	if (a.item_has_map[item] == 2) { // 1 = summed in row space by interval_map_rows
		double m = 0.0;
		int64_t head = (s + 3) & ~(int64_t) 3;
		if (head > e)
			head = e;
		const int64_t body_end = head + ((e - head) & ~(int64_t) 3);
		for (int64_t i = s + lane; i < head; i += kWave)
			m += (double) a.map[i];
		for (int64_t i = head + (int64_t) lane * 4; i < body_end; i += kWave * 4) {
			const float4 q = *reinterpret_cast<const float4 *>(a.map + i);
			m += (double) q.x;
			m += (double) q.y;
			m += (double) q.z;
			m += (double) q.w;
		}
		for (int64_t i = body_end + lane; i < e; i += kWave)
			m += (double) a.map[i];
		m = wave_sum_f64(m);
		if (lane == 0)
			a.map_part[item] = m;
	}
}

// -------------------------------------------------------------------------------------------
// K5 scoring: lpoisson x3, the int-truncated max, the c-score and the CN call (likelihood.c:96-105,131-168).
// Used by interval_score_kernel (one lane per interval) and, when every input of an interval is final before its
// chain starts (tuple space, no mappability track), by the chain kernel itself: the lane that finishes an
// interval's chain scores it and writes the record to HBM and straight into the caller's pinned host buffer: no
// score launch, no copy launch, no gaps between them.  (The stores reach the host when the kernel ends -- measured,
// tools/hostwrite.hip -- so the PCIe time itself is not hidden.)
// -------------------------------------------------------------------------------------------
struct ScoreArgs {
	const int32_t *start;
	const int32_t *end;
	const uint8_t *type;     // 'D' / 'E' per interval
	int64_t n_iv;
	const int32_t *observed;
	const float *expected;
	const double *map_part;
	const int32_t *item_first; // [n_iv + 1] first work item of each interval
	const uint8_t *iv_has_map;
	const int32_t *support;  // split-read support per interval (may be null)
	conga_result *out;
};

// lgamma(n + 1) = ln(n!) for the integer argument lpoisson passes (likelihood.c:104).  A correctly rounded table
// below 32, Stirling's series above (x = n + 1 >= 33: the first omitted term, 1 / (1188 x^9), is below 2e-17, and
// the rounding of (x - 0.5) ln x stays ~1e-10 even for a million reads -- the bar is 1e-6 absolute).  A fifth of
// the instructions of the general routine, once per interval instead of three times.  Negative n (a wrapped depth
// counter) goes to the library routine.
__device__ __forceinline__ double log_factorial(int n)
{
	static const double kTable[32] = {0, 0, 0.69314718055994495, 1.7917594692280554, 3.1780538303479449, 4.7874917427820467, 6.5792512120101021, 8.5251613610654147, 10.604602902745249, 12.801827480081467, 15.104412573075514, 17.502307845873887, 19.987214495661885, 22.552163853123421, 25.191221182738683, 27.89927138384089, 30.671860106080672, 33.505073450136891, 36.395445208033053, 39.339884187199495, 42.335616460753485, 45.380138898476908, 48.47118135183522, 51.606675567764377, 54.784729398112319, 58.003605222980518, 61.261701761002008, 64.557538627006338, 67.889743137181526, 71.257038967168, 74.658236348830172, 78.092223553315307};
	if (n < 0)
		return lgamma((double) n + 1.0);
	if (n < 32)
		return kTable[n];
	const double x = (double) n + 1.0;
	const double r = 1.0 / x, r2 = r * r;
	const double series = r * (1.0 / 12.0 + r2 * (-1.0 / 360.0 + r2 * (1.0 / 1260.0 + r2 * (-1.0 / 1680.0))));
	return (x - 0.5) * log(x) - x + 0.91893853320467274178 + series;
}

__device__ __forceinline__ double lpoisson_dev(int observed, double lambda, double lfact)
{
	// likelihood.c:96-105; lfact = lgamma(observed + 1)
	if (lambda == 0.0)
		lambda = 0.01;
	return (double) observed * log(lambda) - lambda - lfact;
}

__device__ __forceinline__ int trunc_max(double x, double y)
{
	// common.c:262-268: max() takes ints, so both arguments are truncated toward zero first
	const int xi = (int) x, yi = (int) y;
	return (xi < yi) ? yi : xi;
}

// One interval's record from its two reduced depths (likelihood.c:131-168).
__device__ __forceinline__ conga_result score_interval(const ScoreArgs &a, int64_t iv, float expected)
{
	const int observed = a.observed[iv];
	const uint8_t type = a.type[iv];
	const double ex = (double) expected;

	conga_result r;
	r.rp = 0;
	r.border_rp = 0;
	r.reserved = 0;
	r.mappability = 0.0;
	const double lfact = log_factorial(observed);
	if (type == CONGA_DELETION) {
		r.lhomo = lpoisson_dev(observed, 0.0, lfact);
		r.lhetero = lpoisson_dev(observed, 0.5 * ex, lfact);
		r.lnone = lpoisson_dev(observed, ex, lfact);
		r.copy_number = ((float) observed < (expected / 4.0f)) ? 2 : 1; // likelihood.c:146
	} else {
		r.lhomo = lpoisson_dev(observed, (double) (2.0f * expected), lfact); // int * float stays float
		r.lhetero = lpoisson_dev(observed, 1.5 * ex, lfact);
		r.lnone = lpoisson_dev(observed, ex, lfact);
		r.copy_number = (r.lhomo > r.lhetero) ? 2 : 1; // likelihood.c:164
	}
	r.score = (double) trunc_max(r.lhomo, r.lhetero) / r.lnone; // likelihood.c:138,160
	r.observed = observed;
	r.expected = expected;
	if (a.iv_has_map[iv]) {
		double ms = 0.0;
		for (int32_t it = a.item_first[iv]; it < a.item_first[iv + 1]; it++)
			ms += a.map_part[it];
		r.mappability = ms / (double) ((int64_t) a.end[iv] - a.start[iv]); // likelihood.c:128
	}
	if (a.support) {
		if (type == CONGA_DELETION)
			r.border_rp = a.support[iv];
		else
			r.rp = a.support[iv];
	}
	return r;
}

// -------------------------------------------------------------------------------------------
// K4 (chain) interval_chain: the serial float32 accumulation `expected_rd += E[gc]`
// (likelihood.c:111,115-119), bit-exact.  One launch, four classes of intervals (order[] is sorted by the
// number of GC windows, longest first), so the few long chains run beside the many short ones:
//   A+ more than kChainBlockWindows windows  one WORKGROUP per interval, 8 windows per lane and pass (2048 per pass)
//   A  kChainLongWindows + 1 .. kChainBlockWindows      one WAVE per interval, 4 windows per lane and pass
//   B  kChainSerialWindows + 1 .. kChainLongWindows     one 16-lane group per interval, 4 windows per lane and pass
//      (A and B: the first 64 / 16 windows go one per lane -- most binade crossings of a chain fall there)
//   C  at most kChainSerialWindows windows    one LANE per interval
//
// Inside one binade of the accumulator every GC window advances the mantissa by k * delta ulps
// (conga_step_lean, serial_f32.h), an INTEGER.  A+, A and B: each lane sums the advances of its W consecutive
// windows, a DPP prefix sum over the lane group (and LDS across the waves of A+) places them, and a window is
// "regular" when its whole run of k adds stays below the binade top and is not an exact tie.  The first irregular
// window of the group (ballot + ffs, then the lane's first irregular sub-window) is applied with
// conga_window_add_f32 -- which performs the real rounding -- and the pass resumes behind it in the new binade.
// Irregular windows are rare (one per binade crossing, i.e. O(log) per interval), so a 2 Mb interval costs ~10 + 16
// passes instead of 20,000 dependent window updates.  C: the lane applies its windows one after the other (the same
// O(1) fast-forward per window); 64 intervals of similar length share a wave.  Windows are always consumed left to
// right, so the
// rounding sequence is the reference's.
// -------------------------------------------------------------------------------------------
constexpr int kChainBlockWindows = 2048; // class A+ above this
constexpr int kChainLongWindows = 512;
constexpr int kChainSerialWindows = 56;
constexpr int kChainSerialMaxSlots = 32; // class C keeps every chromosome's table in LDS up to this many

struct ChainArgs {
	const int32_t *start;
	const int32_t *end;
	const int32_t *iv_slot;
	const int32_t *order; // interval ids, longest first
	int64_t n_x, n_a, n_b, n_iv; // order[0, n_x): class A+, [n_x, n_x + n_a): class A, then n_b of class B, the rest: class C
	int32_t blocks_ab;            // workgroups shared by classes A and B (behind the n_x of class A+)
	int32_t n_slots;
	const uint8_t *gc_like;
	const Slot *slots;
	Small *small;                      // hist_sum is final (the depth pass is done); E may not be written yet
	const unsigned long long *bases;   // window_per_gc, [n_slots][101]
	int32_t step;
	float *expected; // [n_iv]
	int32_t fused_score;   // 1: score each interval as its chain ends (score.observed etc. are final already)
	ScoreArgs score;
	conga_result *out_host; // pinned host copy of the records (may be null), in INTERVAL order like score.out: a record is one
	                        // 64-byte line wherever it goes (tools/hostwrite.hip: stores into pinned host memory run at the link's
	                        // rate whatever their shape), and the host's fetch is a plain copy -- in processing order, un-permuted by
	                        // the host, 40 000 records cost the fetch 0.25-0.4 ms a step (round 4's last day)
	int32_t table_blocks;   // > 0: that many trailing workgroups do expected_table_kernel's job (one chromosome each)
	Small *host_small;      // its pinned host copy (may be null)
	int32_t zero_blocks;    // > 0: that many workgroups in front of those clear the OTHER accumulator arena (the
	uint4 *zero_ptr;        // per-chromosome blocks and observed[] the next compute will add into), so the next
	int64_t zero_n16;       // step needs no memset launch of its own
};

// expected_read_depth[g] of chromosome sl, computed from the two histograms so that the chain does not have to wait
// for expected_table_kernel (which then only serves the host copy and runs beside the chain)
__device__ __forceinline__ float chain_table_entry(const ChainArgs &a, int sl, int g)
{
	return expected_value(a.small[sl].hist_sum[g], a.bases[(int64_t) sl * kGcBins + g], g);
}

__device__ __forceinline__ void chain_emit(const ChainArgs &a, int32_t iv, int64_t order_pos, float expected)
{
	a.expected[iv] = expected;
	if (a.fused_score) {
		const conga_result r = score_interval(a.score, iv, expected);
		a.score.out[iv] = r;
		if (a.out_host)
			a.out_host[iv] = r;
	}
	(void) order_pos;
}

// Inclusive prefix sum over a lane group through DPP (no LDS crossbar): row_shr 1/2/4/8 inside each
// 16-lane row, then row_bcast:15 / row_bcast:31 to carry row totals across the wave (gfx9 wave64).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
	// lanes without a source (shifted in from outside the row / masked rows) add 0
	return v + (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, CTRL, ROW_MASK, 0xF, false);
}

template <int G> __device__ __forceinline__ uint32_t group_incl_scan_u32(uint32_t v)
{
	static_assert(G == 16 || G == 64, "lane groups are one DPP row or the whole wave");
	v = dpp_add<0x111, 0xF>(v); // row_shr:1
	v = dpp_add<0x112, 0xF>(v); // row_shr:2
	v = dpp_add<0x114, 0xF>(v); // row_shr:4
	v = dpp_add<0x118, 0xF>(v); // row_shr:8
	if (G == 64) {
		v = dpp_add<0x142, 0xA>(v); // row_bcast:15 -> rows 1 and 3
		v = dpp_add<0x143, 0xC>(v); // row_bcast:31 -> rows 2 and 3
	}
	return v;
}

// value of lane `src` (group-relative) for every lane of the group
template <int G> __device__ __forceinline__ uint32_t group_bcast_u32(uint32_t v, int src)
{
	if (G == 64)
		return (uint32_t) __builtin_amdgcn_readlane((int) v, __builtin_amdgcn_readfirstlane(src));
	return (uint32_t) __shfl((int) v, src, G);
}

struct ChainInterval {
	int32_t iv;
	int64_t s0, e0, w_first, w_end, n_win;
	const uint8_t *gc;
	int sl;
};

__device__ __forceinline__ ChainInterval chain_interval(const ChainArgs &a, int64_t at)
{
	ChainInterval c;
	c.iv = a.order[at];
	c.s0 = a.start[c.iv];
	c.e0 = a.end[c.iv];
	c.sl = a.iv_slot[c.iv];
	c.gc = a.gc_like + a.slots[c.sl].gc_off;
	c.n_win = a.slots[c.sl].n_win;
	c.w_first = 0;
	c.w_end = 0;
	if (c.e0 > c.s0) {
		c.w_first = (uint32_t) c.s0 / (uint32_t) a.step;
		c.w_end = (uint32_t) (c.e0 - 1) / (uint32_t) a.step + 1;
	}
	return c;
}

// The passes of one step of classes A and B: lane gl of a group holds W consecutive windows (k adds of the float whose
// bits are bc, ca = conga_addend_of(bc); k = 0: no window) and every lane holds the group's accumulator `s`.  Returns
// the accumulator behind the step's last window.
template <int G, int W> __device__ __forceinline__ float chain_step_passes(float s, const uint32_t (&k)[W], const uint32_t (&bc)[W],
		const conga_addend (&ca)[W], bool any_act, int gl, int grp, unsigned long long gmask)
{
	constexpr int SW = G * W;
	int next = 0; // first position of this step (gl * W + j) that is not applied yet; uniform inside a group
	bool pending = (__ballot(any_act) & gmask) != 0ull;
	while (__any(pending)) {
		const uint32_t bs = conga_f32_bits(s);
		const uint32_t es = bs >> 23;
		const uint32_t ms = (bs & 0x7FFFFFu) | 0x800000u;
		uint32_t adv[W], lim[W], dl[W];
		bool in[W], valid[W];
		uint32_t lane_total = 0;
#pragma unroll
		for (int j = 0; j < W; j++) {
			in[j] = pending && k[j] != 0 && (gl * W + j) >= next;
			// ok implies delta <= 2^21, which keeps k * delta (k <= gc_step <= 1024) inside 32 bits and lets the
			// 24-bit multiplier do it at full rate; larger steps (accumulator within 8x of the addend: the first
			// window or two of an interval) go the irregular way, as do ties and a negative accumulator (es >= 256)
			const conga_lean_step st = conga_step_lean(es, ca[j]);
			valid[j] = in[j] && st.ok != 0u && st.tie == 0u;
			dl[j] = st.delta;
			lim[j] = st.lim;
			const uint32_t a32 = __umul24(k[j], st.delta & 0x3FFFFFu); // (delta <= 2^21 whenever it is used)
			adv[j] = !valid[j] ? 0u : (a32 > (1u << 24)) ? (1u << 24) : a32; // beyond the binade top anyway
			lane_total += adv[j];
		}
		if (lane_total > (1u << 25))
			lane_total = 1u << 25; // keeps the group sum below 2^32; only ever hit behind an irregular window
		const uint32_t incl = group_incl_scan_u32<G>(lane_total);
		uint32_t m = ms + (incl - lane_total); // mantissa in front of this lane's first window
		int jb = W;         // first irregular window of this lane
		uint32_t m_bad = 0; // mantissa in front of it
#pragma unroll
		for (int j = 0; j < W; j++) {
			// regular: the whole run of k adds starts at or below lim.  (A stuck window, delta 0 / lim 2^24 - 1, behind
			// a prefix that landed exactly on the binade top is sent the irregular way too: the top is the next binade.)
			const bool ok = valid[j] & (m <= lim[j]) & (__umul24((k[j] - 1u) & 0x7FFu, dl[j] & 0x3FFFFFu) <= lim[j] - m);
			const bool first_bad = in[j] & !ok & (jb == W);
			jb = first_bad ? j : jb;
			m_bad = first_bad ? m : m_bad;
			m += adv[j];
		}
		const unsigned long long bad_all = __ballot(jb < W);
		const unsigned long long bad = bad_all & gmask;
		const int fb = bad ? (__ffsll((long long) bad) - 1 - grp * G) : 0;
		const uint32_t total = group_bcast_u32<G>(incl, G - 1);
		uint32_t m_fb = 0, k_fb = 0, bc_fb = 0;
		int j_fb = 0;
		if (bad_all) { // some group of this wave has an irregular window
			uint32_t k_sel = 0, bc_sel = 0;
#pragma unroll
			for (int j = 0; j < W; j++)
				if (j == jb) {
					k_sel = k[j];
					bc_sel = bc[j];
				}
			m_fb = group_bcast_u32<G>(m_bad, fb);
			j_fb = (int) group_bcast_u32<G>((uint32_t) jb, fb);
			k_fb = group_bcast_u32<G>(k_sel, fb);
			bc_fb = group_bcast_u32<G>(bc_sel, fb);
		}
		if (pending) {
			if (bad == 0ull) {
				if (total)
					s = conga_compose_f32(es, ms + total);
				pending = false;
			} else {
				if (m_fb != ms)
					s = conga_compose_f32(es, m_fb); // exact state in front of the irregular window
				s = conga_window_add_f32(s, conga_bits_f32(bc_fb), k_fb); // real adds where rounding is not a constant step
				next = fb * W + j_fb + 1;
				if (next >= SW)
					pending = false;
			}
		}
	}
	return s;
}

// Classes A (G = 64) and B (G = 16), W = 4 windows per lane and pass.
// `wave`: index of this wave inside its class; `sE_wave`: this wave's LDS (one depth table per lane group).
template <int G, int W> __device__ __forceinline__ void chain_group_body(const ChainArgs &a, int64_t wave, int64_t first,
		int64_t count, float *sE_wave)
{
	constexpr int kGroups = kWave / G;
	constexpr int SW = G * W; // windows per step
	static_assert(W == 4 || W == 8, "a lane's windows are one aligned 4- or 8-byte word of GC bytes");

	const int lane = threadIdx.x & (kWave - 1);
	const int gl = lane & (G - 1);  // lane inside the group
	const int grp = lane / G;       // group inside the wave
	const unsigned long long gmask = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (grp * G));
	const int64_t slot_idx = wave * kGroups + grp;
	const bool have = slot_idx < count;
	float *E = sE_wave + grp * (kGcBins + 3);

	ChainInterval ci = {0, 0, 0, 0, 0, 1, a.gc_like, 0};
	if (have) {
		ci = chain_interval(a, first + slot_idx);
		for (int g = gl; g < kGcBins; g += G)
			E[g] = chain_table_entry(a, ci.sl, g);
	}
	const int64_t s0 = ci.s0, e0 = ci.e0, w_first = ci.w_first, w_end = ci.w_end, n_win = ci.n_win;
	const uint8_t *gc = ci.gc;
	const int64_t step = a.step;
	const int64_t n_win_pad = (n_win + 15) & ~(int64_t) 15; // the slot's GC region is padded to 16 bytes
	const uint32_t gc_last = have ? gc[n_win - 1] : 0;      // windows past the chromosome end use the last one
	// this lane's W GC bytes of the step that starts at window `step_base` (zero past the padded region)
	auto fetch = [&](int64_t step_base) -> uint64_t {
		const int64_t at = step_base + (int64_t) gl * W;
		if (!have || at >= n_win_pad)
			return 0;
		if (W == 8)
			return *reinterpret_cast<const uint64_t *>(gc + at);
		return *reinterpret_cast<const uint32_t *>(gc + at);
	};
	// k and the addend of window w (k = 0: not one of this interval's windows at or behind w_from)
	auto window = [&](uint32_t w, uint32_t g_raw, uint32_t w_from, uint32_t &k_out, uint32_t &bc_out) -> bool {
		// select-style on purpose: executed by every lane, and a branch around a few instructions costs more (mask
		// bookkeeping on the scalar unit, a refilled instruction buffer) than they do
		const uint32_t edge = w * (uint32_t) step; // first base of window w; positions stay below 2^31 + step
		const bool act = have && w >= w_from && w < (uint32_t) w_end;
		const uint32_t lo = (edge > (uint32_t) s0) ? edge : (uint32_t) s0;
		const uint32_t hi = (edge + (uint32_t) step < (uint32_t) e0) ? edge + (uint32_t) step : (uint32_t) e0;
		k_out = act ? hi - lo : 0u; // >= 1 when active
		const uint32_t g_cur = (w < (uint32_t) n_win) ? g_raw : gc_last;
		const float e_cur = E[(g_cur < (uint32_t) kGcBins) ? g_cur : 0u];
		bc_out = (act && g_cur < (uint32_t) kGcBins) ? conga_f32_bits(e_cur) : 0u;
		return act;
	};

	float s = 0.0f; // uniform inside a group
	// Head: the first G windows, one per lane.  The accumulator doubles within the first window, again by the
	// second, the fourth, the eighth ...: most binade crossings of a chain -- each costs a pass -- fall into its first
	// dozen windows, and a pass over one window per lane is a third of the instructions of a pass over four.
	const uint32_t head_w = (uint32_t) w_first + (uint32_t) gl;
	const uint32_t head_g = (have && head_w < (uint32_t) n_win) ? gc[head_w] : 0u;
	int64_t wb = (w_first + G) & ~(int64_t) (SW - 1); // steps are aligned, lanes in front of w_first + G idle
	if (w_first + G >= w_end)
		wb = (w_end + SW - 1) & ~(int64_t) (SW - 1); // (nothing behind the head)
	uint64_t cur = fetch(wb), nxt1 = fetch(wb + SW);
	__builtin_amdgcn_wave_barrier(); // the table is written and read by lanes of the same wave: LDS ops stay in order
	{
		uint32_t k1[1], bc1[1];
		conga_addend ca1[1];
		const bool act = window(head_w, head_g, (uint32_t) w_first, k1[0], bc1[0]);
		ca1[0] = conga_addend_of(bc1[0]);
		s = chain_step_passes<G, 1>(s, k1, bc1, ca1, act, gl, grp, gmask);
	}
	const uint32_t w_from = (uint32_t) w_first + (uint32_t) G;
	while (__any(wb < w_end)) {
		const uint64_t nxt2 = fetch(wb + 2 * SW); // two steps ahead, so that no step waits on HBM
		uint32_t k[W], bc[W];
		conga_addend ca[W];
		bool any_act = false;
		const uint32_t w0 = (uint32_t) wb + (uint32_t) gl * W;
#pragma unroll
		for (int j = 0; j < W; j++) {
			any_act |= window(w0 + j, (uint32_t) ((cur >> (8 * j)) & 0xFFu), w_from, k[j], bc[j]);
			ca[j] = conga_addend_of(bc[j]);
		}
		s = chain_step_passes<G, W>(s, k, bc, ca, any_act, gl, grp, gmask);
		wb += SW;
		cur = nxt1;
		nxt1 = nxt2;
	}
	if (have && gl == 0)
		chain_emit(a, ci.iv, first + slot_idx, s);
}

// Class A+: one WORKGROUP per interval (the handful of chains with thousands of windows that would otherwise be the
// tail of the launch): the same pass as chain_group_body<64, 8>, 2048 windows wide.  The four waves exchange their
// totals and their first irregular window through LDS (two barriers per pass, slots alternate between passes).
__device__ __forceinline__ void chain_block_body(const ChainArgs &a, int64_t block, int64_t first, int64_t count,
		float *E, uint32_t *xw)
{
	constexpr int W = 8;
	constexpr int SW = 256 * W; // windows per step
	const int lane = threadIdx.x & (kWave - 1);
	const int wid = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / kWave)); // (uniform in the wave; said so, it and what follows from it are scalars)
	const int gl = threadIdx.x; // lane inside the group = the workgroup
	const bool have = block < count;
	ChainInterval ci = {0, 0, 0, 0, 0, 1, a.gc_like, 0};
	if (have) {
		ci = chain_interval(a, first + block);
		if (gl < kGcBins)
			E[gl] = chain_table_entry(a, ci.sl, gl);
	}
	__syncthreads();
	const int64_t s0 = ci.s0, e0 = ci.e0, w_first = ci.w_first, w_end = ci.w_end, n_win = ci.n_win;
	const uint8_t *gc = ci.gc;
	const int64_t step = a.step;
	const int64_t n_win_pad = (n_win + 15) & ~(int64_t) 15;
	const uint32_t gc_last = have ? gc[n_win - 1] : 0;
	auto fetch = [&](int64_t step_base) -> uint64_t {
		const int64_t at = step_base + (int64_t) gl * W;
		return (have && at < n_win_pad) ? *reinterpret_cast<const uint64_t *>(gc + at) : 0ull;
	};

	float s = 0.0f; // uniform in the workgroup
	int64_t wb = w_first & ~(int64_t) (SW - 1);
	uint64_t cur = fetch(wb), nxt1 = fetch(wb + SW);
	uint32_t pass = 0;
	while (wb < w_end) { // workgroup-uniform
		const uint64_t nxt2 = fetch(wb + 2 * SW);
		uint32_t k[W], bc[W];
		conga_addend ca[W];
		bool any_act = false;
		{
			const uint32_t w0 = (uint32_t) wb + (uint32_t) gl * W;
			uint32_t edge = w0 * (uint32_t) step;
#pragma unroll
			for (int j = 0; j < W; j++) {
				const uint32_t w = w0 + j;
				const bool act = w >= (uint32_t) w_first && w < (uint32_t) w_end;
				const uint32_t lo = (edge > (uint32_t) s0) ? edge : (uint32_t) s0;
				const uint32_t hi = (edge + (uint32_t) step < (uint32_t) e0) ? edge + (uint32_t) step : (uint32_t) e0;
				k[j] = act ? hi - lo : 0u;
				const uint32_t g_cur = (w < (uint32_t) n_win) ? (uint32_t) ((cur >> (8 * j)) & 0xFFu) : gc_last;
				const float e_cur = E[(g_cur < (uint32_t) kGcBins) ? g_cur : 0u];
				bc[j] = (act && g_cur < (uint32_t) kGcBins) ? conga_f32_bits(e_cur) : 0u;
				any_act |= act;
				ca[j] = conga_addend_of(bc[j]);
				edge += (uint32_t) step;
			}
		}
		(void) any_act;
		int next = 0;        // first position of this step (gl * W + j) that is not applied yet
		bool pending = true; // the step holds at least one window of the interval (wb < w_end, steps are aligned)
		while (pending) {    // workgroup-uniform
			uint32_t *slot = xw + (pass & 1u) * 32; // [0..3] wave totals, [4..7] first irregular position, [8..19] its m / k / bc
			pass++;
			const uint32_t bs = conga_f32_bits(s);
			const uint32_t es = bs >> 23;
			const uint32_t ms = (bs & 0x7FFFFFu) | 0x800000u;
			uint32_t adv[W], lim[W], dl[W];
			bool in[W], valid[W];
			uint32_t lane_total = 0;
#pragma unroll
			for (int j = 0; j < W; j++) {
				in[j] = k[j] != 0 && (gl * W + j) >= next;
				const conga_lean_step st = conga_step_lean(es, ca[j]);
				valid[j] = in[j] && st.ok != 0u && st.tie == 0u;
				dl[j] = st.delta;
				lim[j] = st.lim;
				const uint32_t a32 = __umul24(k[j], st.delta & 0x3FFFFFu);
				adv[j] = !valid[j] ? 0u : (a32 > (1u << 24)) ? (1u << 24) : a32;
				lane_total += adv[j];
			}
			// A regular prefix never exceeds 2^24, so every clamp below (lane 2^25, wave 2^31, workgroup 2^30) is only
			// ever hit behind an irregular window, where the value is not used.
			if (lane_total > (1u << 25))
				lane_total = 1u << 25;
			const uint32_t incl_w = group_incl_scan_u32<64>(lane_total);
			if (lane == kWave - 1)
				slot[wid] = incl_w;
			__syncthreads();
			uint32_t before = 0, total = 0;
#pragma unroll
			for (int v = 0; v < 4; v++) {
				const uint32_t t = min(slot[v], 1u << 28);
				before += (v < wid) ? t : 0u;
				total += t;
			}
			uint32_t m = ms + before + (incl_w - lane_total); // mantissa in front of this lane's first window
			int jb = W;
			uint32_t m_bad = 0;
#pragma unroll
			for (int j = 0; j < W; j++) {
				// regular: the whole run of k adds starts at or below lim.  (A stuck window, delta 0 / lim 2^24 - 1, behind
				// a prefix that landed exactly on the binade top is sent the irregular way too: the top is the next binade.)
				const bool ok = valid[j] & (m <= lim[j]) & (__umul24((k[j] - 1u) & 0x7FFu, dl[j] & 0x3FFFFFu) <= lim[j] - m);
				const bool first_bad = in[j] & !ok & (jb == W);
				jb = first_bad ? j : jb;
				m_bad = first_bad ? m : m_bad;
				m += adv[j];
			}
			const unsigned long long bad = __ballot(jb < W);
			if (bad) { // wave-uniform: this wave's first irregular window goes to LDS
				const int fb = __ffsll((long long) bad) - 1;
				uint32_t k_sel = 0, bc_sel = 0;
#pragma unroll
				for (int j = 0; j < W; j++)
					if (j == jb) {
						k_sel = k[j];
						bc_sel = bc[j];
					}
				if (lane == fb) {
					slot[4 + wid] = (uint32_t) (gl * W + jb);
					slot[8 + wid] = m_bad;
					slot[12 + wid] = k_sel;
					slot[16 + wid] = bc_sel;
				}
			} else if (lane == 0)
				slot[4 + wid] = 0xFFFFFFFFu;
			__syncthreads();
			uint32_t first_pos = 0xFFFFFFFFu, m_fb = 0, k_fb = 0, bc_fb = 0;
#pragma unroll
			for (int v = 0; v < 4; v++) { // waves hold ascending positions: the first wave with an entry wins
				const uint32_t p = slot[4 + v];
				if (first_pos == 0xFFFFFFFFu && p != 0xFFFFFFFFu) {
					first_pos = p;
					m_fb = slot[8 + v];
					k_fb = slot[12 + v];
					bc_fb = slot[16 + v];
				}
			}
			if (first_pos == 0xFFFFFFFFu) {
				if (total)
					s = conga_compose_f32(es, ms + total);
				pending = false;
			} else {
				if (m_fb != ms)
					s = conga_compose_f32(es, m_fb); // exact state in front of the irregular window
				s = conga_window_add_f32(s, conga_bits_f32(bc_fb), k_fb);
				next = (int) first_pos + 1;
				if (next >= SW)
					pending = false;
			}
		}
		wb += SW;
		cur = nxt1;
		nxt1 = nxt2;
	}
	if (have && gl == 0)
		chain_emit(a, ci.iv, first + block, s);
}

// Class C: one lane per interval.  The addend of window w + 1 (GC byte -> table lookup) is fetched while window w
// is applied, so the only latency a trip waits for is its own dozen dependent ALU operations.
template <bool LDS_TABLES> __device__ __forceinline__ void chain_serial_lanes(const ChainArgs &a, int64_t idx, int64_t first,
		int64_t count, const float *sE_all, uint4 *stage)
{
	const bool have = idx < count;
	ChainInterval ci = {0, 0, 0, 0, 0, 1, a.gc_like, 0};
	if (have)
		ci = chain_interval(a, first + idx);
	const float *E = sE_all + ci.sl * (kGcBins + 3); // LDS_TABLES only
	const int64_t s0 = ci.s0, e0 = ci.e0, n_win = ci.n_win, w_end = ci.w_end;
	const uint8_t *gc = ci.gc;
	const int64_t step = a.step;
	const int64_t n_win_pad = (n_win + 15) & ~(int64_t) 15;
	const uint32_t gc_last = have ? gc[n_win - 1] : 0;
	// GC bytes: 64 windows at a time (16 aligned dwords per lane) sit in the wave's slice of `stage`, one LDS column per
	// lane (dword k of lane l at [k * 64 + l]: conflict-free), and the next 64 are already in registers.  No global
	// load inside the trip loop: one that the loop top has to wait for (every fourth trip, with the words fetched one
	// at a time) cost more than the arithmetic of the trip.
	uint32_t *col = reinterpret_cast<uint32_t *>(stage) + __builtin_amdgcn_readfirstlane((int) (threadIdx.x / kWave)) * (kWave * 16)
			+ (threadIdx.x & (kWave - 1));
	// window indices and positions stay below 2^31 + step: 32-bit arithmetic throughout the trip loop
	const uint32_t w_end32 = (uint32_t) w_end, n_win32 = (uint32_t) n_win;
	uint32_t w = (uint32_t) ci.w_first;
	uint32_t chunk_w = w & ~3u; // first window of the chunk in the column
	// wave-uniform: no lane of this wave ever moves on to a second column (always so at the default class boundary)
	const bool one_chunk = __all(!have || w_end32 - chunk_w <= 64u) != 0;
	uint32_t ahead[16];
	auto load16 = [&](uint32_t from) {
#pragma unroll
		for (int q = 0; q < 16; q++) {
			const int64_t at = (int64_t) from + 4 * q;
			ahead[q] = (have && at < n_win_pad) ? *reinterpret_cast<const uint32_t *>(gc + at) : 0u;
		}
	};
	auto to_column = [&]() {
#pragma unroll
		for (int q = 0; q < 16; q++)
			col[q * kWave] = ahead[q];
	};
	load16(chunk_w);
	to_column();
	if (!one_chunk && w_end32 > chunk_w + 64u)
		load16(chunk_w + 64u);
	// k and the addend's bits for the next window
	uint32_t at = (uint32_t) s0;                           // first base not yet accounted for
	uint32_t edge = (w + 1u) * (uint32_t) step;            // first base of the window after `w`
	auto window = [&](uint32_t &k, uint32_t &bc) {
		if (!one_chunk && w - chunk_w >= 64u && w < w_end32) { // this lane moves on to its next 64 windows
			to_column();
			chunk_w += 64u;
			if (w_end32 > chunk_w + 64u)
				load16(chunk_w + 64u);
		}
		// select-style: every lane runs this once per trip, and a branch around a few instructions costs more than they do
		const bool act = w < w_end32;
		const uint32_t hi = (edge < (uint32_t) e0) ? edge : (uint32_t) e0;
		k = act ? hi - at : 0u;
		at = act ? hi : at;
		const uint32_t rel = act ? w - chunk_w : 0u; // (below 64 when act)
		const uint32_t word = col[(rel >> 2) * kWave];
		const uint32_t g_cur = (w < n_win32) ? ((word >> (8 * (rel & 3u))) & 0xFFu) : gc_last;
		const bool g_ok = g_cur < (uint32_t) kGcBins;
		float c = 0.0f;
		if (LDS_TABLES)
			c = E[g_ok ? g_cur : 0u];
		else if (act && g_ok)
			c = chain_table_entry(a, ci.sl, (int) g_cur);
		bc = (act && g_ok) ? conga_f32_bits(c) : 0u;
		edge += (uint32_t) step;
		w++;
	};
	float s = 0.0f;
	uint32_t k, bc;
	window(k, bc);
	while (__any(k != 0)) {
		uint32_t k_next, bc_next;
		window(k_next, bc_next); // the next addend is on its way while this window is applied
		// Inside one binade k adds are one integer step, and that is what most trips are for every lane of the wave: the
		// one-crossing candidate (a division, a second binade) is only worked out in the trips where some lane needs it.
		const conga_window_head h = conga_window_stage1(s, conga_bits_f32(bc), k);
		if (__all(k == 0u || h.fit)) {
			s = k ? h.res_fit : s;
		} else if (k) {
			s = conga_window_finish(h, s, conga_bits_f32(bc), k);
		}
		k = k_next;
		bc = bc_next;
	}
	if (!a.fused_score || !a.out_host) {
		if (have)
			chain_emit(a, ci.iv, first + idx, s);
		return;
	}
	// Scored here, and the pinned host copy written a whole 64-byte record at a time: every lane parks its record in LDS,
	// then four neighbouring lanes write one record's four 16-byte quarters, sixteen records per instruction, each to its
	// interval's place.
	const int lane = threadIdx.x & (kWave - 1);
	uint4 *my_stage = stage + (size_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x / kWave)) * kWave * 4; // this wave's 64 records
	if (have) {
		a.expected[ci.iv] = s;
		const conga_result r = score_interval(a.score, ci.iv, s);
		a.score.out[ci.iv] = r;
		*reinterpret_cast<conga_result *>(my_stage + lane * 4) = r;
	}

	__builtin_amdgcn_wave_barrier(); // written and read by lanes of the same wave: LDS ops stay in order
	const int32_t my_iv = have ? ci.iv : -1;
#pragma unroll
	for (int t = 0; t < 4; t++) {
		const int src = t * 16 + (lane >> 2);
		const int32_t iv_src = __shfl(my_iv, src, kWave); // the interval whose record lane `src` parked
		const uint4 v = my_stage[src * 4 + (lane & 3)];
		if (iv_src >= 0)
			reinterpret_cast<uint4 *>(a.out_host + iv_src)[lane & 3] = v;
	}
}

__device__ __forceinline__ void chain_serial_body(const ChainArgs &a, int64_t block, int64_t first, int64_t count,
		float *sE_all, uint4 *stage)
{
	const int64_t idx = block * blockDim.x + threadIdx.x;
	if (a.n_slots <= kChainSerialMaxSlots) { // workgroup-uniform: every chromosome's table fits in LDS
		for (int i = threadIdx.x; i < a.n_slots * kGcBins; i += blockDim.x)
			sE_all[(i / kGcBins) * (kGcBins + 3) + i % kGcBins] = chain_table_entry(a, i / kGcBins, i % kGcBins);
		__syncthreads();
		chain_serial_lanes<true>(a, idx, first, count, sE_all, stage);
	} else
		chain_serial_lanes<false>(a, idx, first, count, sE_all, stage);
}

constexpr int kChainLdsWords = kChainSerialMaxSlots * (kGcBins + 3); // >= 16 group tables of class B

// Register budget for 5 waves per SIMD: the launch (~1000 workgroups for a 1000G-sized call set) must stay inside one
// resident wave of workgroups.  At 4 per SIMD the limit is 1024 workgroups, and a call set that needs 1040 runs 25 us
// longer (measured by moving the class-C threshold from 56 to 52 windows).
__global__ __launch_bounds__(256, 5) void interval_chain_kernel(ChainArgs a)
{
	__shared__ float sE[kChainLdsWords];
	__shared__ uint4 stage[256 * 4]; // class C: one 64-byte record per lane on its way to the host
	__shared__ uint32_t xw[64];      // class A+: what the four waves of a pass tell each other
	int b = (int) blockIdx.x;
	if (b < (int) a.n_x) {
		__builtin_amdgcn_s_setprio(3); // a handful of workgroups, and the longest critical path of the launch
		chain_block_body(a, (int64_t) b, 0, a.n_x, sE, xw);
		return;
	}
	b -= (int) a.n_x;
	// The classes share SIMDs.  Giving the many short chains issue priority over the few long ones was measured
	// 13 us faster than equal priorities (and than favouring the long ones): their waves retire early and leave the
	// SIMDs to the long chains.
	const int grid = (int) gridDim.x - (int) a.n_x;
	if (b >= grid - a.table_blocks - a.zero_blocks && b < grid - a.table_blocks) {
		const int zb = b - (grid - a.table_blocks - a.zero_blocks);
		for (int64_t i = (int64_t) zb * blockDim.x + threadIdx.x; i < a.zero_n16; i += (int64_t) a.zero_blocks * blockDim.x)
			a.zero_ptr[i] = make_uint4(0, 0, 0, 0);
		return;
	}
	if (b < a.blocks_ab) {
		// Classes A and B share workgroups: one wave of each of the first n_a workgroups takes a long chain (class A,
		// the slot rotates so that they land on different SIMDs), every other wave takes four class-B chains.  Packing
		// four long chains into one workgroup put them on one CU, whose SIMDs then ran 1.5x the work of the others.
		const int wid = __builtin_amdgcn_readfirstlane((int) (threadIdx.x / kWave)); // (the wave's number: a scalar, and so are the chains it picks)
		float *sE_wave = sE + wid * 4 * (kGcBins + 3);
		const int a_slot = b & 3;
		if (b < (int) a.n_a && wid == a_slot)
			chain_group_body<64, 4>(a, (int64_t) b, a.n_x, a.n_a, sE_wave);
		else {
			const int64_t bw = (b < (int) a.n_a) ? (int64_t) b * 3 + (wid - (wid > a_slot ? 1 : 0))
					: a.n_a * 3 + ((int64_t) b - a.n_a) * 4 + wid;
			__builtin_amdgcn_s_setprio(2);
			chain_group_body<16, 4>(a, bw, a.n_x + a.n_a, a.n_b, sE_wave);
		}
	} else if (b >= grid - a.table_blocks) {
		__builtin_amdgcn_s_setprio(3);
		expected_table_body(a.small, a.bases, a.host_small, b - (grid - a.table_blocks));
	} else {
		__builtin_amdgcn_s_setprio(3);
		chain_serial_body(a, (int64_t) (b - a.blocks_ab), a.n_x + a.n_a + a.n_b, a.n_iv - a.n_x - a.n_a - a.n_b, sE, stage);
	}

}

// -------------------------------------------------------------------------------------------
// K5 interval_score: score_interval for every interval, one lane each (the path that waits for interval_reduce).
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interval_score_kernel(ScoreArgs a)
{
	const int64_t iv = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (iv >= a.n_iv)
		return;
	a.out[iv] = score_interval(a, iv, a.expected[iv]);
}

// (the split-read evidence path -- SURVEY.md section 8 rows a15-a18 -- is split_map.hip.h)

} // namespace conga
