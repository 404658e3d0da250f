// kernels.hip.h -- gfx950 (CDNA4, wave64) kernels of the read-depth / likelihood path.
//
// All kernels are HBM-bound integer/byte work or short scalar chains; none is a contraction, so no
// MFMA is used.  Layout in HBM per chromosome:
//   pos   int32[N], mapq uint8[N]   read tuples in BAM order (sorted by pos)
//   rd    int16[L]                  bam_info.read_depth          (common.h:91)
//   map   float[L]                  bam_info.mappability         (common.h:92)
//   gc_*  uint8[n_win]              rounded GC% per `step`-base window
//   E     float[101]                bam_info.expected_read_depth (common.h:94)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/conga_hip.h"
#include "serial_f32.h"

namespace conga {

constexpr int kWave = 64;
constexpr int kGcBins = 101; // read_distribution.c:51-52

// device status word bits
constexpr uint32_t kStatusUnsorted = 1u;

// device counters (uint64 each)
enum { CNT_COUNTED = 0, CNT_OUT_OF_RANGE, CNT_N };

// -------------------------------------------------------------------------------------------
// wave64 reductions (DPP-backed shuffles)
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

// -------------------------------------------------------------------------------------------
// K0 ingest: one pass over the read tuples of a chromosome.
//   * flags tuples that break the position order (the tile index below needs sorted input,
//     which is what sam_itr_next over an indexed BAM yields: bam_data.c:201,293);
//   * counts tuples outside [0, L) (the reference would write out of bounds: bam_data.c:213);
//   * builds tile_start[t] = index of the first tuple whose position falls in depth tile t or
//     later, so the depth kernel needs no search.  Entries past the last tuple's tile keep the
//     memset value 0xFFFFFFFF (= "n").
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ingest_kernel(const int32_t *__restrict__ pos, int64_t n, int64_t L,
		int32_t tile_len, int64_t n_tiles, uint32_t *__restrict__ tile_start, uint32_t *__restrict__ status,
		unsigned long long *__restrict__ counters)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	unsigned long long oor = 0;
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const int32_t p = pos[i];
		const int32_t prev = (i > 0) ? pos[i - 1] : INT32_MIN;
		if (p < prev)
			atomicOr(status, kStatusUnsorted);
		int64_t t_cur, t_prev;
		if (p < 0) {
			t_cur = 0;
			oor++;
		} else if (p >= L) {
			t_cur = n_tiles;
			oor++;
		} else
			t_cur = p / tile_len;
		if (i == 0)
			t_prev = -1;
		else if (prev < 0)
			t_prev = 0;
		else if (prev >= L)
			t_prev = n_tiles;
		else
			t_prev = prev / tile_len;
		for (int64_t t = t_prev + 1; t <= t_cur; t++)
			tile_start[t] = (uint32_t) i;
	}
	// one atomic per wave
	int w = wave_sum_i32((int) oor);
	if ((threadIdx.x & (kWave - 1)) == 0 && w)
		atomicAdd(&counters[CNT_OUT_OF_RANGE], (unsigned long long) w);
}

// -------------------------------------------------------------------------------------------
// K1 + K2 depth_tile: builds read_depth and the GC-stratified sums in one pass.
//
// Replaces init_rd_per_chr's memset (read_distribution.c:16-17), the increments of
// count_reads_bam (bam_data.c:205-215) and both loops of calc_mean_per_chr
// (read_distribution.c:33-37,63-73).  Each workgroup owns tiles of tile_win * step positions:
// it zeroes 32-bit counters in LDS, adds the tile's tuples with LDS atomics, then streams the
// tile out ONCE as int16 with 16-byte stores (so read_depth is written exactly once and never
// read back for the histogram), accumulating per-window sums -> a 101-bin {sum, bases} histogram
// kept in LDS across all of the workgroup's tiles and flushed with one global atomic per bin.
// All accumulators are integers, so the result does not depend on the order of the atomics.
// -------------------------------------------------------------------------------------------
constexpr int kDepthBlock = 256;
constexpr int kDepthMaxTile = 8192; // positions per tile (32 KiB of LDS counters)

struct DepthArgs {
	const int32_t *pos;
	const uint8_t *mapq;
	int64_t n;
	const uint32_t *tile_start;
	int16_t *rd;
	int64_t L;
	const uint8_t *gc_hist;
	int64_t n_win;
	int32_t step;
	int32_t tile_win;
	int32_t mq_threshold;
	int64_t n_tiles;
	unsigned long long *hist_sum;   // [101]
	unsigned long long *hist_bases; // [101]
	unsigned long long *counters;
	const uint32_t *status;
};

__global__ __launch_bounds__(kDepthBlock) void depth_tile_kernel(DepthArgs a)
{
	__shared__ int32_t cnt[kDepthMaxTile];
	__shared__ int32_t wsum[kDepthMaxTile / 8 + 8];
	__shared__ unsigned long long h_sum[kGcBins];
	__shared__ unsigned int h_bases[kGcBins];

	if (*a.status & kStatusUnsorted)
		return; // tile index is meaningless; the host reports CONGA_ERR_UNSORTED

	const int tid = threadIdx.x;
	const int T = a.tile_win * a.step;
	for (int g = tid; g < kGcBins; g += kDepthBlock) {
		h_sum[g] = 0;
		h_bases[g] = 0;
	}
	unsigned int counted = 0;

	for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
		const int64_t base = tile * T;
		const int len = (int) ((a.L - base < T) ? (a.L - base) : T);

		for (int j = tid; j < T; j += kDepthBlock)
			cnt[j] = 0;
		for (int j = tid; j < a.tile_win; j += kDepthBlock)
			wsum[j] = 0;
		__syncthreads();

		const uint32_t lo = a.tile_start[tile];
		uint32_t hi = a.tile_start[tile + 1];
		if (lo != 0xFFFFFFFFu) {
			if (hi == 0xFFFFFFFFu)
				hi = (uint32_t) a.n;
			for (uint32_t i = lo + tid; i < hi; i += kDepthBlock) {
				const int64_t p = (int64_t) a.pos[i] - base;
				if (p >= 0 && p < len && (int) a.mapq[i] > a.mq_threshold) {
					atomicAdd(&cnt[p], 1);
					counted++;
				}
			}
		}
		__syncthreads();

		// stream the tile out: 8 positions (16 bytes) per lane per step
		for (int j = tid * 8; j < len; j += kDepthBlock * 8) {
			int16_t v[8];
			int w = j / a.step;
			int r = j - w * a.step;
			int acc = 0;
#pragma unroll
			for (int e = 0; e < 8; e++) {
				// `short` wrap of read_depth[pos]++ (two's complement, as gcc does)
				const int16_t s16 = (j + e < len) ? (int16_t) cnt[j + e] : (int16_t) 0;
				v[e] = s16;
				acc += (int) s16;
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
			if (j + 8 <= len) {
				*reinterpret_cast<uint4 *>(a.rd + base + j) = *reinterpret_cast<const uint4 *>(v);
			} else {
				for (int e = 0; j + e < len; e++)
					a.rd[base + j + e] = v[e];
			}
		}
		__syncthreads();

		// per-window sums -> GC bins (read_distribution.c:70-72)
		const int nw = (len + a.step - 1) / a.step;
		const int64_t w0 = tile * a.tile_win;
		for (int w = tid; w < nw; w += kDepthBlock) {
			int64_t wg = w0 + w;
			if (wg >= a.n_win)
				wg = a.n_win - 1;
			const int g = a.gc_hist[wg];
			const int nb = (len - w * a.step < a.step) ? (len - w * a.step) : a.step;
			if (g < kGcBins) {
				const int s = wsum[w];
				if (s)
					atomicAdd(&h_sum[g], (unsigned long long) (long long) s);
				atomicAdd(&h_bases[g], (unsigned int) nb);
			}
		}
		__syncthreads();
	}

	for (int g = tid; g < kGcBins; g += kDepthBlock) {
		if (h_sum[g])
			atomicAdd(&a.hist_sum[g], h_sum[g]);
		if (h_bases[g])
			atomicAdd(&a.hist_bases[g], (unsigned long long) h_bases[g]);
	}
	int w = wave_sum_i32((int) counted);
	if ((tid & (kWave - 1)) == 0 && w)
		atomicAdd(&a.counters[CNT_COUNTED], (unsigned long long) w);
}

// Depth for unsorted input (CONGA_FLAG_READS_UNSORTED): read_depth is zeroed by hipMemsetAsync and
// incremented with global atomics on the containing 32-bit word.
__global__ __launch_bounds__(256) void depth_atomic_kernel(const int32_t *__restrict__ pos,
		const uint8_t *__restrict__ mapq, int64_t n, int64_t L, int32_t mq_threshold, int16_t *rd,
		unsigned long long *counters)
{
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	unsigned int counted = 0;
	for (int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const int64_t p = pos[i];
		if (p < 0 || p >= L || (int) mapq[i] <= mq_threshold)
			continue;
		unsigned int *word = reinterpret_cast<unsigned int *>(rd) + (p >> 1);
		// carry out of the low half would corrupt the high half: split the add when it wraps
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
// K2b expected_table: expected_read_depth[g] = (float) rd_per_gc[g] / window_per_gc[g]
// (read_distribution.c:75-83): float(long) / float(int) in single precision, [0] forced to 0,
// NaN / +-inf -> 0.  Integer -> float goes through double (exact below 2^53, then one rounding),
// which equals the correctly rounded direct conversion.
// -------------------------------------------------------------------------------------------
__global__ void expected_table_kernel(const unsigned long long *__restrict__ hist_sum,
		const unsigned long long *__restrict__ hist_bases, float *__restrict__ E)
{
	const int g = threadIdx.x;
	if (g >= kGcBins)
		return;
	float e = 0.0f;
	if (g > 0) {
		const float num = (float) (double) (long long) hist_sum[g];
		const float den = (float) (double) (int) hist_bases[g];
		e = num / den;
		if (isnan(e) || isinf(e))
			e = 0.0f;
	}
	E[g] = e;
}

// -------------------------------------------------------------------------------------------
// K3 mappability paint.  Semantics (svs.c:363-371): rows in file order, END INCLUSIVE,
// a later row overwrites an earlier one.
//
// paint_sorted_kernel: rows sorted by start with row k+1 starting at or after row k's end (the
// bedGraph-like layout of README.md:77-88; abutting rows share one base, which the later row
// wins).  Each workgroup owns a tile of bases; a base x is covered by the LAST row with
// start <= x, if that row's end >= x.  One pass, every float written exactly once, no memset.
// paint_winner_kernel / paint_resolve_kernel: any row order -- atomicMax of the row index per
// base, then map[i] = val[winner[i]].
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void paint_sorted_kernel(const int32_t *__restrict__ start,
		const int32_t *__restrict__ end, const float *__restrict__ val, int64_t m, float *__restrict__ map,
		int64_t L)
{
	constexpr int kPerThread = 4;
	const int64_t tile = (int64_t) blockDim.x * kPerThread;
	for (int64_t base = (int64_t) blockIdx.x * tile; base < L; base += (int64_t) gridDim.x * tile) {
		const int64_t x0 = base + (int64_t) threadIdx.x * kPerThread;
		if (x0 >= L)
			continue;
		// last row with start <= x0 (upper_bound - 1)
		int64_t lo = 0, hi = m;
		while (lo < hi) {
			const int64_t mid = (lo + hi) >> 1;
			if ((int64_t) start[mid] <= x0)
				lo = mid + 1;
			else
				hi = mid;
		}
		int64_t k = lo - 1;
		float out[kPerThread];
#pragma unroll
		for (int e = 0; e < kPerThread; e++) {
			const int64_t x = x0 + e;
			while (k + 1 < m && (int64_t) start[k + 1] <= x)
				k++;
			out[e] = (k >= 0 && (int64_t) end[k] >= x) ? val[k] : 0.0f;
		}
		if (x0 + kPerThread <= L)
			*reinterpret_cast<float4 *>(map + x0) = *reinterpret_cast<const float4 *>(out);
		else
			for (int e = 0; x0 + e < L; e++)
				map[x0 + e] = out[e];
	}
}

__global__ __launch_bounds__(256) void paint_winner_kernel(const int32_t *__restrict__ start,
		const int32_t *__restrict__ end, int64_t m, int32_t *__restrict__ winner, int64_t L)
{
	// one wave per row; lanes stride over the row's bases
	const int lane = threadIdx.x & (kWave - 1);
	const int64_t wave = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) / kWave;
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
	const float *map; // may be null
	const int32_t *item_iv;
	const int32_t *item_start;
	const int32_t *item_end;
	int64_t n_items;
	int32_t *observed;  // [n_iv], zeroed before launch
	double *map_part;   // [n_items]
};

__global__ __launch_bounds__(256) void interval_reduce_kernel(ReduceArgs a)
{
	const int lane = threadIdx.x & (kWave - 1);
	const int64_t item = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) / kWave;
	if (item >= a.n_items)
		return;
	const int64_t s = a.item_start[item], e = a.item_end[item];

	// ---- depth: int16, 8 per lane
	int acc = 0;
	{
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
	if (a.map) {
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
// K4 (chain) + K5 interval_score: the serial float32 expected_rd accumulation
// (likelihood.c:111,115-119), then lpoisson x3, the int-truncated max, the c-score and the CN
// call (likelihood.c:96-105,131-168).
//
// One lane per interval; lanes are handed intervals in descending window count (order[]) so the
// lanes of a wave finish together.  Each GC window contributes k equal float adds which
// conga_repeat_add_f32 collapses exactly (serial_f32.h); windows are walked left to right, so
// the rounding sequence is the reference's.
// -------------------------------------------------------------------------------------------
// Long intervals: one wave walks one interval, 64 GC windows per step.
//
// Inside one binade of the accumulator every window advances the mantissa by k * delta ulps
// (conga_step_for), an INTEGER, so the 64 per-window advances are combined with a wave prefix sum.
// A window is "regular" when its whole run of k adds stays below the binade top and is not an
// exact tie; the first irregular window (ballot + ffs) is applied with the scalar routine
// conga_repeat_add_f32 -- which performs the real rounding -- and the scan restarts behind it.
// Irregular windows are rare (one per binade crossing, i.e. O(log) per interval), so a 5 Mb
// interval costs ~800 wave steps instead of 50,000 dependent window updates on one lane.
// The result is the same bit pattern as the per-base loop of likelihood.c:115-119.
constexpr int kLongWindows = 192; // intervals with more GC windows than this take the wave path

struct ChainArgs {
	const int32_t *start;
	const int32_t *end;
	const int32_t *order; // longest first; the first n_long entries are the long intervals
	int64_t n_long;
	const uint8_t *gc_like;
	int64_t n_win;
	int32_t step;
	const float *E;
	float *expected; // [n_iv]
};

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
#pragma unroll
	for (int o = 1; o < kWave; o <<= 1) {
		const uint32_t t = __shfl_up(v, o, kWave);
		if (lane >= o)
			v += t;
	}
	return v;
}

__device__ __forceinline__ float compose_f32(uint32_t es, uint32_t ms)
{
	return (ms == 0x1000000u) ? conga_bits_f32((es + 1u) << 23) : conga_bits_f32((es << 23) | (ms & 0x7FFFFFu));
}

__global__ __launch_bounds__(256) void chain_long_kernel(ChainArgs a)
{
	__shared__ float sE[kGcBins];
	for (int g = threadIdx.x; g < kGcBins; g += blockDim.x)
		sE[g] = a.E[g];
	__syncthreads();

	const int lane = threadIdx.x & (kWave - 1);
	const int64_t slot = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) / kWave;
	if (slot >= a.n_long)
		return; // wave-uniform
	const int32_t iv = a.order[slot];
	const int64_t s0 = a.start[iv], e0 = a.end[iv];
	const int64_t step = a.step;
	const int64_t w_first = s0 / step;
	const int64_t w_end = (e0 - 1) / step + 1;

	float s = 0.0f; // wave-uniform accumulator
	for (int64_t wb = w_first; wb < w_end; wb += kWave) {
		const int64_t w = wb + lane;
		const bool active = w < w_end;
		uint32_t k = 0, bc = 0;
		float c = 0.0f;
		if (active) {
			const int64_t lo = (w * step > s0) ? w * step : s0;
			const int64_t hi = ((w + 1) * step < e0) ? (w + 1) * step : e0;
			k = (uint32_t) (hi - lo);
			const uint32_t g = a.gc_like[(w < a.n_win) ? w : a.n_win - 1];
			c = (g < (uint32_t) kGcBins) ? sE[g] : 0.0f;
			bc = conga_f32_bits(c);
		}
		unsigned long long todo = __ballot(active);
		while (todo) {
			const uint32_t bs = conga_f32_bits(s);
			const uint32_t es = bs >> 23;
			const uint32_t ms = (bs & 0x7FFFFFu) | 0x800000u;
			const bool in = (todo >> lane) & 1ull;
			const conga_step st = conga_step_for(es & 0xFFu, bc);
			const bool valid = in && st.delta != 0xFFFFFFFFu && !st.tie && !(bs >> 31) && !(bc >> 31);
			uint32_t adv = 0;
			if (valid) {
				const uint64_t a64 = (uint64_t) k * st.delta;
				adv = (a64 > (1u << 25)) ? (1u << 25) : (uint32_t) a64;
			}
			const uint32_t incl = wave_incl_scan_u32(adv, lane);
			const uint32_t pre = incl - adv;
			const uint32_t m = ms + pre; // mantissa in front of this lane's window (< 2^32)
			bool ok = true;
			if (in)
				ok = valid && (st.delta == 0 || (m <= st.lim && (uint64_t) (k - 1u) * st.delta <= (uint64_t) (st.lim - m)));
			const unsigned long long bad = __ballot(!ok);
			if (bad == 0) {
				const uint32_t total = __shfl(incl, kWave - 1, kWave);
				if (total)
					s = compose_f32(es, ms + total);
				todo = 0;
			} else {
				const int fb = __ffsll((long long) bad) - 1;
				const uint32_t pre_fb = __shfl(pre, fb, kWave);
				if (pre_fb)
					s = compose_f32(es, ms + pre_fb); // exact state in front of the irregular window
				const float c_fb = __shfl(c, fb, kWave);
				const uint32_t k_fb = __shfl(k, fb, kWave);
				s = conga_repeat_add_f32(s, c_fb, k_fb); // real adds where rounding is not a constant step
				todo &= ~((2ull << fb) - 1ull);
			}
		}
	}
	if (lane == 0)
		a.expected[iv] = s;
}

struct ScoreArgs {
	const int32_t *start;
	const int32_t *end;
	const uint8_t *type;     // 'D' / 'E' per interval
	const int32_t *order;    // processing order (longest first)
	int64_t n_iv;
	const uint8_t *gc_like;
	int64_t n_win;
	int32_t step;
	const float *E;
	const int32_t *observed;
	const double *map_part;  // may be null
	const int32_t *item_first; // [n_iv + 1] first work item of each interval
	const int32_t *support;  // split-read support per interval (may be null)
	int32_t has_map;
	int64_t n_long;          // slots below this were computed by chain_long_kernel
	const float *expected_long; // [n_iv], valid for those slots
	conga_result *out;
};

__device__ __forceinline__ double lpoisson_dev(int observed, double lambda)
{
	// likelihood.c:96-105
	if (lambda == 0.0)
		lambda = 0.01;
	return (double) observed * log(lambda) - lambda - lgamma((double) (observed + 1));
}

__device__ __forceinline__ int trunc_max(double x, double y)
{
	// common.c:262-268: max() takes ints, so both arguments are truncated toward zero first
	const int xi = (int) x, yi = (int) y;
	return (xi < yi) ? yi : xi;
}

__device__ __forceinline__ void score_interval(int observed, float expected, uint8_t type, conga_result &r)
{
	const double ex = (double) expected;
	if (type == CONGA_DELETION) {
		r.lhomo = lpoisson_dev(observed, 0.0);
		r.lhetero = lpoisson_dev(observed, 0.5 * ex);
		r.lnone = lpoisson_dev(observed, ex);
		r.copy_number = ((float) observed < (expected / 4.0f)) ? 2 : 1; // likelihood.c:146
	} else {
		r.lhomo = lpoisson_dev(observed, (double) (2.0f * expected)); // int * float stays float
		r.lhetero = lpoisson_dev(observed, 1.5 * ex);
		r.lnone = lpoisson_dev(observed, ex);
		r.copy_number = (r.lhomo > r.lhetero) ? 2 : 1; // likelihood.c:164
	}
	r.score = (double) trunc_max(r.lhomo, r.lhetero) / r.lnone; // likelihood.c:138,160
	r.observed = observed;
	r.expected = expected;
}

__global__ __launch_bounds__(256) void interval_score_kernel(ScoreArgs a)
{
	__shared__ float sE[kGcBins];
	for (int g = threadIdx.x; g < kGcBins; g += blockDim.x)
		sE[g] = a.E[g];
	__syncthreads();

	const int64_t slot = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= a.n_iv)
		return;
	const int32_t iv = a.order[slot];
	const int64_t s = a.start[iv], e = a.end[iv];

	float ex = 0.0f;
	if (slot < a.n_long) {
		ex = a.expected_long[iv];
	} else if (e > s) {
		int64_t w = s / a.step;
		int64_t pos = s;
		const uint32_t *gc32 = reinterpret_cast<const uint32_t *>(a.gc_like);
		const int64_t w_last = a.n_win - 1;
		uint32_t word = 0;
		int64_t word_idx = -1;
		while (pos < e) {
			int64_t next = (w + 1) * (int64_t) a.step;
			if (next > e)
				next = e;
			const int64_t wc = (w < w_last) ? w : w_last; // window index clamped at the chromosome end
			if ((wc >> 2) != word_idx) {
				word_idx = wc >> 2;
				word = gc32[word_idx]; // gc arrays are padded to a multiple of 4 bytes
			}
			const uint32_t g = (word >> (8 * (wc & 3))) & 0xFFu;
			const float c = (g < kGcBins) ? sE[g] : 0.0f;
			ex = conga_repeat_add_f32(ex, c, (uint32_t) (next - pos));
			pos = next;
			w++;
		}
	}

	conga_result r;
	r.rp = 0;
	r.border_rp = 0;
	r.reserved = 0;
	r.mappability = 0.0;
	const uint8_t type = a.type[iv];
	score_interval(a.observed[iv], ex, type, r);
	if (a.has_map) {
		double ms = 0.0;
		for (int32_t it = a.item_first[iv]; it < a.item_first[iv + 1]; it++)
			ms += a.map_part[it];
		r.mappability = ms / (double) (e - s); // likelihood.c:128
	}
	if (a.support) {
		if (type == CONGA_DELETION)
			r.border_rp = a.support[iv];
		else
			r.rp = a.support[iv];
	}
	a.out[iv] = r;
}

} // namespace conga
