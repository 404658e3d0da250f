// kmer_sort.hip.h -- the stable sort behind the 10-mer index: positions 0, 1, 2, ... ordered by their 10-mer's hash.
//
// build_hash_table (split_read.c:394-440) appends every position to its 10-mer's bucket in increasing order; a STABLE sort of
// the positions by the hash gives exactly those buckets, one behind the other (split_map.hip.h).  Least-significant-digit radix
// sort, three passes of 7 bits over the 21-bit key (2^20 = "no valid 10-mer here": behind every bucket):
//   radix_hist_kernel     one WAVE per tile of 2 048 items: how many of each digit (128 counters in LDS)
//   radix_scan_kernel     per digit, an exclusive scan over the tiles' counts (one workgroup per digit) ...
//   radix_base_kernel     ... and over the digits' totals: where each (digit, tile) run starts
//   radix_scatter_kernel  one wave per tile again, 64 consecutive items per step IN ORDER: an item's place is its run's next free
//                         slot plus its rank among the step's equal digits -- seven ballots give every lane the mask of its
//                         equals, the lanes below it are in front of it: stable without a sort inside the tile.  The tile is
//                         put in order in LDS (16 KB per wave) and written out in runs: neighbouring lanes, neighbouring addresses
// Runs once per reference sequence (the index is resident afterwards), never in a sample's step; 8 bytes in and out per item and
// pass.  (Rounds 3's first version called rocPRIM's radix_sort_pairs here: the one library call of the engine, now gone.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace conga {

constexpr int kRadixBits = 7, kRadixBins = 1 << kRadixBits, kRadixTile = 2048, kRadixWaves = 4;

// LDS traffic between the lanes of ONE wave: the hardware runs a wave's LDS instructions in order; this keeps the compiler from
// moving them across
__device__ __forceinline__ void radix_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t radix_digit(uint32_t key, int shift)
{
	return (key >> shift) & (uint32_t) (kRadixBins - 1);
}

// counts[digit * n_tiles + tile]
__global__ __launch_bounds__(64 * kRadixWaves) void radix_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n, int shift, uint32_t n_tiles,
		uint32_t *__restrict__ counts)
{
	__shared__ uint32_t s_cnt[kRadixWaves][kRadixBins];
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t tile = blockIdx.x * kRadixWaves + (uint32_t) wv;
	uint32_t *cnt = s_cnt[wv];
	cnt[lane] = 0;
	cnt[lane + 64] = 0;
	radix_wave_sync();
	if (tile < n_tiles) {
		const uint32_t i0 = tile * (uint32_t) kRadixTile;
		for (int s = 0; s < kRadixTile / 64; s++) {
			const uint32_t i = i0 + (uint32_t) s * 64u + (uint32_t) lane;
			if (i < n)
				atomicAdd(&cnt[radix_digit(keys[i], shift)], 1u);
		}
		radix_wave_sync();
		counts[(uint64_t) lane * n_tiles + tile] = cnt[lane];
		counts[(uint64_t) (lane + 64) * n_tiles + tile] = cnt[lane + 64];
	}
}

// one workgroup per digit: counts[digit][0 .. n_tiles) -> exclusive prefix in place, the digit's total to totals[digit]
__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t *__restrict__ counts, uint32_t n_tiles, uint32_t *__restrict__ totals)
{
	__shared__ uint32_t s_wave[16];
	__shared__ uint32_t s_carry;
	uint32_t *row = counts + (uint64_t) blockIdx.x * n_tiles;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x == 0)
		s_carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n_tiles; base += 1024) {
		const uint32_t t = base + threadIdx.x;
		const uint32_t v = t < n_tiles ? row[t] : 0u;
		uint32_t inc = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t p = (uint32_t) __shfl_up((int) inc, o, 64);
			if (lane >= o)
				inc += p;
		}
		if (lane == 63)
			s_wave[wv] = inc;
		__syncthreads();
		uint32_t before = s_carry;
		for (int k = 0; k < wv; k++)
			before += s_wave[k];
		if (t < n_tiles)
			row[t] = before + inc - v;
		__syncthreads();
		if (threadIdx.x == 1023)
			s_carry = before + inc;
		__syncthreads();
	}
	if (threadIdx.x == 0)
		totals[blockIdx.x] = s_carry;
}

// exclusive scan over the 128 digit totals (one workgroup of 128 threads)
__global__ __launch_bounds__(kRadixBins) void radix_base_kernel(const uint32_t *__restrict__ totals, uint32_t *__restrict__ base)
{
	__shared__ uint32_t s[kRadixBins];
	s[threadIdx.x] = totals[threadIdx.x];
	__syncthreads();
	uint32_t sum = 0;
	for (int k = 0; k < (int) threadIdx.x; k++)
		sum += s[k];
	base[threadIdx.x] = sum;
}

// keys_out / vals_out = the items ordered by this pass's digit, equal digits in their order of arrival.  vals_in == nullptr: the
// values are the items' indices (the first pass: position i).
// The tile is ordered in LDS first (where every digit's run begins inside the tile follows from the tile's own counts; an item's
// place in its run from the 64-item steps taken in order) and then written out front to back: neighbouring lanes write
// neighbouring addresses of a run -- 4-byte stores scattered one by one left L2 as partial lines, five times the bytes sorted
// (profiles/r03e_sr_pmc_fetch_write.json: WRITE_SIZE 2.2 GB per pass of 54 M items).
__global__ __launch_bounds__(64 * kRadixWaves) void radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in, uint32_t n,
		int shift, uint32_t n_tiles, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ base, uint32_t *__restrict__ keys_out,
		int32_t *__restrict__ vals_out)
{
	__shared__ uint32_t s_key[kRadixWaves][kRadixTile];
	__shared__ int32_t s_val[kRadixWaves][kRadixTile];
	__shared__ uint32_t s_next[kRadixWaves][kRadixBins];  // next free place of a digit's run inside the tile
	__shared__ uint32_t s_first[kRadixWaves][kRadixBins]; // where the run begins inside the tile
	__shared__ uint32_t s_glob[kRadixWaves][kRadixBins];  // ... and in the output
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t tile = blockIdx.x * kRadixWaves + (uint32_t) wv;
	if (tile >= n_tiles)
		return;
	uint32_t *key = s_key[wv], *next = s_next[wv], *first = s_first[wv], *glob = s_glob[wv];
	int32_t *val = s_val[wv];
	const uint32_t i0 = tile * (uint32_t) kRadixTile;
	const uint32_t tile_n = min((uint32_t) kRadixTile, n - i0);
	// counts[] holds, per digit, how many items the tiles IN FRONT of this one have (radix_scan_kernel): with base[] that is where
	// the tile's run of the digit begins in the output, and the next tile's entry (the digit's total behind the last tile) minus
	// this one's is how long the run is
	{
		auto run_of = [&](int d, uint32_t &before, uint32_t &len) {
			before = counts[(uint64_t) d * n_tiles + tile];
			const uint32_t total = (d + 1 < kRadixBins ? base[d + 1] : n) - base[d];
			len = (tile + 1 < n_tiles ? counts[(uint64_t) d * n_tiles + tile + 1] : total) - before;
		};
		uint32_t c0, c1, len0, len1;
		run_of(lane, c0, len0);
		run_of(lane + 64, c1, len1);
		glob[lane] = base[lane] + c0;
		glob[lane + 64] = base[lane + 64] + c1;
		// exclusive scan of the 128 lengths: a lane holds digits `lane` and `lane + 64`
		uint32_t inc0 = len0, inc1 = len1;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t p0 = (uint32_t) __shfl_up((int) inc0, o, 64), p1 = (uint32_t) __shfl_up((int) inc1, o, 64);
			if (lane >= o) {
				inc0 += p0;
				inc1 += p1;
			}
		}
		const uint32_t low_total = (uint32_t) __shfl((int) inc0, 63, 64);
		first[lane] = next[lane] = inc0 - len0;
		first[lane + 64] = next[lane + 64] = low_total + inc1 - len1;
	}
	radix_wave_sync();
	const unsigned long long below = (1ull << lane) - 1ull;
	for (int s = 0; s < kRadixTile / 64; s++) {
		const uint32_t i = i0 + (uint32_t) s * 64u + (uint32_t) lane;
		const bool valid = i < n;
		const uint32_t k = valid ? keys_in[i] : 0u;
		const int32_t v = !valid ? 0 : vals_in ? vals_in[i] : (int32_t) i;
		const uint32_t d = radix_digit(k, shift);
		unsigned long long same = __ballot(valid); // the lanes that hold the same digit as this one
#pragma unroll
		for (int b = 0; b < kRadixBits; b++) {
			const bool bit = (d >> b) & 1u;
			const unsigned long long m = __ballot(bit);
			same &= bit ? m : ~m;
		}
		const uint32_t rank = (uint32_t) __popcll(same & below);
		const uint32_t start = next[d]; // (every lane of a group reads the same counter before its last lane moves it on)
		radix_wave_sync();
		if (valid && (same >> lane) < 2ull) // the group's highest lane: nothing of `same` above it
			next[d] = start + (uint32_t) __popcll(same);
		if (valid) {
			key[start + rank] = k;
			val[start + rank] = v;
		}
		radix_wave_sync();
	}
	for (uint32_t j = (uint32_t) lane; j < tile_n; j += 64u) {
		const uint32_t k = key[j];
		const uint32_t d = radix_digit(k, shift);
		const uint32_t at = glob[d] + (j - first[d]);
		keys_out[at] = k;
		vals_out[at] = val[j];
	}
}

} // namespace conga
