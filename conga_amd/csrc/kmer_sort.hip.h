// kmer_sort.hip.h -- the stable sort behind the 10-mer index: positions 0, 1, 2, ... ordered by their 10-mer's hash.
//
// build_hash_table (split_read.c:394-440) appends every position to its 10-mer's bucket in increasing order; a STABLE sort of
// the positions by the hash gives exactly those buckets, one behind the other (split_map.hip.h).  Least-significant-digit radix
// sort, three passes of 7 bits over the 21-bit key (2^20 = "no valid 10-mer here": behind every bucket):
//   radix_hist_kernel     one WAVE per tile of 4 096 items: how many of each digit (128 counters in LDS)
//   radix_scan_kernel     per digit, an exclusive scan over the tiles' counts (one workgroup per digit) ...
//   radix_base_kernel     ... and over the digits' totals: where each (digit, tile) run starts
//   radix_scatter_kernel  one wave per tile again, 64 consecutive items per step IN ORDER: an item's place is its run's next free
//                         slot plus its rank among the step's equal digits -- seven ballots give every lane the mask of its
//                         equals, the lanes below it are in front of it: stable without a sort inside the tile
// 128 digits keep a tile's 4 096 items in runs of ~32 per digit -- one cache line each -- so the scatter writes whole lines.
// Runs once per reference sequence (the index is resident afterwards), never in a sample's step; 8 bytes in and out per item and
// pass.  (Rounds 3's first version called rocPRIM's radix_sort_pairs here: the one library call of the engine, now gone.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace conga {

constexpr int kRadixBits = 7, kRadixBins = 1 << kRadixBits, kRadixTile = 4096, kRadixWaves = 4;

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
__global__ __launch_bounds__(64 * kRadixWaves) void radix_scatter_kernel(const uint32_t *__restrict__ keys_in, const int32_t *__restrict__ vals_in, uint32_t n,
		int shift, uint32_t n_tiles, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ base, uint32_t *__restrict__ keys_out,
		int32_t *__restrict__ vals_out)
{
	__shared__ uint32_t s_off[kRadixWaves][kRadixBins];
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t tile = blockIdx.x * kRadixWaves + (uint32_t) wv;
	if (tile >= n_tiles)
		return;
	uint32_t *off = s_off[wv];
	off[lane] = base[lane] + counts[(uint64_t) lane * n_tiles + tile];
	off[lane + 64] = base[lane + 64] + counts[(uint64_t) (lane + 64) * n_tiles + tile];
	radix_wave_sync();
	const uint32_t i0 = tile * (uint32_t) kRadixTile;
	const unsigned long long below = (1ull << lane) - 1ull;
	for (int s = 0; s < kRadixTile / 64; s++) {
		const uint32_t i = i0 + (uint32_t) s * 64u + (uint32_t) lane;
		const bool valid = i < n;
		const uint32_t key = valid ? keys_in[i] : 0u;
		const int32_t val = !valid ? 0 : vals_in ? vals_in[i] : (int32_t) i;
		const uint32_t d = radix_digit(key, shift);
		unsigned long long same = __ballot(valid); // the lanes that hold the same digit as this one
#pragma unroll
		for (int b = 0; b < kRadixBits; b++) {
			const bool bit = (d >> b) & 1u;
			const unsigned long long m = __ballot(bit);
			same &= bit ? m : ~m;
		}
		const uint32_t rank = (uint32_t) __popcll(same & below);
		const uint32_t start = off[d]; // (every lane of a group reads the same counter before its last lane moves it on)
		radix_wave_sync();
		if (valid && (same >> lane) < 2ull) // the group's highest lane: nothing of `same` above it
			off[d] = start + (uint32_t) __popcll(same);
		radix_wave_sync();
		if (valid) {
			keys_out[start + rank] = key;
			vals_out[start + rank] = val;
		}
	}
}

} // namespace conga
