// membw.hip -- store-bandwidth calibration for the depth_tile kernel (not part of the product).
// Measures, on the same 5.76 GB buffer the whole-genome read_depth occupies:
//   (a) hipMemsetAsync, (b) a grid-stride 16-byte-store fill, (c) the depth kernel's store pattern
//   (every wave owns a contiguous run of 4000-byte tiles and writes 1 KiB per wave-instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_stride(uint4 *p, size_t n16)
{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
		p[i] = make_uint4(0, 0, 0, 0);
}

template <bool NT> __global__ __launch_bounds__(256) void fill_wave_runs(uint4 *p, size_t n_tiles, size_t tiles_per_wave, int tile16)
{
	const int lane = threadIdx.x & 63;
	const size_t wave = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t t0 = wave * tiles_per_wave;
	size_t t1 = t0 + tiles_per_wave;
	if (t1 > n_tiles)
		t1 = n_tiles;
	for (size_t t = t0; t < t1; t++) {
		uint4 *q = p + t * tile16;
		for (int j = lane; j < tile16; j += 64) {
			typedef unsigned int v4u __attribute__((ext_vector_type(4)));
			if (NT)
				__builtin_nontemporal_store((v4u) (0u), reinterpret_cast<v4u *>(q + j));
			else
				q[j] = make_uint4(0, 0, 0, 0);
		}
	}
}

// tiles dealt round-robin over the waves in runs of `run` consecutive tiles
__global__ __launch_bounds__(256) void fill_round_robin(uint4 *p, size_t n_tiles, int tile16, int run)
{
	const int lane = threadIdx.x & 63;
	const size_t n_waves = (size_t) gridDim.x * 4;
	const size_t wave = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	for (size_t t0 = wave * run; t0 < n_tiles; t0 += n_waves * run)
		for (size_t t = t0; t < t0 + run && t < n_tiles; t++) {
			uint4 *q = p + t * tile16;
			for (int j = lane; j < tile16; j += 64)
				q[j] = make_uint4(0, 0, 0, 0);
		}
}

// the waves of a workgroup share one contiguous region and alternate 4 KiB tiles inside it
template <int BLOCK> __global__ __launch_bounds__(BLOCK) void fill_block_interleaved(uint4 *p, size_t n_tiles, size_t tiles_per_block)
{
	constexpr int W = BLOCK / 64;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const size_t t0 = (size_t) blockIdx.x * tiles_per_block;
	size_t t1 = t0 + tiles_per_block;
	if (t1 > n_tiles)
		t1 = n_tiles;
	for (size_t t = t0 + wv; t < t1; t += W) {
		uint4 *q = p + t * 256;
		for (int j = lane; j < 256; j += 64)
			q[j] = make_uint4(0, 0, 0, 0);
	}
}

int main()
{
	const size_t bytes = 5762066572ull / 16 * 16;
	void *buf;
	CK(hipMalloc(&buf, bytes + 4096));
	hipStream_t st;
	CK(hipStreamCreate(&st));
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	auto time = [&](const char *name, auto fn) {
		for (int w = 0; w < 2; w++)
			fn();
		CK(hipEventRecord(a, st));
		const int reps = 10;
		for (int r = 0; r < reps; r++)
			fn();
		CK(hipEventRecord(b, st));
		CK(hipEventSynchronize(b));
		float ms;
		CK(hipEventElapsedTime(&ms, a, b));
		printf("%-34s %.3f ms  %.0f GB/s\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
	};
	time("hipMemsetAsync", [&] { CK(hipMemsetAsync(buf, 0, bytes, st)); });
	for (int grid : {2048, 4096, 16384})
		time(grid == 2048 ? "fill grid-stride 2048 blocks" : grid == 4096 ? "fill grid-stride 4096 blocks" : "fill grid-stride 16384 blocks",
				[&] { hipLaunchKernelGGL(fill_stride, dim3(grid), dim3(256), 0, st, (uint4 *) buf, bytes / 16); });
	for (int per_cu : {4, 7, 8}) {
		const size_t n_tiles = bytes / 4000, waves = (size_t) 256 * per_cu * 4;
		const size_t tpw = (n_tiles + waves - 1) / waves;
		char nm[64];
		snprintf(nm, sizeof nm, "wave runs 4000 B, %d blocks/CU", per_cu);
		time(nm, [&] { hipLaunchKernelGGL(fill_wave_runs<false>, dim3(256 * per_cu), dim3(256), 0, st, (uint4 *) buf, n_tiles, tpw, 250); });
		snprintf(nm, sizeof nm, "wave runs 4000 B nt, %d blocks/CU", per_cu);
		time(nm, [&] { hipLaunchKernelGGL(fill_wave_runs<true>, dim3(256 * per_cu), dim3(256), 0, st, (uint4 *) buf, n_tiles, tpw, 250); });
	}
	for (int tile_bytes : {4096, 8192}) {
		const size_t n_tiles = bytes / tile_bytes, waves = (size_t) 256 * 7 * 4;
		const size_t tpw = (n_tiles + waves - 1) / waves;
		char nm[64];
		snprintf(nm, sizeof nm, "wave runs %d B aligned, 7 blk/CU", tile_bytes);
		time(nm, [&] { hipLaunchKernelGGL(fill_wave_runs<false>, dim3(256 * 7), dim3(256), 0, st, (uint4 *) buf, n_tiles, tpw, tile_bytes / 16); });
	}
	{
		const size_t n_tiles = bytes / 4096;
		size_t blocks = 256 * 7, tpb = (n_tiles + blocks - 1) / blocks;
		time("block-interleaved 4 waves, 7 blk/CU", [&] { hipLaunchKernelGGL(fill_block_interleaved<256>, dim3(blocks), dim3(256), 0, st, (uint4 *) buf, n_tiles, tpb); });
		blocks = 256 * 2; tpb = (n_tiles + blocks - 1) / blocks;
		time("block-interleaved 16 waves, 2 blk/CU", [&] { hipLaunchKernelGGL(fill_block_interleaved<1024>, dim3(blocks), dim3(1024), 0, st, (uint4 *) buf, n_tiles, tpb); });
		blocks = 256; tpb = (n_tiles + blocks - 1) / blocks;
		time("block-interleaved 16 waves, 1 blk/CU", [&] { hipLaunchKernelGGL(fill_block_interleaved<1024>, dim3(blocks), dim3(1024), 0, st, (uint4 *) buf, n_tiles, tpb); });
		// non-persistent: one 16-wave workgroup per 64 KiB, dispatched in order
		blocks = (n_tiles + 15) / 16;
		time("non-persistent 16 waves x 1 tile", [&] { hipLaunchKernelGGL(fill_block_interleaved<1024>, dim3(blocks), dim3(1024), 0, st, (uint4 *) buf, n_tiles, (size_t) 16); });
		blocks = (n_tiles + 3) / 4;
		time("non-persistent 4 waves x 1 tile", [&] { hipLaunchKernelGGL(fill_block_interleaved<256>, dim3(blocks), dim3(256), 0, st, (uint4 *) buf, n_tiles, (size_t) 4); });
		blocks = (n_tiles + 31) / 32;
		time("non-persistent 4 waves x 8 tiles", [&] { hipLaunchKernelGGL(fill_block_interleaved<256>, dim3(blocks), dim3(256), 0, st, (uint4 *) buf, n_tiles, (size_t) 32); });
	}
	for (int tile_bytes : {4096})
		for (int run : {1, 16}) {
			const size_t n_tiles = bytes / tile_bytes;
			char nm[64];
			snprintf(nm, sizeof nm, "round-robin %d B x%d, 7 blk/CU", tile_bytes, run);
			time(nm, [&] { hipLaunchKernelGGL(fill_round_robin, dim3(256 * 7), dim3(256), 0, st, (uint4 *) buf, n_tiles, tile_bytes / 16, run); });
		}
	return 0;
}
