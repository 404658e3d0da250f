// fetchcal.hip -- what rocprofv3's FETCH_SIZE reports on gfx950 for streaming reads of different widths.
// MI355X_MICROARCH.md (HBM): "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
// Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern."  The kernels below
// read a buffer far larger than the 256 MiB Infinity Cache exactly once, in the widths and shapes the library's kernels use:
//   read16   16 B per lane, coalesced        (the positions in ingest_tuples_kernel)
//   read4     4 B per lane, coalesced        (its MAPQ bytes: four per lane; positions / tile index in depth_tile_kernel)
//   read1     1 B per lane, coalesced
//   gather1   1 B per lane at sorted, slowly advancing addresses (the GC byte of a read's window: ~1 new byte per 100 bases)
// Run each under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and divide the counter by the bytes printed here.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fetchcal tools/fetchcal.hip && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/fetchcal
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read16(const uint4 *p, size_t n, unsigned *sink)
{
	unsigned acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
		const uint4 v = p[i];
		acc += v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x12345678u)
		*sink = acc;
}

__global__ __launch_bounds__(256) void read4(const uint32_t *p, size_t n, unsigned *sink)
{
	unsigned acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
		acc += p[i];
	if (acc == 0x12345678u)
		*sink = acc;
}

__global__ __launch_bounds__(256) void read1(const uint8_t *p, size_t n, unsigned *sink)
{
	unsigned acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
		acc += p[i];
	if (acc == 0x12345678u)
		*sink = acc;
}

// lane i of step s reads byte (s * 64 + i) * 25 / 100 ... i.e. a byte index that advances by one every four lanes:
// what a coordinate-sorted 1x sample does to the per-100-base GC track (0.25 reads per base -> 25 reads per window)
__global__ __launch_bounds__(256) void gather1(const uint8_t *p, size_t n_reads, unsigned *sink)
{
	unsigned acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += (size_t) gridDim.x * blockDim.x)
		acc += p[i / 25];
	if (acc == 0x12345678u)
		*sink = acc;
}

int main()
{
	const size_t bytes = (size_t) 2 << 30; // 2 GiB: eight times the Infinity Cache
	uint8_t *buf;
	unsigned *sink;
	CHECK(hipMalloc(&buf, bytes));
	CHECK(hipMalloc(&sink, 4));
	CHECK(hipMemset(buf, 1, bytes));
	CHECK(hipDeviceSynchronize());
	const int grid = 256 * 8;
	for (int rep = 0; rep < 2; rep++) {
		hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const uint4 *) buf, bytes / 16, sink);
		hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, (const uint32_t *) buf, bytes / 4, sink);
		hipLaunchKernelGGL(read1, dim3(grid), dim3(256), 0, 0, buf, bytes / 4, sink);
		hipLaunchKernelGGL(gather1, dim3(grid), dim3(256), 0, 0, buf, (size_t) 25 * (bytes / 4), sink);
		CHECK(hipDeviceSynchronize());
	}
	printf("{\"read16_bytes\": %zu, \"read4_bytes\": %zu, \"read1_bytes\": %zu, \"gather1_bytes\": %zu}\n", bytes, bytes, bytes / 4, bytes / 4);
	return 0;
}
