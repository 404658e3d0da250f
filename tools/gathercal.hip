// gathercal.hip -- what the machine does with RANDOM 4-byte probes: the access pattern of a hash-index lookup (split_map_kernel's
// bucket searches), for which the streaming roofline says nothing.  Every lane reads 4 bytes at pseudo-random dword addresses of a
// table that fits one XCD's L2 (2 MiB), the Infinity Cache (128 MiB) or neither (2 GiB):
//   probe     the addresses do not depend on what was read (as many requests in flight as the hardware takes)
//   chase     each address is computed from the value read before (a binary search, a chain walk: latency x occupancy)
// Prints probes per second from HIP events; under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` the counter divided by the probes
// printed here is what ONE probe costs in fabric traffic.
//   hipcc --offload-arch=gfx950 -O3 -o tools/gathercal tools/gathercal.hip && tools/gathercal
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
	x ^= x >> 16;
	x *= 0x7feb352du;
	x ^= x >> 15;
	x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}

constexpr int kProbes = 64;

__global__ __launch_bounds__(256) void probe(const uint32_t *__restrict__ t, uint32_t mask, unsigned *sink)
{
	const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
#pragma unroll 8
	for (int k = 0; k < kProbes; k++)
		acc += t[mix(id * (uint32_t) kProbes + (uint32_t) k) & mask];
	if (acc == 0x12345678u)
		*sink = acc;
}

__global__ __launch_bounds__(256) void chase(const uint32_t *__restrict__ t, uint32_t mask, unsigned *sink)
{
	const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t at = mix(id) & mask, acc = 0;
	for (int k = 0; k < kProbes; k++) {
		const uint32_t v = t[at];
		acc += v;
		at = mix(v + id + (uint32_t) k) & mask;
	}
	if (acc == 0x12345678u)
		*sink = acc;
}

__global__ void fill(uint32_t *t, size_t n)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
		t[i] = mix((uint32_t) i);
}

int main()
{
	const size_t bytes = (size_t) 2 << 30;
	uint32_t *buf;
	unsigned *sink;
	CHECK(hipMalloc(&buf, bytes));
	CHECK(hipMalloc(&sink, 4));
	hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, buf, bytes / 4);
	CHECK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const size_t sizes[3] = {(size_t) 2 << 20, (size_t) 128 << 20, bytes};
	const char *names[3] = {"2MiB", "128MiB", "2GiB"};
	const int grid = 256 * 32; // 8 192 workgroups of 256: 2.1 M lanes, 134 M probes per launch
	printf("{\"probes_per_launch\": %zu", (size_t) grid * 256 * kProbes);
	for (int s = 0; s < 3; s++) {
		const uint32_t mask = (uint32_t) (sizes[s] / 4 - 1);
		for (int mode = 0; mode < 2; mode++) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; rep++) {
				CHECK(hipEventRecord(e0, 0));
				if (mode == 0)
					hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, buf, mask, sink);
				else
					hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, 0, buf, mask, sink);
				CHECK(hipEventRecord(e1, 0));
				CHECK(hipEventSynchronize(e1));
				float ms;
				CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best)
					best = ms;
			}
			printf(", \"%s_%s_ms\": %.4f, \"%s_%s_Gprobes_per_s\": %.2f", mode ? "chase" : "probe", names[s], best, mode ? "chase" : "probe", names[s],
					(double) grid * 256 * kProbes / (best * 1e-3) / 1e9);
		}
	}
	printf("}\n");
	return 0;
}
