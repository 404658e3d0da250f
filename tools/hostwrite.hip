// hostwrite.hip -- calibration of device-initiated stores into pinned host memory (not part of the product).
// The chain kernel writes its 64-byte result records straight into the caller's pinned buffer; this measures what
// PCIe sustains for the store patterns a kernel can produce, next to the copy engine's rate for the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one lane = one 64-byte record (four 16-byte stores, records 64 bytes apart across lanes)
__global__ __launch_bounds__(256) void per_lane_records(uint4 *dst, size_t n_rec)
{
	const size_t r = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_rec)
		return;
	for (int q = 0; q < 4; q++)
		dst[r * 4 + q] = make_uint4((unsigned) r, q, 0, 0);
}

// four lanes = one record: a wave instruction writes 1 KiB contiguous, a wave 4 KiB contiguous
__global__ __launch_bounds__(256) void transposed_records(uint4 *dst, size_t n_rec)
{
	const int lane = threadIdx.x & 63;
	const size_t wave_first = (((size_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 64;
	for (int t = 0; t < 4; t++) {
		const size_t r = wave_first + t * 16 + (lane >> 2);
		if (r < n_rec)
			dst[r * 4 + (lane & 3)] = make_uint4((unsigned) r, t, 0, 0);
	}
}

// plain streaming fill: consecutive lanes, consecutive 16 bytes, grid-stride
__global__ __launch_bounds__(256) void stream_fill(uint4 *dst, size_t n16)
{
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
		dst[i] = make_uint4((unsigned) i, 0, 0, 0);
}

// every wave writes its records first and then keeps computing for `spin` clock ticks (100 MHz): do the stores
// travel while the kernel is still busy, or only when it ends?
__global__ __launch_bounds__(256) void write_then_spin(uint4 *dst, size_t n_rec, long long spin, unsigned *sink)
{
	const size_t r = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n_rec)
		for (int q = 0; q < 4; q++)
			dst[r * 4 + q] = make_uint4((unsigned) r, q, 0, 0);
	unsigned x = (unsigned) r;
	for (long long i = 0; i < spin * 6; i++)
		x = x * 1664525u + 1013904223u;
	if (x == 0xdeadbeefu)
		*sink = x;
}

template <int MODE> __global__ __launch_bounds__(256) void write_then_spin_mode(uint4 *dst, size_t n_rec, long long spin, unsigned *sink)
{
	typedef unsigned int v4u __attribute__((ext_vector_type(4)));
	const size_t r = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (r < n_rec)
		for (int q = 0; q < 4; q++) {
			if (MODE == 1)
				__builtin_nontemporal_store((v4u) {(unsigned) r, (unsigned) q, 0u, 0u}, reinterpret_cast<v4u *>(dst + r * 4 + q));
			else if (MODE == 2) {
				unsigned *w = reinterpret_cast<unsigned *>(dst + r * 4 + q);
				for (int k = 0; k < 4; k++)
					__hip_atomic_store(w + k, (unsigned) r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			} else
				dst[r * 4 + q] = make_uint4((unsigned) r, q, 0, 0);
		}
	if (MODE == 3)
		__threadfence_system();
	if (MODE == 4 && (threadIdx.x & 63) == 0)
		__threadfence_system();
	// pure ALU delay (no clock polling: that is a memory transaction of its own)
	unsigned x = (unsigned) r;
	for (long long i = 0; i < spin * 6; i++)
		x = x * 1664525u + 1013904223u;
	if (x == 0xdeadbeefu)
		*sink = x;
}

int main(int argc, char **argv)
{
	const size_t n_rec = argc > 1 ? (size_t) atol(argv[1]) : 40000;
	const size_t bytes = n_rec * 64;
	void *host, *dev;
	CK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
	CK(hipMalloc(&dev, bytes));
	hipStream_t st;
	CK(hipStreamCreate(&st));
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	auto time = [&](const char *name, auto fn) {
		for (int w = 0; w < 3; w++)
			fn();
		CK(hipStreamSynchronize(st));
		const int reps = 20;
		float total = 0;
		for (int r = 0; r < reps; r++) {
			CK(hipEventRecord(a, st));
			fn();
			CK(hipEventRecord(b, st));
			CK(hipEventSynchronize(b));
			float ms;
			CK(hipEventElapsedTime(&ms, a, b));
			total += ms;
		}
		printf("%-44s %7.1f us  %5.1f GB/s\n", name, total / reps * 1e3, bytes / (total / reps * 1e-3) / 1e9);
	};
	printf("%zu records of 64 bytes = %.2f MB\n", n_rec, bytes / 1e6);
	time("hipMemcpyAsync device -> pinned host", [&] { CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st)); });
	const int blocks = (int) ((n_rec + 255) / 256);
	time("kernel, one lane per record", [&] { hipLaunchKernelGGL(per_lane_records, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec); });
	time("kernel, 4 lanes per record (1 KiB / instr)", [&] { hipLaunchKernelGGL(transposed_records, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec); });
	for (int g : {16, 64, 256, 1024})
	{
		char nm[64];
		snprintf(nm, sizeof nm, "kernel, streaming fill, %d workgroups", g);
		time(nm, [&] { hipLaunchKernelGGL(stream_fill, dim3(g), dim3(256), 0, st, (uint4 *) host, bytes / 16); });
	}
	unsigned *sink;
	CK(hipMalloc((void **) &sink, 4));
	time("kernel, spin 6000 only (no stores)", [&] { hipLaunchKernelGGL(write_then_spin, dim3(blocks), dim3(256), 0, st, (uint4 *) host, (size_t) 0, 6000ll, sink); });
	for (long long us : {0, 30, 60, 100}) {
		char nm[64];
		snprintf(nm, sizeof nm, "kernel, write records then spin %lld us", us);
		time(nm, [&] { hipLaunchKernelGGL(write_then_spin, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec, us * 100, sink); });
	}
	time("  nontemporal stores, then spin 60 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<1>, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec, 6000ll, sink); });
	time("  system-scope atomic stores, spin 60 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<2>, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec, 6000ll, sink); });
	time("  plain + __threadfence_system, spin 60 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<3>, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec, 6000ll, sink); });
	time("  plain + fence by lane 0 only, spin 60 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<4>, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec, 6000ll, sink); });
	time("  plain + __threadfence_system, spin 0 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<3>, dim3(blocks), dim3(256), 0, st, (uint4 *) host, n_rec, 0ll, sink); });
	{
		void *coh;
		CK(hipHostMalloc(&coh, bytes, hipHostMallocCoherent));
		time("  hipHostMallocCoherent, plain, spin 60 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<0>, dim3(blocks), dim3(256), 0, st, (uint4 *) coh, n_rec, 6000ll, sink); });
		time("  hipHostMallocCoherent, plain, spin 0 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<0>, dim3(blocks), dim3(256), 0, st, (uint4 *) coh, n_rec, 0ll, sink); });
		time("  hipHostMallocCoherent, nontemporal, spin 60", [&] { hipLaunchKernelGGL(write_then_spin_mode<1>, dim3(blocks), dim3(256), 0, st, (uint4 *) coh, n_rec, 6000ll, sink); });
		void *nc;
		CK(hipHostMalloc(&nc, bytes, hipHostMallocNonCoherent));
		time("  hipHostMallocNonCoherent, plain, spin 60 us", [&] { hipLaunchKernelGGL(write_then_spin_mode<0>, dim3(blocks), dim3(256), 0, st, (uint4 *) nc, n_rec, 6000ll, sink); });
	}
	time("kernel, one lane per record -> device", [&] { hipLaunchKernelGGL(per_lane_records, dim3(blocks), dim3(256), 0, st, (uint4 *) dev, n_rec); });
	return 0;
}
