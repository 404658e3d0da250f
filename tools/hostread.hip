// hostread.hip -- what the HOST pays for its first read of pinned memory the GPU has just written (the fetch of a small step:
// 0.27-0.34 ms right behind a compute against 0.05 ms for records it has read before, bench.py CONGA_BENCH_PHASES=1).
//   hipcc --offload-arch=gfx950 -O2 -o tools/hostread tools/hostread.hip && tools/hostread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void fill(uint64_t *p, size_t n, uint64_t v)
{
	for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
		p[i] = v + i;
}

static double now_us()
{
	return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
	const unsigned flags[] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocCoherent, hipHostMallocNumaUser};
	const char *names[] = {"default", "non-coherent", "coherent", "numa-user"};
	hipStream_t st;
	hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	uint64_t *dev = nullptr;
	hipMalloc(&dev, 4 << 20);
	for (int f = 0; f < 4; f++)
		for (size_t kb : {4, 48, 512, 2560}) {
			const size_t n = kb * 1024 / 8;
			uint64_t *h = nullptr;
			if (hipHostMalloc(&h, n * 8, flags[f]) != hipSuccess) {
				printf("%-13s %5zu KB: hipHostMalloc failed\n", names[f], kb);
				continue;
			}
			memset(h, 0, n * 8);
			std::vector<uint64_t> out(n);
			double by_kernel = 0, again = 0, by_copy = 0, waits = 0;
			const int reps = 20;
			volatile uint64_t sink = 0;
			for (int r = 0; r < reps; r++) {
				hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, st, h, n, (uint64_t) r);
				double t0 = now_us();
				hipStreamSynchronize(st);
				double t1 = now_us();
				memcpy(out.data(), h, n * 8); // the host's first read of what the kernel wrote
				double t2 = now_us();
				sink += out[n / 2];
				memcpy(out.data(), h, n * 8);
				double t3 = now_us();
				sink += out[n / 3];
				waits += t1 - t0, by_kernel += t2 - t1, again += t3 - t2;
				hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, st, dev, n, (uint64_t) r);
				hipMemcpyAsync(h, dev, n * 8, hipMemcpyDeviceToHost, st);
				hipStreamSynchronize(st);
				t0 = now_us();
				memcpy(out.data(), h, n * 8); // ... of what the copy engine wrote
				by_copy += now_us() - t0;
				sink += out[n / 4];
			}
			printf("%-13s %5zu KB: first read behind a kernel's stores %8.1f us, read again %7.1f us, first read behind a D2H copy %8.1f us (sync %6.1f us)\n", names[f], kb,
					by_kernel / reps, again / reps, by_copy / reps, waits / reps);
			hipHostFree(h);
		}
	return 0;
}
