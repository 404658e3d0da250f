// exitcost.hip -- what a process that used the GPU costs to start and to leave (tools/e2e_quick.sh's wall - main gap).
//   exitcost <device GB> <pinned MB> <0: _exit | 1: free everything first | 2: return from main>
// prints its own times; the caller's clock around it gives what lies behind _exit.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void touch(uint8_t *p, size_t n)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x * 4096)
		p[i] = 1;
}

int main(int argc, char **argv)
{
	const double gb = argc > 1 ? atof(argv[1]) : 0;
	const double mb = argc > 2 ? atof(argv[2]) : 0;
	const int how = argc > 3 ? atoi(argv[3]) : 0;
	auto t0 = std::chrono::steady_clock::now();
	auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return 1;
	const double t_init = ms();
	(void) hipSetDevice(0);
	hipStream_t st;
	(void) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	const double t_stream = ms();
	uint8_t *d = nullptr, *h = nullptr;
	const size_t nb = (size_t) (gb * (1u << 30));
	if (nb && hipMalloc(&d, nb) != hipSuccess)
		return 2;
	const double t_malloc = ms();
	if (nb)
		touch<<<1024, 256, 0, st>>>(d, nb);
	(void) hipStreamSynchronize(st);
	const double t_kernel = ms();
	const size_t np = (size_t) (mb * (1u << 20));
	if (np && hipHostMalloc(&h, np) != hipSuccess)
		return 3;
	const double t_pin = ms();
	if (how == 1) {
		(void) hipFree(d);
		(void) hipHostFree(h);
		(void) hipStreamDestroy(st);
	}
	const double t_free = ms();
	printf("init %.1f, stream %.1f, hipMalloc %.1f, first kernel %.1f, hipHostMalloc %.1f, frees %.1f ms; leaving at %.1f ms\n", t_init,
			t_stream - t_init, t_malloc - t_stream, t_kernel - t_malloc, t_pin - t_kernel, t_free - t_pin, ms());
	fflush(nullptr);
	if (how != 2)
		_exit(0);
	return 0;
}
