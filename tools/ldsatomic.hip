// ldsatomic.hip -- how the LDS serves ds_add_u32 (no return) on gfx950: which lanes of a wave conflict.
// A workgroup of 256 lanes adds into a 16 K-dword LDS array, 4096 times per lane, at an address pattern per lane;
// 8 workgroups per CU (the tuple pass's shape).  Prints the time per pattern; the ratio to pattern 0 is what counts.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ldsatomic tools/ldsatomic.hip && tools/ldsatomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 8) void k(int pattern, int iters, unsigned *sink)
{
	__shared__ unsigned h[4096];
	for (int i = threadIdx.x; i < 4096; i += 256)
		h[i] = 0;
	__syncthreads();
	const unsigned lane = threadIdx.x & 63u, l32 = lane & 31u, half = lane >> 5;
	unsigned a;
	switch (pattern) {
	case 0: a = lane; break;                          // 64 consecutive dwords: banks 0..31 twice, all addresses distinct
	case 1: a = l32; break;                           // lanes l and l + 32 on the SAME address
	case 2: a = l32 + 64u * half; break;              // l, l + 32: same bank (mod 32 and mod 64), different address
	case 3: a = l32 + 32u * half + 64u; break;        // as 0 (control)
	case 4: a = 2u * lane; break;                     // stride 2: banks mod 32 used twice within each half
	case 5: a = 32u * l32 + half; break;              // every lane of a half on bank `half` ... 32 rows: worst case
	case 6: a = l32 * 33u + 7u * half; break;         // skewed rows
	case 7: a = (l32 * 101u + (lane * 7u >> 4)) ; break; // the tuple pass's layout with a slowly drifting bin
	case 8: a = (lane >> 1); break;                   // pairs of neighbours on one address
	case 9: a = 0; break;                             // all on one address
	default: a = lane; break;
	}
	a &= 4095u;
	for (int i = 0; i < iters; i++) {
		atomicAdd(&h[a], 1u);
		a = (a + 128u) & 4095u; // (same banks, another row: keeps the pattern, defeats any same-address shortcut across instructions)
	}
	__syncthreads();
	if (threadIdx.x == 0)
		sink[blockIdx.x] = h[5];
}

int main()
{
	unsigned *sink;
	CHECK(hipMalloc(&sink, 4096 * 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int iters = 4096, grid = 2048;
	for (int p = 0; p < 10; p++) {
		float best = 1e30f;
		for (int r = 0; r < 3; r++) {
			CHECK(hipEventRecord(e0, 0));
			hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p, iters, sink);
			CHECK(hipEventRecord(e1, 0));
			CHECK(hipEventSynchronize(e1));
			float ms;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
		}
		// wave-instructions per CU: grid / 256 CUs * 4 waves * iters
		const double per_cu = (double) grid / 256.0 * 4.0 * iters;
		printf("pattern %d: %.3f ms  = %.2f ns per wave-instruction per CU\n", p, best, best * 1e6 / per_cu);
	}
	return 0;
}
