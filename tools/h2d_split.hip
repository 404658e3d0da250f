// h2d_split.hip -- host-to-device rate of one pinned copy against the same bytes as 2 / 4 concurrent pieces on streams of
// their own (do several SDMA engines together get more out of the link than one?).  100 MB per copy, like one 1x sample.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
	const size_t n = (size_t) 102 << 20;
	uint8_t *h = nullptr, *d = nullptr;
	CHECK(hipHostMalloc(&h, n));
	memset(h, 1, n);
	CHECK(hipMalloc(&d, n));
	hipStream_t st[4];
	for (auto &s : st)
		CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	for (int pieces : {1, 2, 4, 1, 2, 4}) {
		double best = 1e9;
		for (int rep = 0; rep < 12; rep++) {
			CHECK(hipDeviceSynchronize());
			const auto t0 = std::chrono::steady_clock::now();
			for (int k = 0; k < pieces; k++)
				CHECK(hipMemcpyAsync(d + n / pieces * k, h + n / pieces * k, n / pieces, hipMemcpyHostToDevice, st[k]));
			for (int k = 0; k < pieces; k++)
				CHECK(hipStreamSynchronize(st[k]));
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			if (rep >= 2 && ms < best)
				best = ms;
		}
		printf("%d piece(s): %.3f ms = %.1f GB/s\n", pieces, best, n / best / 1e6);
	}
	return 0;
}
