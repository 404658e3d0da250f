// packbench.cpp -- the host producer of the packed hand-over by itself (conga_amd/csrc/pack_host.h): how long a pool of T threads
// takes to turn a 1x genome's 25.6 M positions into 10-bit differences, from ordinary memory.  What bench.py's
// `hand_over.packed_encode_timed` can reach is this or the link, whichever is slower.
//   g++ -O3 -std=c++17 -o tools/packbench tools/packbench.cpp -lpthread ; tools/packbench [threads ...]
#include "../conga_amd/csrc/pack_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
	const uint64_t n = 25600000;
	std::vector<int32_t> pos(n);
	uint64_t x = 88172645463325252ull;
	int32_t p = 0;
	for (uint64_t i = 0; i < n; i++) {
		x ^= x << 13, x ^= x >> 7, x ^= x << 17;
		if (i == n / 2)
			p = 0;
		p += (int32_t) (x % 199u) + ((x >> 40) % 4000u == 0 ? 30000 : 0);
		pos[i] = p;
	}
	const uint64_t off[3] = {0, n / 2, n};
	std::vector<uint8_t> out(conga_pack::bound(n, n / 16));
	int spread = 0;
	for (int a = 1; a < (argc > 1 ? argc : 2); a++) {
		if (argc > 1 && strcmp(argv[a], "--spread") == 0 && a + 1 < argc) {
			spread = atoi(argv[++a]);
			continue;
		}
		const int nt = argc > 1 ? atoi(argv[a]) : 8;
		conga_pack::Packer pk(nt, spread);
		double best = 1e30;
		int w = 0;
		size_t ne = 0, nb = 0;
		for (int rep = 0; rep < 12; rep++) {
			const auto t0 = std::chrono::steady_clock::now();
			if (pk.start(pos.data(), off, 2, 0, out.data(), out.size()) != 0 || pk.finish(&w, &ne, &nb) != 0)
				return 1;
			best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
		}
		printf("%2d threads (spread %d): %.3f ms best of 12 (width %d, %zu exceptions, %zu bytes) = %.1f GB/s of positions read, %s\n", nt, spread, best, w, ne, nb,
				4.0 * n / best / 1e6, conga_pack::have_avx2_bmi2() ? "avx2+bmi2" : "scalar");
	}
	return 0;
}
