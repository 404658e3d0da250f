// packbench.cpp -- the host producer of the packed hand-over by itself (conga_amd/csrc/pack_host.h): how long a pool of T threads
// takes to turn a 1x genome's 25.6 M positions into 10-bit differences, from ordinary memory.  What bench.py's
// `hand_over.packed_encode_timed` can reach is this or the link, whichever is slower.
//   g++ -O3 -std=c++17 -o tools/packbench tools/packbench.cpp -lpthread ; tools/packbench [threads ...]
#include "../conga_amd/csrc/pack_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>

int main(int argc, char **argv)
{
	// --rotate R: R samples of 25.6 M positions taken in turn, as bench.py's step does (three: 307 MB between two encodes of one
	// sample, more than a socket's L3 -- one sample encoded again and again is partly read from cache)
	const uint64_t n = 25600000;
	int rotate = 1;
	for (int a = 1; a + 1 < argc; a++)
		if (strcmp(argv[a], "--rotate") == 0)
			rotate = std::max(1, atoi(argv[a + 1]));
	// --huge 1 / 0: the positions in memory the kernel is asked to back with 2 MB pages / with 4 KB pages (madvise); default: as malloc gives it
	int huge = -1;
	for (int a = 1; a + 1 < argc; a++)
		if (strcmp(argv[a], "--huge") == 0)
			huge = atoi(argv[a + 1]);
	std::vector<int32_t *> samples;
	for (int r = 0; r < rotate; r++) {
		const size_t bytes = ((n * 4 + (2u << 20) - 1) / (2u << 20) + 1) * (2u << 20);
		char *m = (char *) mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
		if (m == MAP_FAILED)
			return 2;
		char *al = (char *) (((uintptr_t) m + (2u << 20) - 1) & ~(uintptr_t) ((2u << 20) - 1));
		if (huge >= 0)
			madvise(al, bytes - (2u << 20), huge ? MADV_HUGEPAGE : MADV_NOHUGEPAGE);
		samples.push_back((int32_t *) al);
	}
	uint64_t x = 88172645463325252ull;
	for (int32_t *pos : samples) {
		int32_t p = 0;
		for (uint64_t i = 0; i < n; i++) {
			x ^= x << 13, x ^= x >> 7, x ^= x << 17;
			if (i == n / 2)
				p = 0;
			p += (int32_t) (x % 199u) + ((x >> 40) % 4000u == 0 ? 30000 : 0);
			pos[i] = p;
		}
	}
	const uint64_t off[3] = {0, n / 2, n};
	std::vector<std::vector<uint8_t>> outs((size_t) rotate, std::vector<uint8_t>(conga_pack::bound(n, n / 16)));
	int spread = 0;
	bool any = false;
	for (int a = 1; a < argc || !any; a++) {
		if (a < argc && strcmp(argv[a], "--spread") == 0 && a + 1 < argc) {
			spread = atoi(argv[++a]);
			continue;
		}
		if (a < argc && (strcmp(argv[a], "--rotate") == 0 || strcmp(argv[a], "--huge") == 0) && a + 1 < argc) {
			a++;
			continue;
		}
		any = true;
		const int nt = a < argc ? atoi(argv[a]) : 8;
		conga_pack::Packer pk(nt, spread);
		std::vector<double> ms;
		int w = 0;
		size_t ne = 0, nb = 0;
		const int reps = 12 * rotate;
		for (int rep = 0; rep < reps; rep++) {
			const auto t0 = std::chrono::steady_clock::now();
			std::vector<uint8_t> &out = outs[(size_t) (rep % rotate)];
			if (pk.start(samples[(size_t) (rep % rotate)], off, 2, 0, out.data(), out.size()) != 0 || pk.finish(&w, &ne, &nb) != 0)
				return 1;
			ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
		}
		std::sort(ms.begin() + rotate, ms.end()); // (the first turn touches the pages)
		const double best = ms[(size_t) rotate], median = ms[(size_t) rotate + (ms.size() - (size_t) rotate) / 2];
		printf("%2d threads (spread %d, rotate %d, huge %d): %.3f ms best, %.3f median of %d (width %d, %zu exceptions, %zu bytes) = %.1f GB/s of positions read, %s\n", nt, spread,
				rotate, huge, best, median, reps - rotate, w, ne, nb, 4.0 * n / best / 1e6, conga_pack::have_avx2_bmi2() ? "avx2+bmi2" : "scalar");
	}
	return 0;
}
