// pure compute rate of the encoder: one thread, 64 K reads that stay in L2, no pool
#include "../conga_amd/csrc/pack_host.h"
#include <chrono>
#include <cstdio>
int main()
{
	const uint64_t n = 65536;
	std::vector<int32_t> pos(n + 16);
	uint64_t x = 88172645463325252ull;
	int32_t p = 0;
	for (uint64_t i = 0; i < n + 16; i++) { x ^= x << 13, x ^= x >> 7, x ^= x << 17; p += (int32_t) (x % 199u); pos[i] = p; }
	alignas(64) static uint8_t out[65536 * 2 + 256];
	std::vector<conga_pack::Exc> exc;
	for (int w : {8, 10}) {
		double best = 1e30;
		for (int rep = 0; rep < 200; rep++) {
			exc.clear();
			const auto t0 = std::chrono::steady_clock::now();
			for (int k = 0; k < 20; k++) {
				if (w == 8) conga_pack::encode_whole_groups<8>(pos.data(), 8, n, out, exc);
				else conga_pack::encode_whole_groups<10>(pos.data(), 8, n, out, exc);
			}
			best = std::min(best, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 20);
		}
		printf("width %2d: %.3f ns per read on one thread, data in cache (%.2f G reads/s); %zu exceptions\n", w, best / (n - 8), (n - 8) / best, exc.size());
	}
	return 0;
}
