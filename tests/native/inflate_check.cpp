// inflate_check.cpp -- conga_host::inflate_raw against zlib on generated streams (built and run by tests/test_inflate.py).
//   inflate_check SEED N_STREAMS     -> "ok <streams> <bytes>" or a description of the first difference
// Streams: raw deflate made by zlib at every level and strategy (stored, fixed-Huffman, dynamic, RLE, huffman-only) from
// data of several kinds (random bytes, few symbols, long runs, text-like, BAM-like records), sizes 0 .. 65 280; then
// truncations and bit flips of valid streams, which must be refused or still pass zlib's own verdict, never crash.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../conga_amd/host/inflate_fast.h"

static uint64_t rng_state;
static uint64_t rnd()
{
	uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

static bool same(const uint8_t *a, const uint8_t *b, size_t n) { return n == 0 || memcmp(a, b, n) == 0; }

static std::vector<uint8_t> make_data(size_t n, int kind)
{
	std::vector<uint8_t> d(n);
	switch (kind) {
	case 0: for (auto &x : d) x = (uint8_t) rnd(); break;                            // incompressible
	case 1: for (auto &x : d) x = (uint8_t) "ACGT"[rnd() & 3]; break;                 // four symbols
	case 2: { uint8_t v = 0; size_t run = 0; for (auto &x : d) { if (!run) { v = (uint8_t) rnd(); run = 1 + rnd() % 600; } x = v; run--; } break; } // runs
	case 3: { const char *w[] = {"the ", "quick ", "brown ", "fox ", "jumps ", "over ", "lazy ", "dog\n"}; size_t i = 0; while (i < n) { const char *s = w[rnd() & 7]; while (*s && i < n) d[i++] = (uint8_t) *s++; } break; }
	case 4: { // BAM-like: a fixed core, a counter name, random packed bases and qualities
		size_t i = 0; uint32_t id = 0;
		while (i < n) {
			uint8_t rec[64]; memset(rec, 0, sizeof rec); uint32_t p = id * 113; memcpy(rec + 8, &p, 4); rec[12] = 12; rec[13] = 60;
			int len = snprintf((char *) rec + 36, 20, "r%010u", id++); (void) len;
			for (size_t k = 0; k < sizeof rec && i < n; k++) d[i++] = rec[k];
			for (int k = 0; k < 50 && i < n; k++) d[i++] = (uint8_t) ((1u << (rnd() & 3)) << 4 | (1u << (rnd() & 3)));
			for (int k = 0; k < 100 && i < n; k++) d[i++] = (uint8_t) (2 + rnd() % 39);
		}
		break; }
	default: if (n) memset(d.data(), 0, n); break;                                           // all zero
	}
	return d;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &d, int level, int strategy, int mem_level)
{
	z_stream zs; memset(&zs, 0, sizeof zs);
	if (deflateInit2(&zs, level, Z_DEFLATED, -15, mem_level, strategy) != Z_OK) abort();
	std::vector<uint8_t> out(deflateBound(&zs, (uLong) d.size()) + 64);
	zs.next_in = const_cast<Bytef *>(d.data()); zs.avail_in = (uInt) d.size();
	zs.next_out = out.data(); zs.avail_out = (uInt) out.size();
	// now and then a stream of several blocks: flush in the middle
	if (d.size() > 100 && (rnd() & 3) == 0) {
		zs.avail_in = (uInt) (d.size() / 2);
		deflate(&zs, (rnd() & 1) ? Z_FULL_FLUSH : Z_SYNC_FLUSH);
		zs.avail_in = (uInt) (d.size() - d.size() / 2);
	}
	if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
	out.resize(zs.total_out);
	deflateEnd(&zs);
	return out;
}

static bool zlib_inflate(const std::vector<uint8_t> &c, std::vector<uint8_t> &out)
{
	z_stream zs; memset(&zs, 0, sizeof zs);
	if (inflateInit2(&zs, -15) != Z_OK) abort();
	zs.next_in = const_cast<Bytef *>(c.data()); zs.avail_in = (uInt) c.size();
	uint8_t none[1];
	zs.next_out = out.empty() ? none : out.data(); zs.avail_out = (uInt) out.size(); // (zlib wants a pointer even for 0 bytes)
	const int rc = inflate(&zs, Z_FINISH);
	const bool ok = rc == Z_STREAM_END && zs.avail_out == 0;
	inflateEnd(&zs);
	return ok;
}

int main(int argc, char **argv)
{
	rng_state = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
	const long n_streams = argc > 2 ? atol(argv[2]) : 1000;
	unsigned long long bytes = 0;
	for (long s = 0; s < n_streams; s++) {
		static const size_t sizes[] = {0, 1, 2, 7, 8, 9, 63, 64, 257, 258, 259, 1000, 4096, 32768, 32769, 65279, 65280};
		const size_t n = (rnd() & 1) ? sizes[rnd() % (sizeof sizes / sizeof sizes[0])] : rnd() % 65281;
		const std::vector<uint8_t> d = make_data(n, (int) (rnd() % 6));
		static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
		const std::vector<uint8_t> c = deflate_raw(d, (int) (rnd() % 10), strategies[rnd() % 6], 1 + (int) (rnd() % 9));
		// guard bytes around the output: nothing outside [0, n) may be written
		std::vector<uint8_t> out(n + 32, 0xA5);
		if (!conga_host::inflate_raw(c.data(), c.size(), out.data() + 16, n) || !same(out.data() + 16, d.data(), n)) {
			printf("stream %ld: valid stream of %zu -> %zu bytes refused or wrong\n", s, c.size(), n);
			return 1;
		}
		for (int k = 0; k < 16; k++)
			if (out[k] != 0xA5 || out[16 + n + k] != 0xA5) {
				printf("stream %ld: wrote outside the output buffer\n", s);
				return 1;
			}
		bytes += n;
		// wrong output sizes are refused
		if (n > 0 && conga_host::inflate_raw(c.data(), c.size(), out.data() + 16, n - 1)) {
			printf("stream %ld: accepted with a short output buffer\n", s);
			return 1;
		}
		if (conga_host::inflate_raw(c.data(), c.size(), out.data() + 16, n + 1)) {
			printf("stream %ld: accepted with a long output buffer\n", s);
			return 1;
		}
		// damaged streams: same verdict as zlib whenever this decoder accepts (it may also refuse: the caller falls back)
		for (int t = 0; t < 6 && !c.empty(); t++) {
			std::vector<uint8_t> bad = c;
			if (t < 2)
				bad.resize(rnd() % c.size());
			else
				bad[rnd() % bad.size()] ^= (uint8_t) (1u << (rnd() & 7));
			std::vector<uint8_t> o1(n + 32, 0), o2(n);
			const bool a = conga_host::inflate_raw(bad.data(), bad.size(), o1.data() + 16, n);
			if (a) {
				const bool z = zlib_inflate(bad, o2);
				if (!z || !same(o1.data() + 16, o2.data(), n)) {
					printf("stream %ld: damaged stream (kind %d, %zu of %zu bytes, output %zu) accepted; zlib %s\n", s, t, bad.size(),
							c.size(), n, z ? "accepts it with other bytes" : "refuses it");
					return 1;
				}
			}
		}
	}
	printf("ok %ld %llu\n", n_streams, bytes);
	return 0;
}
