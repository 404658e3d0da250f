// pack_host.h -- the PRODUCER's side of conga_sample_reads_packed, in C++ for the host (no device code, no HIP call).
//
// The seam is count_reads_bam (bam_data.c:192-221): a loop that has bam1_core_t.pos of every record in a register, one record
// after the other in position order.  A producer that subtracts as it goes hands the engine differences of W bits instead of
// 32-bit positions (include/conga_hip.h: the format; delta16.hip.h: what the engine does with it).  This file is that producer
// for positions that already lie in an array -- the CLI's host decoders and tuple containers leave them so, and bench.py's
// timed step starts from such an array --: a pool of threads of the packer's own encodes runs of 8 192 reads (eight
// differences are W whole bytes, so every run starts on a byte) and collects the exceptions run by run; the caller's thread
// goes on with something else between start() and finish().  The producer is bound by the host's memory, so what it does with
// memory is the design: a run is put together in a buffer of the thread's own and leaves the core with non-temporal stores (the
// pinned buffer's lines are never read, and are in memory -- not modified in some core's cache -- when the copy engine asks),
// a thread takes eight runs in a row and asks for its positions 8 KB ahead (bench.py's step with the producer inside:
// 0.99-1.03 ms -> 0.68-0.71 on one box, the same differences encoded beforehand 0.62; tools/packprobe_bench.sh).  Rounds 3's encoder was numpy in the test binding, outside
// anything timed ("what's weak" 3).
//
// Host-only on purpose: tests/test_pack_host.py builds it with g++ (also under -fsanitize=thread) without the HIP runtime.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <sys/syscall.h>
#include <unistd.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace conga_pack {

constexpr uint64_t kRun = 8192; // reads per unit of work (a multiple of 8)
// runs a worker takes at a time (measurement switch: CONGA_DEBUG=1 CONGA_PACK_BATCH=n)
inline size_t batch_runs()
{
	static const size_t n = [] {
		const char *d = getenv("CONGA_DEBUG"), *e = getenv("CONGA_PACK_BATCH");
		return (size_t) (d && atoi(d) != 0 && e ? std::max(1, atoi(e)) : 0); // 0: by the sample's size
	}();
	return n;
}

// bytes `out` must hold for any width and up to max_esc exceptions: differences, padding, the two lists, the 64 bytes of slack the
// expansion's last 16-byte load may touch
inline size_t bound(uint64_t n_reads, size_t max_esc)
{
	return (size_t) ((n_reads + 7) / 8 * 16 + 16 + 8 * (uint64_t) max_esc + 64);
}

struct Exc {
	uint32_t index;
	int32_t pos;
};

// the group's eight values -> its W bytes: difference k occupies bits [k * W, (k + 1) * W)
template <int W> inline void store_group(const uint32_t v[8], uint8_t *dst)
{
	const uint64_t lo = (uint64_t) v[0] | (uint64_t) v[1] << W | (uint64_t) v[2] << (2 * W) | (uint64_t) v[3] << (3 * W);
	const uint64_t hi = (uint64_t) v[4] | (uint64_t) v[5] << W | (uint64_t) v[6] << (2 * W) | (uint64_t) v[7] << (3 * W);
	uint64_t w[2];
	if (W == 16) {
		w[0] = lo;
		w[1] = hi;
	} else {
		w[0] = lo | hi << (4 * W);
		w[1] = hi >> (64 - 4 * W);
	}
	memcpy(dst, w, W);
}

// whole groups [i0, i1) (multiples of 8, i0 > 0) without a chromosome border among them: every read has its predecessor
template <int W> inline void encode_groups(const int32_t *pos, uint64_t i0, uint64_t i1, uint8_t *out, std::vector<Exc> &exc)
{
	constexpr uint32_t kTop = (1u << W) - 1u;
	for (uint64_t i = i0; i < i1; i += 8) {
		uint32_t v[8], any = 0;
#pragma GCC unroll 8
		for (int k = 0; k < 8; k++) {
			const uint32_t d = (uint32_t) pos[i + k] - (uint32_t) pos[i + k - 1]; // (a position in front of its predecessor: huge)
			v[k] = d < kTop ? d : kTop;
			any |= d >= kTop ? 1u : 0u;
		}
		if (any)
			for (int k = 0; k < 8; k++)
				if (v[k] == kTop)
					exc.push_back(Exc{(uint32_t) (i + (uint64_t) k), pos[i + (uint64_t) k]});
		store_group<W>(v, out + (i >> 3) * W);
	}
}

#if defined(__x86_64__)
// The same with AVX2 + BMI2 (every x86 host an MI355X sits in has both; taken when the CPU says so): eight differences in one
// subtract, the clamp to all-ones a `min`, the exceptions a compare + movemask that is zero for all but one group in a few
// hundred, and the bits squeezed together by two `pext` -- a dozen instructions per eight reads where the scalar loop has sixty.
// What is left is the read of the positions themselves: 4 bytes per read from host memory.
// how far in front of the load the positions are asked for, in reads (CONGA_DEBUG=1 CONGA_PACK_PREFETCH=n; 0: no prefetch).  A core
// of a two-socket host waits ~120 ns for a line and has a few dozen lines in flight: one sequential stream per thread gets 8-10 GB/s
// from the hardware's own prefetcher, which also stops at every 4 KB page; asking 8 KB ahead keeps more lines under way
// (tools/packprobe.sh, three samples in turn on a GPU box: 0.84-0.90 ms against 0.95-1.17 without; 256 ... 4096 reads ahead tried).
inline int prefetch_reads()
{
	static const int n = [] {
		const char *d = getenv("CONGA_DEBUG"), *e = getenv("CONGA_PACK_PREFETCH");
		return d && atoi(d) != 0 && e ? std::max(0, atoi(e)) : 2048;
	}();
	return n;
}

template <int W> __attribute__((target("avx2,bmi2"))) inline void encode_group_avx2(const int32_t *pos, uint64_t i, uint8_t *out, std::vector<Exc> &exc)
{
	constexpr uint32_t kTop = (1u << W) - 1u;
	constexpr uint64_t kMask = (uint64_t) kTop * 0x0001000100010001ull;
	const __m256i top = _mm256_set1_epi32((int) kTop);
	const __m256i cur = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(pos + i));
	const __m256i prev = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(pos + i - 1));
	const __m256i v = _mm256_min_epu32(_mm256_sub_epi32(cur, prev), top); // (a position in front of its predecessor: huge, clamped)
	const int esc = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(v, top)));
	if (__builtin_expect(esc != 0, 0))
		for (int k = 0; k < 8; k++)
			if (esc >> k & 1)
				exc.push_back(Exc{(uint32_t) (i + (uint64_t) k), pos[i + (uint64_t) k]});
	const __m256i p16 = _mm256_packus_epi32(v, v); // 16 bits each: v0..v3 in the low lane's first quadword, v4..v7 in the high lane's
	const uint64_t lo = _pext_u64((uint64_t) _mm256_extract_epi64(p16, 0), kMask), hi = _pext_u64((uint64_t) _mm256_extract_epi64(p16, 2), kMask);
	uint64_t w[2];
	if (W == 16) {
		w[0] = lo;
		w[1] = hi;
	} else {
		w[0] = lo | hi << (4 * W);
		w[1] = hi >> (64 - 4 * W);
	}
	memcpy(out + (i >> 3) * W, w, W);
}

template <int W> __attribute__((target("avx2,bmi2"))) inline void encode_groups_avx2(const int32_t *pos, uint64_t i0, uint64_t i1, uint8_t *out, std::vector<Exc> &exc)
{
	const uint64_t ahead = (uint64_t) prefetch_reads();
	uint64_t i = i0;
	if (ahead)
		for (; i + 16 <= i1; i += 16) { // sixteen reads are one line of positions
			_mm_prefetch(reinterpret_cast<const char *>(pos + i + ahead), _MM_HINT_T0);
			encode_group_avx2<W>(pos, i, out, exc);
			encode_group_avx2<W>(pos, i + 8, out, exc);
		}
	for (; i < i1; i += 8)
		encode_group_avx2<W>(pos, i, out, exc);
}

inline bool have_avx2_bmi2()
{
	static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && !(getenv("CONGA_DEBUG") && getenv("CONGA_PACK_SCALAR")); // (measurement switch, read once: CONGA_DEBUG=1 CONGA_PACK_SCALAR=1)
	return yes;
}
#endif

template <int W> inline void encode_whole_groups(const int32_t *pos, uint64_t i0, uint64_t i1, uint8_t *out, std::vector<Exc> &exc)
{
#if defined(__x86_64__)
	if (have_avx2_bmi2())
		return encode_groups_avx2<W>(pos, i0, i1, out, exc);
#endif
	encode_groups<W>(pos, i0, i1, out, exc);
}

// the group that starts at i (a multiple of 8) read by read: chromosome borders among its reads (`forced`, ascending, consumed
// through *f), the sample's first read, the sample's end (`n`) inside it.  `pos` is the array read j is read from as pos[j]; with one
// array per chromosome (`fbase` != nullptr: fbase[k] is that array for the chromosome whose first read is forced[k], shifted so that the
// sample's index applies) it moves on as the borders are consumed -- a read is only ever compared with a predecessor of its own chromosome
template <int W> inline void encode_group_slow(const int32_t *&pos, uint64_t i, uint64_t n, const uint64_t *forced, const int32_t *const *fbase, size_t n_forced,
		size_t *f, uint8_t *out, std::vector<Exc> &exc)
{
	constexpr uint32_t kTop = (1u << W) - 1u;
	uint32_t v[8];
	for (int k = 0; k < 8; k++) {
		const uint64_t j = i + (uint64_t) k;
		if (j >= n) {
			v[k] = 0;
			continue;
		}
		bool is_first = j == 0;
		if (*f < n_forced && forced[*f] == j) {
			is_first = true;
			if (fbase)
				pos = fbase[*f];
			(*f)++;
		}
		const uint32_t d = is_first ? kTop : (uint32_t) pos[j] - (uint32_t) pos[j - 1];
		v[k] = d < kTop ? d : kTop;
		if (d >= kTop)
			exc.push_back(Exc{(uint32_t) j, pos[j]});
	}
	store_group<W>(v, out + (i >> 3) * W);
}

// one run [r0, r1) of the sample (r0 a multiple of 8; r1 one too, or the sample's end) at width W; `forced` = indices in [r0, r1)
// that are a chromosome's first read (ascending); `pos`: the array of read r0 (see encode_group_slow)
template <int W>
inline void encode_run_to(const int32_t *pos, uint64_t r0, uint64_t r1, const uint64_t *forced, const int32_t *const *fbase, size_t n_forced, uint8_t *out,
		std::vector<Exc> &exc)
{
	size_t f = 0;
	uint64_t cur = r0;
	if (cur == 0 && r1 > 0) { // (the sample's first read has no predecessor)
		encode_group_slow<W>(pos, 0, r1, forced, fbase, n_forced, &f, out, exc);
		cur = 8;
	}
	while (f < n_forced) {
		const uint64_t g = forced[f] & ~(uint64_t) 7;
		if (g > cur)
			encode_whole_groups<W>(pos, cur, g, out, exc);
		encode_group_slow<W>(pos, g, r1, forced, fbase, n_forced, &f, out, exc);
		cur = g + 8;
	}
	const uint64_t whole_end = r1 & ~(uint64_t) 7;
	if (whole_end > cur) {
		encode_whole_groups<W>(pos, cur, whole_end, out, exc);
		cur = whole_end;
	}
	if (cur < r1)
		encode_group_slow<W>(pos, cur, r1, forced, fbase, n_forced, &f, out, exc);
}

#if defined(__x86_64__)
// The run's bytes leave the core with non-temporal stores (CONGA_DEBUG=1 CONGA_PACK_NO_STREAM=1: plain ones): `out` is the pinned
// buffer the link reads next and nobody else -- written the plain way every line of it is first READ from memory (a quarter more
// traffic for a producer that is bound by memory) and then sits modified in some core's cache when the copy engine asks for it
// (engine_bgzf: the pinned ring goes up at 50 GB/s filled this way, at 43-44 filled by plain stores).
inline bool stream_out()
{
	static const bool yes = !(getenv("CONGA_DEBUG") && atoi(getenv("CONGA_DEBUG")) != 0 && getenv("CONGA_PACK_NO_STREAM"));
	return yes;
}
inline void stream_copy(uint8_t *dst, const uint8_t *src, size_t n) // src 16-byte aligned
{
	size_t k = 0;
	if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)
		for (; k + 16 <= n; k += 16)
			_mm_stream_si128(reinterpret_cast<__m128i *>(dst + k), _mm_load_si128(reinterpret_cast<const __m128i *>(src + k)));
	memcpy(dst + k, src + k, n - k);
}
#endif

template <int W>
inline void encode_run(const int32_t *pos, uint64_t r0, uint64_t r1, const uint64_t *forced, const int32_t *const *fbase, size_t n_forced, uint8_t *out,
		std::vector<Exc> &exc)
{
#if defined(__x86_64__)
	if (stream_out() && r1 - r0 <= kRun) {
		// the run is put together in a buffer of the thread's own (10 KB at ten bits: it stays in L1) ...
		alignas(64) uint8_t own[kRun / 8 * 16 + 64];
		const size_t at = (size_t) (r0 >> 3) * W, n_bytes = (size_t) ((r1 - r0 + 7) / 8) * W;
		// (the encoder addresses group g at base + g * W: a base computed as an integer, only the run's own bytes are written through it)
		uint8_t *base = reinterpret_cast<uint8_t *>(reinterpret_cast<uintptr_t>(own) - (uintptr_t) at);
		encode_run_to<W>(pos, r0, r1, forced, fbase, n_forced, base, exc);
		stream_copy(out + at, own, n_bytes); // ... and leaves it in one piece (run r begins r * 1024 * W bytes in: a multiple of 16)
		return;
	}
#endif
	encode_run_to<W>(pos, r0, r1, forced, fbase, n_forced, out, exc);
}

inline void encode_run_any(int width, const int32_t *pos, uint64_t r0, uint64_t r1, const uint64_t *forced, const int32_t *const *fbase, size_t n_forced,
		uint8_t *out, std::vector<Exc> &exc)
{
	switch (width) {
	case 4: return encode_run<4>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 5: return encode_run<5>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 6: return encode_run<6>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 7: return encode_run<7>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 8: return encode_run<8>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 9: return encode_run<9>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 10: return encode_run<10>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 11: return encode_run<11>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 12: return encode_run<12>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 13: return encode_run<13>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 14: return encode_run<14>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	case 15: return encode_run<15>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	default: return encode_run<16>(pos, r0, r1, forced, fbase, n_forced, out, exc);
	}
}

// The producer's rule for the width (conga_hip.h): the fewest bytes -- differences + 8 per exception -- among the widths that keep
// exceptions at or below one read in a thousand; taken from every 61st run of 256 reads (a 1x genome: 420 000 of 25.6 M
// differences looked at, 0.03 ms).  Deterministic: the same positions give the same width.
// (at(i): the position of the sample's read i; asked for in ascending order of i)
template <class At> inline int choose_width_at(At at, uint64_t n)
{
	if (n < 2)
		return 16;
	uint64_t need[18] = {0}; // need[b]: sampled differences that want exactly b bits (17: negative or >= 2^16 - 1)
	uint64_t seen = 0;
	const uint64_t stride = n > ((uint64_t) 1 << 18) ? 61 * 256 : 256;
	for (uint64_t a = 1; a < n; a += stride) {
		uint32_t before = (uint32_t) at(a - 1);
		for (uint64_t i = a; i < std::min(n, a + 256); i++) {
			const uint32_t here = (uint32_t) at(i);
			const uint32_t d = here - before;
			before = here;
			// the smallest W with d < 2^W - 1
			int b = 17;
			if (d < 0xFFFFu)
				b = d == 0 ? 1 : 32 - __builtin_clz(d + 1u);
			need[b]++;
			seen++;
		}
	}
	int best = 16;
	double best_cost = 1e300;
	uint64_t over = need[17];
	for (int w = 16; w >= 4; w--) { // over = sampled differences that do not fit w bits
		const double frac = (double) over / (double) std::max<uint64_t>(seen, 1);
		const double cost = (double) w / 8.0 + 8.0 * frac;
		if ((frac <= 1e-3 || w == 16) && cost < best_cost) {
			best_cost = cost;
			best = w;
		}
		over += need[w];
	}
	return best;
}

inline int choose_width(const int32_t *pos, uint64_t n)
{
	return choose_width_at([pos](uint64_t i) { return pos[i]; }, n);
}

// ... of a sample whose positions lie in one array per chromosome (the same reads give the same width as in one array)
inline int choose_width(const int32_t *const *chrom_pos, const uint64_t *chrom_off, int n_chrom)
{
	int c = 0;
	return choose_width_at([&](uint64_t i) {
		while (c + 1 < n_chrom && i >= chrom_off[c + 1])
			c++;
		return chrom_pos[c][i - chrom_off[c]];
	}, chrom_off[n_chrom]);
}

// A pool of threads that encodes one sample at a time.  start() returns at once; finish() waits and puts the exception lists behind
// the differences (the one-copy layout of conga_sample_reads_packed with esc_index == NULL).
inline void cpu_relax()
{
#if defined(__x86_64__)
	_mm_pause();
#else
	std::this_thread::yield();
#endif
}

// how long a worker looks for the next sample before it sleeps, in pauses (measurement switch: CONGA_DEBUG=1 CONGA_PACK_SPIN=n)
inline int spin_limit()
{
	static const int n = [] {
		const char *d = getenv("CONGA_DEBUG"), *e = getenv("CONGA_PACK_SPIN");
		return d && atoi(d) != 0 && e ? atoi(e) : 20000;
	}();
	return n;
}

#if defined(__linux__)
// The memory node a page lies on (move_pages in its query form: no page moves), -1: not known (not resident yet, no such call here).
inline int node_of(const void *p)
{
	void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t) 4095);
	int status = -1;
	const long rc = syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0);
	return rc == 0 && status >= 0 ? status : -1;
}
// the CPUs of a memory node (/sys/devices/system/node/nodeN/cpulist: "0-63,128-191")
inline bool node_cpus(int node, cpu_set_t *set)
{
	char path[96], text[4096];
	snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
	FILE *f = fopen(path, "r");
	if (!f)
		return false;
	const size_t n = fread(text, 1, sizeof text - 1, f);
	fclose(f);
	text[n] = 0;
	CPU_ZERO(set);
	int any = 0;
	for (const char *q = text; *q;) {
		char *end;
		const long a = strtol(q, &end, 10);
		if (end == q)
			break;
		long b = a;
		if (*end == '-')
			b = strtol(end + 1, &end, 10);
		for (long c = a; c <= b && c < CPU_SETSIZE; c++) {
			CPU_SET((int) c, set);
			any = 1;
		}
		q = *end == ',' ? end + 1 : end;
		if (*end != ',')
			break;
	}
	return any != 0;
}
#endif

class Packer {
public:
	// spread > 0: worker t pins itself to CPU (t * spread) mod the CPUs of the machine -- one worker per L3 domain reads through
	// that domain's own link to memory instead of eight sharing one (tools/packbench --spread)
	explicit Packer(int n_threads, int spread = 0) : n_threads_(std::max(1, n_threads))
	{
#if defined(__linux__)
		have_allowed_ = sched_getaffinity(0, sizeof allowed_, &allowed_) == 0;
		follow_node_ = spread == 0 && !(getenv("CONGA_DEBUG") && getenv("CONGA_PACK_NO_NUMA")); // (measurement switch)
#endif
		for (int t = 0; t < n_threads_; t++)
			threads_.emplace_back([this, t, spread] {
#if defined(__linux__)
				if (spread > 0) {
					const long n_cpu = sysconf(_SC_NPROCESSORS_CONF);
					cpu_set_t set;
					CPU_ZERO(&set);
					CPU_SET((int) (((long) t * spread) % std::max(1L, n_cpu)), &set);
					(void) pthread_setaffinity_np(pthread_self(), sizeof set, &set); // (refused: the worker runs where the scheduler puts it)
				}
#endif
				loop();
			});
	}
	~Packer()
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			quit_ = true;
			quit_hint_.store(true, std::memory_order_relaxed);
		}
		cv_.notify_all();
		for (std::thread &t : threads_)
			t.join();
	}
	int threads() const { return n_threads_; }

	// -> 0, or -1: arguments, -4: out_cap cannot hold the differences
	int start(const int32_t *pos, const uint64_t *chrom_off, int n_chrom, int width, uint8_t *out, size_t out_cap)
	{
		return start_any(pos, nullptr, chrom_off, n_chrom, width, out, out_cap);
	}
	// the same for positions that lie in one array per chromosome: chrom_pos[c][0 .. chrom_off[c + 1] - chrom_off[c]) (what a decoder
	// that works chromosome by chromosome leaves; chrom_pos[c] of a chromosome without reads is not looked at)
	int start_v(const int32_t *const *chrom_pos, const uint64_t *chrom_off, int n_chrom, int width, uint8_t *out, size_t out_cap)
	{
		if (!chrom_pos && n_chrom > 0)
			return -1;
		return start_any(nullptr, chrom_pos, chrom_off, n_chrom, width, out, out_cap);
	}

private:
	int start_any(const int32_t *pos, const int32_t *const *chrom_pos, const uint64_t *chrom_off, int n_chrom, int width, uint8_t *out, size_t out_cap)
	{
		if (!chrom_off || n_chrom < 0 || chrom_off[0] != 0 || !out || busy_)
			return -1;
		for (int c = 0; c < n_chrom; c++)
			if (chrom_off[c + 1] < chrom_off[c] || (chrom_pos && chrom_off[c + 1] > chrom_off[c] && !chrom_pos[c]))
				return -1;
		const uint64_t n = chrom_off[n_chrom];
		if ((n && !pos && !chrom_pos) || n >= 0xFFFFFFF0ull || (width != 0 && (width < 4 || width > 16)))
			return -1;
		if (width == 0)
			width = chrom_pos ? choose_width(chrom_pos, chrom_off, n_chrom) : choose_width(pos, n);
		const size_t d_bytes = (size_t) ((n + 7) / 8) * (size_t) width;
		if (((d_bytes + 15) & ~(size_t) 15) + 64 > out_cap)
			return -4;
		std::vector<uint64_t> forced;
		std::vector<const int32_t *> fbase;
		for (int c = 0; c < n_chrom; c++)
			if (chrom_off[c + 1] > chrom_off[c]) {
				forced.push_back(chrom_off[c]);
				// the chromosome's array as the sample's index sees it: element chrom_off[c] is its first read (an address computed as an
				// integer; only elements of the chromosome's own index range are ever read through it)
				fbase.push_back(chrom_pos ? reinterpret_cast<const int32_t *>(reinterpret_cast<uintptr_t>(chrom_pos[c]) - (uintptr_t) (4 * chrom_off[c])) : pos);
			}
#if defined(__linux__)
		// The workers stay on the memory node the positions lie on: a process that may run anywhere on a two-socket host (a GPU box
		// grants 16 CPUs' worth of time on all 256) otherwise has some of them read across the sockets' link -- the same encode,
		// fourteen threads: 0.92-0.95 ms inside one node, 0.92-1.30 anywhere (tools/packbench under taskset, profiles/r04k_packbench_numa.log).
		if (follow_node_ && have_allowed_ && n) {
			const int32_t *first = pos; // the sample's first read
			for (int c = 0; chrom_pos && c < n_chrom && first == nullptr; c++)
				if (chrom_off[c + 1] > chrom_off[c])
					first = chrom_pos[c];
			int node = first ? node_of(first) : -1;
			if (node < 0)
				node = node_of(out);
			if (node >= 0)
				want_node_.store(node, std::memory_order_relaxed);
		}
#endif
		{
			// (a thread that slept through the sample before wakes up whenever it likes, finds nothing left and goes back to sleep: it
			// reads these fields meanwhile -- they change under the lock, with no thread inside its loop)
			std::unique_lock<std::mutex> lk(mu_);
			done_cv_.wait(lk, [&] { return active_ == 0; });
			gather_ = chrom_pos != nullptr;
			n_ = n;
			width_ = width;
			out_ = out;
			out_cap_ = out_cap;
			forced_.swap(forced);
			fbase_.swap(fbase);
			n_runs_ = (size_t) ((n + kRun - 1) / kRun);
			if (exc_.size() < n_runs_)
				exc_.resize(n_runs_);
			busy_ = true;
			next_.store(0, std::memory_order_relaxed);
			left_ = n_runs_;
			generation_++;
			generation_hint_.store(generation_, std::memory_order_release);
		}
		cv_.notify_all();
		return 0;
	}

public:
	// -> 0, -1: nothing was started, -4: the exceptions do not fit behind the differences in out_cap
	int finish(int *width, size_t *n_esc, size_t *out_bytes)
	{
		if (!busy_)
			return -1;
		{
			std::unique_lock<std::mutex> lk(mu_);
			done_cv_.wait(lk, [&] { return left_ == 0 && active_ == 0; });
		}
		busy_ = false;
		size_t k = 0;
		for (size_t r = 0; r < n_runs_; r++)
			k += exc_[r].size();
		const size_t d_bytes = (size_t) ((n_ + 7) / 8) * (size_t) width_;
		const size_t esc_at = (d_bytes + 15) & ~(size_t) 15;
		if (width)
			*width = width_;
		if (n_esc)
			*n_esc = k;
		if (out_bytes)
			*out_bytes = esc_at + 8 * k;
		if (esc_at + 8 * k + 64 > out_cap_)
			return -4;
		memset(out_ + d_bytes, 0, esc_at - d_bytes);
		uint32_t *ei = reinterpret_cast<uint32_t *>(out_ + esc_at);
		int32_t *ep = reinterpret_cast<int32_t *>(out_ + esc_at) + k;
		size_t at = 0;
		for (size_t r = 0; r < n_runs_; r++)
			for (const Exc &e : exc_[r]) {
				ei[at] = e.index;
				ep[at] = e.pos;
				at++;
			}
		return 0;
	}

private:
	void loop()
	{
		uint64_t seen = 0;
		int my_node = -1;
		for (;;) {
#if defined(__linux__)
			{
				const int want = want_node_.load(std::memory_order_relaxed);
				if (want != my_node && want >= 0) { // (once per packer, as a rule)
					cpu_set_t on_node, mine;
					if (node_cpus(want, &on_node)) {
						CPU_AND(&mine, &on_node, &allowed_);
						if (CPU_COUNT(&mine) > 0)
							(void) pthread_setaffinity_np(pthread_self(), sizeof mine, &mine);
					}
					my_node = want;
				}
			}
#endif
			// (a caller that packs sample after sample -- a cohort, bench.py's step -- starts the next one within a millisecond: a
			// thread that goes to sleep at once pays a wake-up, and a core that idled its clock ramp, on every sample.  Look for the
			// next sample for that long before sleeping.)
			for (int spin = 0; spin < spin_limit() && generation_hint_.load(std::memory_order_acquire) == seen && !quit_hint_.load(std::memory_order_relaxed); spin++)
				cpu_relax();
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
				if (quit_)
					return;
				seen = generation_;
				active_++; // (finish() waits for every thread that took this sample up: none is inside the loop below when the next one starts)
			}
			size_t mine = 0;
			for (;;) {
				// (eight runs in a row per turn: a thread reads 256 KB of positions as one stream -- the hardware's prefetcher and the
				// requests made ahead in encode_groups_avx2 both start anew wherever a thread jumps)
				// (... as long as every thread still gets eight turns or so: a rank's share of a genome on eight GPUs is 390 runs)
				const size_t kBatch = batch_runs() ? batch_runs() : std::max<size_t>(1, std::min<size_t>(8, n_runs_ / (8 * (size_t) n_threads_)));
				const size_t b = next_.fetch_add(1, std::memory_order_relaxed);
				if (b * kBatch >= n_runs_)
					break;
				for (size_t r = b * kBatch; r < std::min(n_runs_, (b + 1) * kBatch); r++) {
					const uint64_t r0 = (uint64_t) r * kRun, r1 = std::min(n_, r0 + kRun);
					const auto f0 = std::lower_bound(forced_.begin(), forced_.end(), r0), f1 = std::lower_bound(f0, forced_.end(), r1);
					const size_t k0 = (size_t) (f0 - forced_.begin());
					// the array read r0 lies in: that of the last chromosome that begins at or before it (forced_[0] == 0 whenever there are reads)
					const int32_t *at_r0 = fbase_[(f0 != forced_.end() && *f0 == r0) ? k0 : k0 - 1];
					exc_[r].clear();
					encode_run_any(width_, at_r0, r0, r1, forced_.data() + k0, gather_ ? fbase_.data() + k0 : nullptr, (size_t) (f1 - f0), out_, exc_[r]);
					mine++;
				}
			}
#if defined(__x86_64__)
			_mm_sfence(); // the non-temporal stores of this thread's runs are in memory before finish() says so
#endif
			{
				std::lock_guard<std::mutex> g(mu_);
				left_ -= mine;
				active_--;
				if (active_ == 0)
					done_cv_.notify_all();
			}
		}
	}

	const int n_threads_;
#if defined(__linux__)
	cpu_set_t allowed_;
	bool have_allowed_ = false, follow_node_ = false;
#endif
	std::atomic<int> want_node_{-1}; // the memory node the workers keep to (-1: wherever the scheduler puts them)
	std::vector<std::thread> threads_;
	std::mutex mu_;
	std::condition_variable cv_, done_cv_;
	bool quit_ = false, busy_ = false;
	uint64_t generation_ = 0;
	std::atomic<uint64_t> generation_hint_{0}; // generation_, readable without the lock (the workers' spin)
	std::atomic<bool> quit_hint_{false};
	size_t left_ = 0, n_runs_ = 0;
	int active_ = 0;
	std::atomic<size_t> next_{0};
	bool gather_ = false;
	uint64_t n_ = 0;
	int width_ = 16;
	uint8_t *out_ = nullptr;
	size_t out_cap_ = 0;
	std::vector<uint64_t> forced_;        // first read of every chromosome that has reads
	std::vector<const int32_t *> fbase_;  // ... and the array its reads are read from, indexed by the sample's read index
	std::vector<std::vector<Exc>> exc_;
};

} // namespace conga_pack
