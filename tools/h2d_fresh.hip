// h2d_fresh.hip -- why does the BAM route's pinned ring go up at 38-42 GB/s when one large pinned copy goes up at 57?
// A ring of 12 x 8 MB pinned slots is filled by host threads from ordinary memory and every slot is sent up with hipMemcpyAsync as
// soon as it is full (a slot is reused when its copy's event has fired) -- the product's upload, without the file and the kernels:
//   cold    nobody writes: the slots go up again and again (what the copy engine does with 8 MB pieces of memory that lies in DRAM)
//   cached  the threads fill the slots with memcpy (ordinary stores: the lines are dirty in the cores' caches when the engine reads)
//   stream  the threads fill the slots with non-temporal stores (the lines go past the caches to DRAM)
//   kernels as `cached`, with every wave slot of the machine taken by a long launch on a low-priority stream (the inflate's stand-in)
//   workers as `cached`, but every filling thread waits for its slot's event itself (hipEventSynchronize in eight threads beside the
//           thread that sends: the product's structure)
//   pread   the threads fill the slots with pread() from a file in the page cache (the product's source)
//   pread+b / pread+c   pread() 256 KB at a time into a buffer of the thread's own, from there into the slot with non-temporal / ordinary stores
//   bounce  the threads copy 256 KB at a time into a buffer of their own with memcpy (stand-in for pread: the kernel's copy ends in
//           the core's cache) and from there into the slot with non-temporal stores
//   hipcc --offload-arch=gfx950 -O3 -mavx2 -o tools/h2d_fresh tools/h2d_fresh.hip -lpthread && tools/h2d_fresh [threads] [total MB]
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// a launch that keeps every wave slot of the machine busy for `ticks` of the 100 MHz clock (stands in for the inflate launches the
// upload shares the machine with)
__global__ __launch_bounds__(256) void busy_kernel(unsigned long long ticks, unsigned *sink)
{
	const unsigned long long t0 = wall_clock64();
	unsigned x = threadIdx.x;
	while (wall_clock64() - t0 < ticks)
		for (int k = 0; k < 256; k++)
			x = x * 1664525u + 1013904223u;
	if (x == 12345u)
		sink[0] = x;
}

static void copy_stream(uint8_t *dst, const uint8_t *src, size_t n) // n a multiple of 64, dst 32-byte aligned
{
	for (size_t i = 0; i < n; i += 64) {
		const __m256i a = _mm256_loadu_si256((const __m256i *) (src + i)), b = _mm256_loadu_si256((const __m256i *) (src + i + 32));
		_mm256_stream_si256((__m256i *) (dst + i), a);
		_mm256_stream_si256((__m256i *) (dst + i + 32), b);
	}
	_mm_sfence();
}

int main(int argc, char **argv)
{
	const int n_threads = argc > 1 ? atoi(argv[1]) : 8;
	const size_t total = (size_t) (argc > 2 ? atoi(argv[2]) : 1400) << 20;
	const size_t piece = (size_t) 8 << 20;
	const int n_slots = 12;
	const size_t n_pieces = total / piece;
	uint8_t *ring, *dev;
	CK(hipHostMalloc((void **) &ring, piece * n_slots, hipHostMallocDefault));
	CK(hipMalloc((void **) &dev, total));
	std::vector<uint8_t> src(total);
	for (size_t i = 0; i < total; i += 4096)
		src[i] = (uint8_t) (i >> 12);
	memset(ring, 1, piece * n_slots);
	// the same bytes as a file in the page cache (written, then read twice so that its pages are settled)
	char path[] = "/tmp/h2d_fresh_XXXXXX";
	const int fd = mkstemp(path);
	if (fd < 0 || write(fd, src.data(), total) != (ssize_t) total)
		return 2;
	unlink(path);
	for (int k = 0; k < 2; k++)
		for (size_t o = 0; o < total; o += piece)
			if (pread(fd, ring, piece, (off_t) o) != (ssize_t) piece)
				return 2;
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	std::vector<hipEvent_t> ev(n_slots);
	for (auto &e : ev)
		CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	const char *names[] = {"cold", "cached", "stream", "bounce", "kernels", "workers", "pread", "pread+b", "pread+c"};
	hipStream_t kst;
	int lo_prio = 0, hi_prio = 0;
	CK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
	CK(hipStreamCreateWithPriority(&kst, hipStreamNonBlocking, lo_prio));
	unsigned *sink;
	CK(hipMalloc((void **) &sink, 64));
	for (int rep = 0; rep < 2; rep++)
		for (int mode = 0; mode < 9; mode++) {
			if (mode == 4) // the "cached" fill with every wave slot of the machine taken by a long launch on a low-priority stream
				hipLaunchKernelGGL(busy_kernel, dim3(2048), dim3(256), 0, kst, 6000000ull /* 60 ms */, sink);
			std::mutex mu;
			std::condition_variable cv;
			std::vector<int> filled(n_pieces, 0);
			std::vector<char> busy(n_slots, 0);
			std::atomic<size_t> next{0}, issued{0};
			std::atomic<long long> ns_copy{0}, ns_wait{0};
			const auto t0 = std::chrono::steady_clock::now();
			auto worker = [&]() {
				std::vector<uint8_t> bounce((size_t) 256 << 10);
				for (;;) {
					const size_t c = next.fetch_add(1);
					if (c >= n_pieces)
						break;
					const int slot = (int) (c % n_slots);
					const auto w0 = std::chrono::steady_clock::now();
					if (c >= (size_t) n_slots) { // the slot's previous piece must have gone up
						std::unique_lock<std::mutex> l(mu);
						if (mode == 5) { // the product's way: wait until that piece's copy is ISSUED, then for its event -- in this thread
							cv.wait(l, [&] { return issued.load() > c - n_slots; });
							l.unlock();
							CK(hipEventSynchronize(ev[slot]));
						} else
							cv.wait(l, [&] { return filled[c - n_slots] == 2; });
					}
					const auto w1 = std::chrono::steady_clock::now();
					uint8_t *dst = ring + (size_t) slot * piece;
					const uint8_t *from = src.data() + c * piece;
					if (mode == 1 || mode == 4 || mode == 5)
						memcpy(dst, from, piece);
					else if (mode == 7 || mode == 8) { // pread into a buffer of the thread's own, from there into the slot (non-temporal / ordinary stores)
						for (size_t o = 0; o < piece; o += bounce.size()) {
							if (pread(fd, bounce.data(), bounce.size(), (off_t) (c * piece + o)) != (ssize_t) bounce.size())
								exit(3);
							if (mode == 7)
								copy_stream(dst + o, bounce.data(), bounce.size());
							else
								memcpy(dst + o, bounce.data(), bounce.size());
						}
					} else if (mode == 6) { // the product's source: the page cache, by way of the kernel's copy
						if (pread(fd, dst, piece, (off_t) (c * piece)) != (ssize_t) piece)
							exit(3);
					} else if (mode == 2)
						copy_stream(dst, from, piece);
					else if (mode == 3)
						for (size_t o = 0; o < piece; o += bounce.size()) {
							memcpy(bounce.data(), from + o, bounce.size());
							copy_stream(dst + o, bounce.data(), bounce.size());
						}
					const auto w2 = std::chrono::steady_clock::now();
					ns_wait += std::chrono::duration_cast<std::chrono::nanoseconds>(w1 - w0).count();
					ns_copy += std::chrono::duration_cast<std::chrono::nanoseconds>(w2 - w1).count();
					{
						std::lock_guard<std::mutex> l(mu);
						filled[c] = 1;
					}
					cv.notify_all();
				}
			};
			std::vector<std::thread> th;
			for (int t = 0; t < n_threads; t++)
				th.emplace_back(worker);
			// this thread sends the pieces up in order and frees their slots
			size_t sent = 0, freed = 0;
			while (freed < n_pieces) {
				if (sent < n_pieces) {
					bool ready;
					{
						std::unique_lock<std::mutex> l(mu);
						ready = filled[sent] == 1;
						if (!ready && freed == sent)
							cv.wait(l, [&] { return filled[sent] == 1; }), ready = true;
					}
					if (ready) {
						CK(hipMemcpyAsync(dev + sent * piece, ring + (sent % n_slots) * piece, piece, hipMemcpyHostToDevice, st));
						CK(hipEventRecord(ev[sent % n_slots], st));
						sent++;
						if (mode == 5) {
							{
								std::lock_guard<std::mutex> l(mu);
								issued = sent;
							}
							cv.notify_all();
						}
						continue;
					}
				}
				if (mode == 5 && sent < n_pieces) { // (the workers free the slots: this thread only sends)
					std::unique_lock<std::mutex> l(mu);
					cv.wait(l, [&] { return filled[sent] == 1; });
					continue;
				}
				CK(hipEventSynchronize(ev[freed % n_slots]));
				{
					std::lock_guard<std::mutex> l(mu);
					filled[freed] = 2;
				}
				cv.notify_all();
				freed++;
			}
			for (auto &t : th)
				t.join();
			CK(hipStreamSynchronize(st));
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			printf("%-7s %5.1f ms = %5.1f GB/s  (%d threads: %.1f ms filling, %.1f ms waiting for a slot, each)\n", names[mode], ms, total / ms / 1e6,
					n_threads, ns_copy / 1e6 / n_threads, ns_wait / 1e6 / n_threads);
			(void) busy;
			CK(hipStreamSynchronize(kst));
		}
	return 0;
}
