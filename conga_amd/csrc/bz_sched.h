// bz_sched.h -- the upload pipeline of conga_reads_bgzf*: jobs, tickets, who owns which buffer.  HOST code only.
//
// A cohort's BAM files are a pipeline two samples deep (the reference reads one sample per process,
// bam_data.c:253-339, and nothing of its results depends on when a sample's bytes were copied): the next samples' bytes are NAMED to
// the engine (conga_reads_bgzf_next_fd), an upload thread brings them up as jobs of their own through a ring of pinned pieces that
// host threads fill, reads their BGZF block table off the bytes on the way, and a thread per job launches their inflates AHEAD into a
// spare output set; the conga_reads_bgzf_fd call for a sample then finds its stream inflated.  What can go wrong here is ORDER:
//   * two device buffers carry the compressed bytes of any number of jobs: a buffer is its job's until nothing reads it any more;
//   * ONE spare output set is owned by one named job at a time and goes to the OLDEST ticket that waits for it (round 3's last day,
//     tests/soak.py --bam seed 81 case 38: it went to whoever woke first, the call in front waited for the job named first, that job
//     for the set, the set for the call behind -- a standstill), also to a job whose table the CALLER brought (such a
//     job never entered the waiting list and slept for ever);
//   * a job may be taken up by its call at any moment (before it started, while it goes up, when it is through), given up
//     (conga_reads_bgzf_forget, an error, the context's end), or fail on the way; none of that may leave a thread waiting.
// Everything the pipeline asks of the GPU goes through the small `Machine` interface below -- conga_api.hip implements it with
// HIP, tests/test_bz_sched.py with a fake in ordinary memory and runs this file under -fsanitize=thread, no GPU: round 3's
// VERDICT found that the scheduler (a context-wide and a per-job mutex taken nested, five kinds of helper thread) had no test
// that runs without a GPU and no sanitizer coverage.
#pragma once
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <errno.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include "../../include/conga_hip.h"

namespace bz {

// Measurement switch (CONGA_DEBUG=1 CONGA_BGZF_TRACE=1): what happened when, on stderr -- milliseconds of the steady clock (the last
// five digits), so that lines of several threads and of the caller's own clock can be laid side by side.
inline std::atomic<bool> &trace_on()
{
	static std::atomic<bool> on{false};
	return on;
}
inline void trace(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
inline void trace(const char *fmt, ...)
{
	if (!trace_on().load(std::memory_order_relaxed))
		return;
	char line[320];
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	int n = snprintf(line, sizeof line, "[bz %9.2f] ", ms - 100000.0 * (double) (long long) (ms / 100000.0));
	va_list ap;
	va_start(ap, fmt);
	n += vsnprintf(line + n, sizeof line - (size_t) n - 2, fmt, ap);
	va_end(ap);
	n = std::min<int>(n, (int) sizeof line - 2);
	line[n] = '\n';
	(void) !fwrite(line, 1, (size_t) n + 1, stderr);
}

// ---- what the pipeline asks of the machine.  Calls may come from the upload thread, its copying threads, a job's inflating
// thread or the caller's thread, as noted; an implementation keeps whatever per-thread binding its runtime wants in bind().
struct Machine {
	virtual ~Machine() {}
	virtual bool bind() = 0;                                     // any thread, first: make it ready to talk to the device
	virtual uint8_t *ring_slot(int slot) = 0;                    // pinned piece `slot` of the staging ring (host address)
	virtual bool slot_wait(int slot) = 0;                        // copying thread: the last copy out of the slot is through
	virtual uint8_t *up_buffer(int which, size_t bytes) = 0;     // upload thread: device buffer 0 / 1 for compressed bytes, grown; NULL: no memory
	virtual void *event_create() = 0;                            // upload thread: an event for one batch (NULL: failed)
	virtual void event_destroy(void *ev) = 0;                    // any thread (the job's end)
	virtual bool copy_up(uint8_t *dst, int slot, size_t len) = 0; // upload thread: slot -> dst on the copy stream, the slot busy until it is through
	virtual bool event_record(void *ev) = 0;                     // upload thread: on the copy stream, behind the copies so far
	// inflating ahead (a job's own thread)
	virtual bool ahead_possible() = 0;                           // upload thread: tables for the inflate are there, no split-read layout, the usual kernel
	virtual bool spare_reserve(size_t n_blocks, uint64_t out_bytes) = 0; // room in the spare output set (grown quietly)
	virtual bool ahead_launch(void *batch_event, const uint8_t *d_bytes, const conga_bgzf_block *blocks, const uint64_t *out_off, size_t first,
			size_t n, int launch) = 0;                            // blocks [first, first + n) of the job's table, behind the batch's event
	virtual bool ahead_mark() = 0;                               // behind the last launch: what the adopting call waits for on ITS stream
	virtual bool ahead_wait() = 0;                               // host wait for that mark
	virtual void ahead_drain() = 0;                              // everything launched ahead is through (a job given up)
	virtual void prewarm_join() = 0;                             // buffers being allocated on the side are there (or not)
};

// where the compressed bytes are: in the caller's memory, or in a file (read with pread: no mapping, no page faults)
struct ByteSource {
	const uint8_t *bytes = nullptr;
	int fd = -1;
	uint64_t file_off = 0;
	mutable std::atomic<bool> io_error{false}; // the last failed fetch was pread() failing (errno), not the file ending early
	ByteSource() = default;
	ByteSource(const ByteSource &o) : bytes(o.bytes), fd(o.fd), file_off(o.file_off), io_error(o.io_error.load()) {}
	ByteSource &operator=(const ByteSource &o)
	{
		bytes = o.bytes;
		fd = o.fd;
		file_off = o.file_off;
		io_error.store(o.io_error.load());
		return *this;
	}
	bool fetch(size_t at, void *dst, size_t n) const
	{
		if (bytes) {
			memcpy(dst, bytes + at, n);
			return true;
		}
		uint8_t *p = static_cast<uint8_t *>(dst);
		while (n) {
			const ssize_t got = pread(fd, p, n, (off_t) (file_off + at));
			if (got < 0 && (errno == EINTR || errno == EAGAIN))
				continue; // (a signal during the upload is not the file's end)
			if (got <= 0) {
				io_error = got < 0;
				return false;
			}
			p += got;
			at += (size_t) got;
			n -= (size_t) got;
		}
		return true;
	}
	// The same into a slot of the pinned ring (16-byte aligned): a file's bytes come 256 KB at a time into a buffer of the calling
	// thread's own -- the kernel's copy ends in the core's cache -- and go from there into the slot with non-temporal stores.
	// tools/h2d_fresh.hip: a ring filled by pread() itself goes up at 43-44 GB/s, filled this way at 50 (what it does filled from
	// ordinary memory), and the filling threads are through in two thirds of the time.
	bool fetch_into_ring(size_t at, uint8_t *dst, size_t n) const
	{
		if (bytes || ((uintptr_t) dst & 15u) != 0) // (memory of the caller's: one copy either way)
			return fetch(at, dst, n);
#if !defined(__x86_64__)
		return fetch(at, dst, n); // (the non-temporal stores below are SSE2; elsewhere pread() fills the slot itself)
#else
		constexpr size_t kBounce = (size_t) 256 << 10;
		static thread_local std::unique_ptr<uint8_t[]> bounce;
		if (!bounce)
			bounce.reset(new uint8_t[kBounce + 64]);
		uint8_t *b = (uint8_t *) (((uintptr_t) bounce.get() + 63u) & ~(uintptr_t) 63u);
		while (n) {
			const size_t want = std::min(n, kBounce);
			size_t have = 0;
			while (have < want) {
				const ssize_t got = pread(fd, b + have, want - have, (off_t) (file_off + at + have));
				if (got < 0 && (errno == EINTR || errno == EAGAIN))
					continue;
				if (got <= 0) {
					io_error = got < 0;
					return false;
				}
				have += (size_t) got;
			}
			size_t i = 0;
			for (; i + 64 <= want; i += 64) {
				const __m128i v0 = _mm_load_si128((const __m128i *) (b + i)), v1 = _mm_load_si128((const __m128i *) (b + i + 16));
				const __m128i v2 = _mm_load_si128((const __m128i *) (b + i + 32)), v3 = _mm_load_si128((const __m128i *) (b + i + 48));
				_mm_stream_si128((__m128i *) (dst + i), v0);
				_mm_stream_si128((__m128i *) (dst + i + 16), v1);
				_mm_stream_si128((__m128i *) (dst + i + 32), v2);
				_mm_stream_si128((__m128i *) (dst + i + 48), v3);
			}
			if (i < want)
				memcpy(dst + i, b + i, want - i);
			dst += want;
			at += want;
			n -= want;
		}
		_mm_sfence(); // (the slot is handed to the copy engine next)
		return true;
#endif
	}
};

// the pipeline's sizes and switches (conga_api.hip fills them from the engine's one table of knobs; the test picks its own)
struct Config {
	size_t slot_bytes = (size_t) 8 << 20; // a slot of the ring; pieces are at most a slot
	int n_slots = 12;
	size_t piece = 0;                     // 0: a slot; tests: small pieces, so that a small file goes through every part of this
	int pieces_per_launch_small = 16;     // a batch of small pieces
	int copy_threads = 0;                 // 0: half of the cores the process may use, at most the ring's slots
	unsigned cpus = 8;                    // cores the process may use
	bool plain_pread = false;             // measurement switch: pread() straight into the slot
	bool no_inflate_ahead = false, no_table = false, no_ahead = false;
	bool timing = false;
	// A job that gets the output set while its bytes are still on their way: launches that follow the batches as they arrive hide
	// the inflate behind a SLOW source (a file that is not in the page cache) -- but beside launches of its own job a fast upload
	// runs at half its rate (28 GB/s against 50, the copy threads waiting for ring slots: profiles/r04j_cohort_two_modes.log), the
	// next job then finds its bytes a fifth up when ITS turn comes, and the cohort stays at 50 ms per sample instead of 32.  So:
	// when the rest of the bytes will be up within `ahead_wait_factor` times what the whole job takes to inflate (at the rate they
	// have come at so far -- a throttled upload included, hence the factor), the job waits for them and is ONE launch; the upload
	// behind it then runs beside that one launch at full rate and is complete when its own turn comes: one cycle, and the steady
	// state is the fast one.  0: follow the batches whatever the rate (round 4's rule until its last day).
	double ahead_wait_factor = 2.0;
	double inflate_ms_per_gb = 20.0; // of file: 184 GB/s inflated at a ratio of 3.7 (a 1x genome: 1.4 GB, 28 ms)
};

// One sample's compressed bytes on their way to HBM (the upload thread runs it): host threads copy pieces of the file into the ring
// of pinned buffers, each piece goes up on the copy stream as soon as it is full, and behind every batch of pieces an event is
// recorded that the inflate launches of that batch wait for.
struct Job {
	Machine *m = nullptr;
	ByteSource src;
	size_t n_bytes = 0, piece = 0, n_pieces = 0, pieces_per_batch = 0, n_batches = 0;
	int buf = -1;        // which of the two device buffers (taken when the job starts)
	std::atomic<bool> released{false}; // nothing reads the buffer any more
	uint64_t ticket = 0; // name_next()'s
	std::mutex mu;
	std::condition_variable cv;
	std::vector<uint8_t> filled;
	size_t issued = 0;        // pieces whose copy up has been enqueued
	size_t batches_ready = 0; // batches whose event has been recorded
	bool failed = false, short_read = false, done = false, started = false;
	bool queued = false; // handed to the upload thread (bytes named ahead wait for the call in front of theirs to queue its own)
	bool adopted = false; // a call has taken the job up (mu): no inflating ahead begins behind its back
	std::atomic<bool> cancel{false};
	std::vector<void *> ev_batch;
	const uint8_t *d_bytes = nullptr; // where the bytes go (set when the job starts)
	std::string error;
	std::chrono::steady_clock::time_point t_queued, t_started;
	std::chrono::steady_clock::time_point t_upload_begin; // the job has its device buffer: its first piece is about to be read (written before `started`)
	double ms_enqueued = 0, ms_copy = 0, ms_wait = 0; // (timing)
	int n_threads = 0;
	// inflate ahead: a thread launches the batches' inflates into the spare output set
	std::vector<conga_bgzf_block> blocks;
	std::vector<uint64_t> out_off;
	uint64_t total_out = 0;
	std::thread inflater;
	bool inflate_asked = false, inflate_done = false, inflate_ok = false; // (mu)
	int launches_ahead = 0;
	size_t first_launch_blocks = 0; // (timing)
	double ms_inflate_ahead = 0;
	// The block table read off the bytes as they pass through the pinned ring (name_next with the block starts the caller knows
	// from the index): every copying thread walks the chain of headers inside its piece from the first known start on, the upload
	// thread joins the pieces' findings in file order (what straddles two pieces it reads from the file: a trailer, now and then a
	// header) and publishes the table batch by batch -- the inflate-ahead thread needs no more.
	struct PieceTable {
		std::vector<conga_bgzf_block> blocks; // complete inside the piece
		uint64_t seed = ~0ull;                 // where the walk began (~0: no known start inside the piece)
		uint64_t arrived = 0;                  // where it stopped: the offset of the next header
		bool open = false;                     // the last block's header is in, its trailer lies behind the piece
		conga_bgzf_block open_block = {};
		uint64_t open_end = 0;
		bool bad = false;                      // something that is not a BGZF block of the usual form
	};
	bool build_table = false;
	std::vector<uint64_t> known; // piece-relative offsets of block headers, ascending
	uint64_t stop_at = 0;        // 0, or one more than an offset: the table ends with the first block that starts at or behind that offset
	                             // (one more: a next target that begins in the stretch's FIRST block is offset 0 -- which read as "none"
	                             // and left the engine's table longer than the file's, tests/soak.py --bam seed 3001 case 362)
	std::vector<PieceTable> piece_tables;
	uint64_t expect = 0;         // where the chain goes on (upload thread)
	bool table_failed = false, table_stopped = false; // (upload thread; published with the counters below)
	size_t cap_blocks = 0;
	uint64_t cap_out = 0;        // 0: no inflate ahead (only the table)
	size_t table_n = 0;          // blocks published (mu)
	bool table_final = false, table_ok = false; // (mu)
	~Job()
	{
		if (inflater.joinable())
			inflater.join();
		for (void *e : ev_batch)
			if (e && m)
				m->event_destroy(e);
	}
};

// BGZF header of the usual form at h[0..18) -> BSIZE + 1 (the block's length), or 0
inline uint32_t bgzf_block_len(const uint8_t *h)
{
	if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[10] != 6 || h[11] != 0 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0)
		return 0;
	const uint32_t len = (uint32_t) (h[16] | (h[17] << 8)) + 1u;
	return len >= 26u ? len : 0u;
}

// one piece's share of the table: buf = the piece's bytes [begin, begin + len) of the job's stretch
inline void walk_piece(Job &job, size_t c, const uint8_t *buf, uint64_t begin, size_t len)
{
	Job::PieceTable &pt = job.piece_tables[c];
	const uint64_t end = begin + len;
	const auto it = std::lower_bound(job.known.begin(), job.known.end(), begin);
	if (it == job.known.end() || *it >= end)
		return;
	pt.seed = *it;
	uint64_t at = pt.seed;
	while (at < end) {
		if (at + 18 > end)
			break; // (the header straddles the piece's end: the upload thread reads it from the file)
		const uint8_t *h = buf + (at - begin);
		const uint32_t blen = bgzf_block_len(h);
		if (!blen) {
			pt.bad = true;
			break;
		}
		const uint64_t bend = at + blen;
		if (bend > job.n_bytes)
			break; // the stretch ends inside this block (behind the last one that counts)
		conga_bgzf_block b = {};
		b.data_off = at + 18;
		b.data_len = blen - 26u;
		if (bend <= end) {
			memcpy(&b.crc32, buf + (bend - 8 - begin), 4);
			memcpy(&b.inflated_len, buf + (bend - 4 - begin), 4);
			if (b.inflated_len)
				pt.blocks.push_back(b);
			at = bend;
		} else {
			pt.open = true;
			pt.open_block = b;
			pt.open_end = bend;
			at = bend;
			break;
		}
	}
	pt.arrived = at;
}

// piece c is in its slot (buf): its findings join the table.  Upload thread, pieces in file order.
inline void join_piece(Job &job, size_t c, const uint8_t *buf, uint64_t begin, size_t len)
{
	if (job.table_failed || job.table_stopped)
		return;
	const uint64_t end = begin + len;
	Job::PieceTable &pt = job.piece_tables[c];
	auto read_at = [&](uint64_t off, void *dst, size_t n) { // from the slot when it is all there, from the file otherwise
		if (off >= begin && off + n <= end) {
			memcpy(dst, buf + (off - begin), n);
			return true;
		}
		return off + n <= job.n_bytes && job.src.fetch((size_t) off, dst, n);
	};
	auto append = [&](const conga_bgzf_block &b) {
		if (b.inflated_len == 0)
			return;
		if (b.inflated_len > 65536u || job.blocks.size() >= job.cap_blocks) {
			job.table_failed = true;
			return;
		}
		job.out_off.push_back(job.total_out);
		job.blocks.push_back(b);
		job.total_out += b.inflated_len;
		if (job.stop_at && b.data_off - 18 + 1 >= job.stop_at)
			job.table_stopped = true; // (the block in which the next target begins is in: enough)
	};
	if (pt.bad) {
		job.table_failed = true;
		return;
	}
	// the chain from where it stood up to the piece's first known start (or through the whole piece when it has none): blocks the
	// index does not know, a header cut by the piece before
	const uint64_t upto = pt.seed != ~0ull ? pt.seed : end;
	while (job.expect < upto && !job.table_failed && !job.table_stopped) {
		uint8_t h[18];
		if (job.expect + 18 > job.n_bytes || !read_at(job.expect, h, 18)) {
			job.table_stopped = true; // (the stretch ends here)
			return;
		}
		const uint32_t blen = bgzf_block_len(h);
		if (!blen) {
			job.table_failed = true;
			return;
		}
		const uint64_t bend = job.expect + blen;
		if (bend > job.n_bytes) {
			job.table_stopped = true;
			return;
		}
		uint8_t t[8];
		if (!read_at(bend - 8, t, 8)) {
			job.table_failed = true;
			return;
		}
		conga_bgzf_block b = {};
		b.data_off = job.expect + 18;
		b.data_len = blen - 26u;
		memcpy(&b.crc32, t, 4);
		memcpy(&b.inflated_len, t + 4, 4);
		append(b);
		job.expect = bend;
	}
	if (job.table_failed || job.table_stopped || pt.seed == ~0ull)
		return;
	if (job.expect != pt.seed) { // (the chain does not arrive at what the index calls a block's start: the index is not believed)
		job.table_failed = true;
		return;
	}
	for (const conga_bgzf_block &b : pt.blocks) {
		append(b);
		if (job.table_failed || job.table_stopped)
			return;
	}
	if (pt.open) {
		uint8_t t[8];
		if (!read_at(pt.open_end - 8, t, 8)) {
			job.table_failed = true;
			return;
		}
		memcpy(&pt.open_block.crc32, t, 4);
		memcpy(&pt.open_block.inflated_len, t + 4, 4);
		append(pt.open_block);
	}
	job.expect = pt.arrived;
}

class Scheduler {
public:
	typedef std::shared_ptr<Job> JobPtr;

	Machine *m = nullptr;
	Config cfg;
	std::mutex mu; // (was conga_ctx::bz_up_mu)
	std::condition_variable cv;
	std::deque<JobPtr> queue;
	bool quit = false, busy = false;
	// named ahead, not yet taken up by a call, in the order of their naming: at most three (a cohort names two samples ahead, and its
	// planning thread may do so before the call for the sample in front has taken ITS bytes up)
	std::vector<JobPtr> named;
	bool in_call = false;   // a conga_reads_bgzf* call is between queueing its bytes and its return
	double ratio = 0;       // inflated bytes per compressed byte of the largest call so far: sizes the spare output buffer
	JobPtr spare_owner;     // the named job whose inflates fill the spare output set (until the call that takes it up swaps the sets)
	std::set<uint64_t> spare_waiting; // tickets of the jobs that will inflate ahead and have not got the set yet: it goes to the OLDEST (mu)
	JobPtr buf_owner[2];    // a device buffer is its job's until the call that took the bytes up is through with them
	uint64_t tickets = 0;
	bool slot_used[64] = {};
	JobPtr job_kept;        // the job of the last call: its events are waited for by work still in flight
	std::thread up_thread;
	// test hook: called by a job's inflating thread when it starts to wait for the spare set (the standstill needed the job named
	// SECOND to get there first)
	void (*hook_spare_wait)(Scheduler *, Job *) = nullptr;

	size_t slot_bytes() const { return cfg.slot_bytes; }

	// ---- the upload thread
	void upload_loop()
	{
		for (;;) {
			JobPtr job;
			{
				std::unique_lock<std::mutex> lk(mu);
				busy = false;
				cv.notify_all();
				cv.wait(lk, [&] { return quit || !queue.empty(); });
				if (queue.empty())
					return; // (quit, nothing left)
				job = queue.front();
				queue.pop_front();
				busy = true;
			}
			if (job->cancel.load()) {
				std::lock_guard<std::mutex> g(job->mu);
				job->failed = job->done = true;
				job->table_final = true;
				job->error = "given up";
				job->cv.notify_all();
				continue;
			}
			run_job(job);
		}
	}

	void enqueue(const JobPtr &job) // (mu held by the caller)
	{
		if (job->queued)
			return;
		job->queued = true;
		job->t_queued = std::chrono::steady_clock::now();
		if (!up_thread.joinable())
			up_thread = std::thread([this] { upload_loop(); });
		queue.push_back(job);
		cv.notify_all();
	}

	// a job for n_bytes of `src`, queued behind whatever the upload thread is doing if `now` (mu held by the caller)
	JobPtr queue_job(const ByteSource &src, size_t n_bytes, bool now = true)
	{
		JobPtr job = std::make_shared<Job>();
		job->m = m;
		job->src = src;
		job->n_bytes = n_bytes;
		const size_t piece = cfg.piece ? std::min(cfg.slot_bytes, std::max<size_t>(4096, cfg.piece)) : cfg.slot_bytes;
		job->piece = piece;
		job->n_pieces = (n_bytes + piece - 1) / piece;
		// a batch = what one inflate launch takes with three launch streams: 128 MB (small test pieces: sixteen of them)
		job->pieces_per_batch = piece < cfg.slot_bytes ? (size_t) cfg.pieces_per_launch_small : std::max<size_t>(1, ((size_t) 128 << 20) / piece);
		job->n_batches = (job->n_pieces + job->pieces_per_batch - 1) / job->pieces_per_batch;
		job->filled.assign(job->n_pieces, 0);
		job->t_queued = std::chrono::steady_clock::now();
		if (now)
			enqueue(job);
		return job;
	}

	// the job's device buffer may be written again
	void release(const JobPtr &job)
	{
		if (!job)
			return;
		std::lock_guard<std::mutex> g(mu);
		job->released.store(true);
		cv.notify_all();
	}

	// A call that launches its job's inflates itself (nothing was inflated ahead) keeps the spare set out of the reach of the jobs
	// behind until ITS sample's compute is enqueued as well (spare_free), like a call that swapped the set in: the walks, the layout
	// and the compute of the sample in front are a few milliseconds of launches and, the first time, dozens of allocations -- and an
	// allocation beside a launch that holds every wave slot for 28 ms waits for it (a cohort's first sample: "walks + checks 537 ms").
	// false: a job is inflating into the set already.
	bool hold_spare(const JobPtr &job)
	{
		std::lock_guard<std::mutex> g(mu);
		if (spare_owner)
			return false;
		spare_owner = job;
		return true;
	}

	// the spare output set is nobody's again (when it was this job's)
	void spare_free(const JobPtr &job)
	{
		std::lock_guard<std::mutex> g(mu);
		if (spare_owner == job) {
			spare_owner.reset();
			cv.notify_all();
		}
	}

	// gives a job up and waits until the upload thread is through with it
	void abandon(const JobPtr &job)
	{
		if (!job)
			return;
		job->cancel.store(true);
		job->cv.notify_all();
		{
			std::lock_guard<std::mutex> g(mu);
			cv.notify_all(); // (it may be waiting for a buffer, its inflating thread for the spare set)
			if (!job->queued) { // (never handed to the upload thread: nobody else will say it is done)
				std::lock_guard<std::mutex> g2(job->mu);
				job->failed = job->done = true;
				job->table_final = true;
				job->cv.notify_all();
			}
		}
		{
			std::unique_lock<std::mutex> lk(job->mu);
			job->cv.wait(lk, [&] { return job->done && (!job->inflate_asked || job->inflate_done); });
		}
		if (job->inflater.joinable())
			job->inflater.join();
		if (job->inflate_asked)
			m->ahead_drain(); // (what it launched reads the job's bytes)
		{
			std::lock_guard<std::mutex> g(mu);
			spare_waiting.erase(job->ticket);
			cv.notify_all();
		}
		spare_free(job);
		release(job);
	}

	// everything the upload thread has been given is through (conga_release_staging, conga_destroy)
	void quiesce(bool and_quit)
	{
		std::vector<JobPtr> pre;
		{
			std::lock_guard<std::mutex> g(mu);
			pre.swap(named);
		}
		for (const JobPtr &j : pre)
			abandon(j);
		m->prewarm_join();
		release(job_kept); // (the call that took those bytes up has returned: nothing reads them)
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return queue.empty() && !busy; });
			if (and_quit)
				quit = true;
			cv.notify_all();
		}
		if (and_quit && up_thread.joinable())
			up_thread.join();
	}

	// ---- one job, on the upload thread
	void run_job(const JobPtr &self)
	{
		Job &job = *self;
		auto ms_since = [](std::chrono::steady_clock::time_point t) {
			return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
		};
		auto give_up = [&](const std::string &why, bool short_read) {
			std::lock_guard<std::mutex> g(job.mu);
			job.failed = true;
			job.short_read = job.short_read || short_read;
			if (job.error.empty())
				job.error = why;
			job.done = true;
			job.table_final = true;
			job.cv.notify_all();
		};
		job.t_started = std::chrono::steady_clock::now();
		trace("job %llu: the upload thread takes it up (%zu bytes)", (unsigned long long) job.ticket, job.n_bytes);
		if (!m->bind())
			return give_up("the device cannot be selected", false);
		m->prewarm_join(); // (the buffers it allocates are about to be looked at)
		{ // one of the two device buffers: the one no job owns, or whose job's bytes nobody reads any more
			std::unique_lock<std::mutex> lk(mu);
			auto free_buf = [&]() {
				for (int b = 0; b < 2; b++)
					if (!buf_owner[b] || buf_owner[b]->released.load())
						return b;
				return -1;
			};
			cv.wait(lk, [&] { return free_buf() >= 0 || job.cancel.load(); });
			if (job.cancel.load()) {
				lk.unlock();
				return give_up("given up", false);
			}
			job.buf = free_buf();
			buf_owner[job.buf] = self;
		}
		trace("job %llu: has device buffer %d", (unsigned long long) job.ticket, job.buf);
		// the device buffer (grown only: a cohort's samples are of a size) and the batches' events
		uint8_t *const d_dst = m->up_buffer(job.buf, job.n_bytes + 512);
		if (!d_dst)
			return give_up("no device memory for the file's bytes", false);
		job.ev_batch.assign(job.n_batches, nullptr);
		for (size_t k = 0; k < job.n_batches; k++)
			if (!(job.ev_batch[k] = m->event_create()))
				return give_up("an event cannot be made", false);
		// The table is made here, batch by batch: the inflates can follow it into the spare output set (its thread waits until the set
		// is free: the named job in front of this one owns it until the call that takes THAT one up has swapped it in) -- when a
		// call of this context has shown how much such a file inflates to, and the job is a named one.
		bool ahead = false;
		if (job.build_table && job.ticket != 0 && !cfg.no_inflate_ahead && m->ahead_possible()) {
			std::lock_guard<std::mutex> g(mu);
			if (ratio > 0) {
				job.cap_out = (uint64_t) ((double) job.n_bytes * ratio * 1.25) + ((uint64_t) 64 << 20);
				ahead = true;
			}
		}
		if (ahead) { // (in the order the jobs begin, which is the order they were named in: the spare set goes to the oldest ticket)
			std::lock_guard<std::mutex> g(mu);
			spare_waiting.insert(job.ticket);
		}
		bool inflating = false;
		{
			std::lock_guard<std::mutex> g(job.mu);
			job.d_bytes = d_dst;
			job.t_upload_begin = std::chrono::steady_clock::now();
			job.started = true;
			if (ahead && !job.inflate_asked && !job.adopted && !job.cancel.load()) {
				job.inflate_asked = inflating = true;
				job.inflater = std::thread([this, self] { inflate_ahead(self); });
			}
		}
		if (ahead && !inflating) { // (taken up by its call, or given up, in the meantime: it will not ask for the set)
			std::lock_guard<std::mutex> g(mu);
			spare_waiting.erase(job.ticket);
			cv.notify_all();
		}
		job.cv.notify_all();

		const int n_slots = cfg.n_slots;
		const size_t piece = job.piece, n_pieces = job.n_pieces, n_bytes = job.n_bytes;
		std::atomic<size_t> next_piece{0};
		std::atomic<long long> us_copy{0}, us_wait{0}, us_wait_order{0};
		std::atomic<int> slow_slot_waits{0};
		auto worker = [&]() {
			(void) m->bind();
			for (;;) {
				const size_t c = next_piece.fetch_add(1);
				if (c >= n_pieces || job.cancel.load())
					return;
				const auto tw = std::chrono::steady_clock::now();
				const int slot = (int) (c % (size_t) n_slots);
				bool first_use;
				if (c >= (size_t) n_slots) { // the slot still holds piece c - n_slots until that one's copy up is done
					std::unique_lock<std::mutex> lk(job.mu);
					job.cv.wait(lk, [&] { return job.failed || job.cancel.load() || job.issued > c - (size_t) n_slots; });
					if (job.failed || job.cancel.load())
						return;
					first_use = false;
				} else {
					std::lock_guard<std::mutex> g(job.mu); // (slot_used is written by the upload thread under this lock)
					first_use = !slot_used[slot];
				}
				const auto to_ = std::chrono::steady_clock::now();
				us_wait_order += (long long) std::chrono::duration<double, std::micro>(to_ - tw).count();
				// (a slot's first use in this job: the job before may have left its last pieces in the ring)
				const bool slot_ok = first_use || m->slot_wait(slot);
				if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - to_).count() > 300.0)
					slow_slot_waits++;
				if (!slot_ok) {
					std::lock_guard<std::mutex> g(job.mu);
					job.failed = true;
					job.cv.notify_all();
					return;
				}
				const size_t at = c * piece, len = std::min(piece, n_bytes - at);
				const auto tc = std::chrono::steady_clock::now();
				uint8_t *dst = m->ring_slot(slot);
				const bool got = cfg.plain_pread ? job.src.fetch(at, dst, len) : job.src.fetch_into_ring(at, dst, len);
				if (got && job.build_table)
					walk_piece(job, c, dst, at, len);
				us_wait += (long long) std::chrono::duration<double, std::micro>(tc - tw).count();
				us_copy += (long long) std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tc).count();
				{
					std::lock_guard<std::mutex> g(job.mu);
					job.filled[c] = 1;
					if (!got) {
						job.failed = job.short_read = true; // (a file that ends early -- or a read that failed: said as such)
						if (job.src.io_error.load() && job.error.empty())
							job.error = "reading the file failed";
					}
				}
				job.cv.notify_all();
				if (!got)
					return;
			}
		};
		// (half of the cores this process MAY use -- the cgroup's quota, not the machine's 256 --: 6-10 GB/s of page cache -> pinned
		// memory per thread against a link of 45-55 GB/s.  With a quota of 16: 8 threads have a 1.4 GB file enqueued after 29-35 ms, 6
		// after 28-34, 12 after 37-39 (profiles/r03d_upload_modes.log); the other half is the caller's -- a cohort reads the next sample's
		// block table meanwhile on six threads -- and a quota overdrawn stalls them all: 12 + 16 threads made a 5x genome's upload take
		// 450-1 200 ms instead of 290-370)
		int n_threads = (int) std::min<size_t>(n_pieces, std::min<unsigned>(std::max(2u, cfg.cpus / 2), (unsigned) n_slots));
		if (cfg.copy_threads > 0)
			n_threads = std::max(1, std::min(cfg.copy_threads, n_slots));
		job.n_threads = n_threads;
		std::vector<std::thread> threads;
		for (int t = 0; t < n_threads; t++)
			threads.emplace_back(worker);

		std::string why;
		double us_pieces = 0, us_join = 0, us_issue = 0; // (the upload thread's own time: waiting for a filled piece, the table, the copy calls)
		auto lap = [](std::chrono::steady_clock::time_point &t) {
			const auto n = std::chrono::steady_clock::now();
			const double us = std::chrono::duration<double, std::micro>(n - t).count();
			t = n;
			return us;
		};
		auto t_lap = std::chrono::steady_clock::now();
		for (size_t c = 0; c < n_pieces && why.empty(); c++) {
			{
				std::unique_lock<std::mutex> lk(job.mu);
				job.cv.wait(lk, [&] { return job.failed || job.filled[c] || job.cancel.load(); });
				if (job.failed || job.cancel.load()) {
					why = job.cancel.load() ? "given up" : job.short_read ? "the file ends inside the piece that was named" : "waiting for a pinned piece failed";
					break;
				}
			}
			const size_t at = c * piece, len = std::min(piece, n_bytes - at);
			const int slot = (int) (c % (size_t) n_slots);
			us_pieces += lap(t_lap);
			if (job.build_table)
				join_piece(job, c, m->ring_slot(slot), at, len);
			us_join += lap(t_lap);
			bool ok = m->copy_up(d_dst + at, slot, len);
			const bool batch_end = (c + 1) % job.pieces_per_batch == 0 || c + 1 == n_pieces;
			const size_t batch = c / job.pieces_per_batch;
			if (ok && batch_end)
				ok = m->event_record(job.ev_batch[batch]);
			us_issue += lap(t_lap);
			if (!ok) {
				why = "copy up failed";
				break;
			}
			{
				std::lock_guard<std::mutex> g(job.mu);
				slot_used[slot] = true;
				job.issued = c + 1;
				if (batch_end) {
					job.batches_ready = batch + 1;
					job.table_n = job.blocks.size();
					if (job.build_table && (job.table_failed || c + 1 == n_pieces)) {
						job.table_final = true;
						job.table_ok = !job.table_failed && !job.blocks.empty();
					}
				}
			}
			job.cv.notify_all();
		}
		if (!why.empty()) {
			std::lock_guard<std::mutex> g(job.mu);
			job.failed = true;
			if (job.error.empty())
				job.error = why;
		}
		job.cv.notify_all();
		for (std::thread &t : threads)
			t.join();
		job.ms_enqueued = ms_since(job.t_started);
		trace("job %llu: every piece is enqueued%s (the upload thread: %.1f ms waiting for filled pieces, %.1f ms joining the table, %.1f ms in the copy calls; "
				"a copying thread: %.1f ms copying, %.1f ms waiting for its turn in the ring, %.1f ms for the slot's copy up to end -- %d of those waits over 0.3 ms)",
				(unsigned long long) job.ticket, why.empty() ? "" : " (given up)", us_pieces / 1e3, us_join / 1e3, us_issue / 1e3, us_copy / 1e3 / n_threads,
				us_wait_order / 1e3 / n_threads, (us_wait - us_wait_order) / 1e3 / n_threads, slow_slot_waits.load());
		job.ms_copy = us_copy / 1e3 / n_threads;
		job.ms_wait = us_wait / 1e3 / n_threads;
		{
			std::lock_guard<std::mutex> g(job.mu);
			job.done = true;
			if (job.build_table && !job.table_final) { // (given up on the way)
				job.table_final = true;
				job.table_ok = false;
			}
		}
		job.cv.notify_all();
	}

	// ---- a job's inflating thread: the batches' inflates into the spare output set, behind the upload
	void inflate_ahead(const JobPtr &self)
	{
		Job &job = *self;
		bool ok = m->bind();
		m->prewarm_join();
		if (hook_spare_wait)
			hook_spare_wait(this, &job);
		{ // the spare output set: one job at a time, in the order the jobs were named
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] {
#if defined(BZ_TEST_SPARE_TO_WHOEVER_WAKES_FIRST) // (tests/test_bz_sched.py only: round 3's rule until its last day -- the standstill)
				return job.cancel.load() || !spare_owner;
#else
				return job.cancel.load() || (!spare_owner && !spare_waiting.empty() && *spare_waiting.begin() == job.ticket);
#endif
			});
			spare_waiting.erase(job.ticket);
			if (job.cancel.load())
				ok = false;
			else
				spare_owner = self;
			cv.notify_all(); // (the next ticket may be waiting for this one to be out of the way)
		}
		trace("job %llu: its inflating thread has the spare output set", (unsigned long long) job.ticket);
		const auto t0 = std::chrono::steady_clock::now();
		// room for the table and the stream: the table's own size when the caller brought it, a bound when it grows with the upload
		size_t room_blocks;
		uint64_t room_out;
		{
			std::lock_guard<std::mutex> g(job.mu);
			room_blocks = std::max(job.cap_blocks, job.table_n);
			room_out = std::max<uint64_t>(job.cap_out, job.table_final ? job.total_out : 0);
		}
		ok = ok && m->spare_reserve(room_blocks, room_out);
		size_t b_done = 0;
		int launches = 0;
		// One launch takes every batch that is up by now: a job that gets the set when its bytes have long arrived -- the rule in a
		// cohort's steady state -- is ONE launch of all its blocks (79 086 of a 1x genome: ten rounds of the machine's 8 192 waves,
		// the last round's idle tail paid once) instead of eleven launches of one round each, 27-30 ms instead of 38
		// (profiles/r04b_cohort_spare_held.log); a job whose bytes are still on their way gets a launch per batch as before.
		for (size_t batch = 0; ok && batch < job.n_batches; batch++) {
			size_t n_avail = 0;
			bool final = false;
			{
				std::unique_lock<std::mutex> lk(job.mu);
				job.cv.wait(lk, [&] { return job.failed || job.cancel.load() || job.batches_ready > batch; });
				if (!job.failed && !job.cancel.load() && job.batches_ready < job.n_batches && cfg.ahead_wait_factor > 0) {
					// (see Config::ahead_wait_factor) the rest of the bytes at the rate they have come at so far
					const double so_far = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job.t_upload_begin).count();
					const double rest = so_far / (double) job.batches_ready * (double) (job.n_batches - job.batches_ready);
					if (rest <= cfg.ahead_wait_factor * cfg.inflate_ms_per_gb * (double) job.n_bytes / 1e9) {
						trace("job %llu: %zu of %zu batches are up, the rest within %.0f ms: one launch when they are", (unsigned long long) job.ticket,
								job.batches_ready, job.n_batches, rest);
						batch = job.n_batches - 1;
						job.cv.wait(lk, [&] { return job.failed || job.cancel.load() || job.batches_ready > batch; });
					}
				}
				if (job.failed || job.cancel.load() || (job.table_final && !job.table_ok)) {
					ok = false;
					break;
				}
				batch = job.batches_ready - 1; // (the newest batch whose event is recorded: its copies are behind all earlier ones)
				n_avail = job.table_n;
				final = job.table_final;
			}
			const bool last = batch + 1 == job.n_batches;
			const size_t have = std::min(job.n_bytes, (batch + 1) * job.pieces_per_batch * job.piece);
			size_t b1 = b_done;
			while (b1 < n_avail && job.blocks[b1].data_off + job.blocks[b1].data_len + 8 <= have)
				b1++;
			if (last) {
				ok = final;
				b1 = n_avail;
			}
			if (ok && b1 > b_done) {
				const size_t n = b1 - b_done;
				ok = job.out_off[b1 - 1] + job.blocks[b1 - 1].inflated_len <= room_out && b1 <= room_blocks
						&& m->ahead_launch(job.ev_batch[batch], job.d_bytes, job.blocks.data(), job.out_off.data(), b_done, n, launches);
				if (launches == 0)
					job.first_launch_blocks = n;
				trace("job %llu: inflate launch %d made ahead, %zu blocks (batch %zu of %zu)", (unsigned long long) job.ticket, launches, n, batch + 1, job.n_batches);
				launches++;
				b_done = b1;
			}
		}
		ok = ok && m->ahead_mark();
		job.launches_ahead = launches;
		job.ms_inflate_ahead = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		{
			std::lock_guard<std::mutex> g(job.mu);
			job.inflate_ok = ok;
			job.inflate_done = true;
		}
		job.cv.notify_all();
		// The compressed bytes have done their duty once the last launch is through: their buffer goes back now -- the job named
		// behind this one can start on its way up before the call for this one has even begun (two buffers carry any depth).
		if (ok) {
			const bool through = m->ahead_wait();
			trace("job %llu: its launches are through", (unsigned long long) job.ticket);
			bool uploaded;
			{
				std::unique_lock<std::mutex> lk(job.mu);
				job.cv.wait(lk, [&] { return job.done; });
				uploaded = !job.failed;
			}
			if (through && uploaded)
				release(self);
		}
	}

	// ---- the callers' side --------------------------------------------------------------------------------------------------------

	// conga_reads_bgzf_next_fd: names bytes a later call will bring.  ring_ready: the pinned ring and its copy stream exist.
	uint64_t name_next(int fd, uint64_t file_off, size_t n_bytes, const uint64_t *known_starts, size_t n_known, uint64_t stop_at, bool ring_ready)
	{
		std::lock_guard<std::mutex> g(mu);
		if (!ring_ready || named.size() >= 3 || quit)
			return 0;
		ByteSource src;
		src.fd = fd;
		src.file_off = file_off;
		// (inside a call: behind that call's bytes, at once.  Between calls -- or before the call in front of these bytes' own has
		// begun, which a cohort's planning thread cannot know --: when the next call begins, behind its bytes or as its bytes)
		// (only behind a job that is itself on its way: what was named first goes up first)
		const bool now = in_call && (named.empty() || named.back()->queued);
		JobPtr job = queue_job(src, n_bytes, false);
		if (n_known && known_starts[0] == 0 && !cfg.no_table) {
			job->build_table = true;
			job->known.assign(known_starts, known_starts + n_known);
			job->stop_at = stop_at;
			job->piece_tables.resize(job->n_pieces);
			job->cap_blocks = n_bytes / 4096 + 65536; // (BAM writers fill a block with ~64 KB of records: 10-30 KB deflated)
			job->blocks.reserve(job->cap_blocks);     // (the inflate-ahead thread reads what is published while the upload thread appends)
			job->out_off.reserve(job->cap_blocks);
		}
		job->ticket = ++tickets;
		trace("job %llu: named (%zu bytes)%s", (unsigned long long) job->ticket, n_bytes, now ? ", queued for the upload thread" : ", to be queued later");
		if (now)
			enqueue(job);
		named.push_back(job);
		return job->ticket;
	}

	// conga_reads_bgzf_next_blocks: the caller read the table by itself; the named bytes can be inflated ahead with it.
	// tables_ready: a call of this context has inflated something (the inflate's own tables are on the device).
	void bring_table(uint64_t ticket, const conga_bgzf_block *blocks, size_t n_blocks, bool tables_ready)
	{
		if (ticket == 0 || cfg.no_inflate_ahead)
			return;
		// (one spare output set: one named job at a time is inflated ahead -- the set is free again when the call that takes those
		// bytes up has swapped it in)
		std::lock_guard<std::mutex> g(mu);
		JobPtr job;
		for (const JobPtr &j : named) {
			std::lock_guard<std::mutex> gj(j->mu);
			if (j->ticket == ticket)
				job = j;
			else if (j->inflate_asked)
				return;
		}
		if (!job || !tables_ready) // (taken up already, or no call of this context has inflated anything yet)
			return;
		// the table as conga_reads_bgzf* checks it: in file order, inside the bytes, 1..64 KiB each
		std::vector<uint64_t> out_off(n_blocks);
		uint64_t total = 0;
		for (size_t b = 0; b < n_blocks; b++) {
			const conga_bgzf_block &bl = blocks[b];
			if (bl.data_off > job->n_bytes || (uint64_t) bl.data_len > job->n_bytes - bl.data_off || bl.inflated_len == 0 || bl.inflated_len > 65536u
					|| (b && bl.data_off < blocks[b - 1].data_off + blocks[b - 1].data_len))
				return; // (the call itself will say what is wrong with it)
			out_off[b] = total;
			total += bl.inflated_len;
		}
		std::lock_guard<std::mutex> gj(job->mu);
		if (job->inflate_asked || job->adopted || job->cancel.load())
			return;
		if (job->build_table)
			return; // (the engine reads the table off the bytes itself)
		job->blocks.assign(blocks, blocks + n_blocks);
		job->out_off.swap(out_off);
		job->total_out = total;
		job->table_n = n_blocks;
		job->table_final = job->table_ok = true;
		job->inflate_asked = true;
		// (the spare output set goes to the oldest ticket that WAITS for it: a job whose table the caller brought never passes
		// run_job's `ahead` branch, so it is entered here -- without this its thread slept for ever and the call that adopted the job
		// with it)
#if !defined(BZ_TEST_BROUGHT_TABLE_NOT_ENTERED) // (tests/test_bz_sched.py only: round 3's last commit -- the job sleeps for ever)
		spare_waiting.insert(job->ticket);
#endif
		JobPtr self = job;
		job->inflater = std::thread([this, self] { inflate_ahead(self); });
	}

	// conga_reads_bgzf_next_table: waits for the table the engine reads off the named bytes; nothing: *n_blocks stays 0
	void wait_table(uint64_t ticket, const conga_bgzf_block **blocks, size_t *n_blocks)
	{
		JobPtr job;
		{
			std::unique_lock<std::mutex> lk(mu);
			for (const JobPtr &j : named)
				if (ticket != 0 && j->ticket == ticket)
					job = j;
			if (!job || !job->build_table)
				return;
			// Bytes named before the call in front of theirs has begun go up when it does -- a cohort's planning thread names sample
			// k + 1 a few milliseconds before the call for sample k begins.  Not for ever: with no such call to come (the caller
			// decodes sample k on the host after all) there will be no table, and the caller must not wait for one.
			// (a wait on the system clock: pthread_cond_timedwait -- the steady clock's pthread_cond_clockwait is not known to the thread
			// sanitizer of this toolchain, which then loses track of who holds the mutex)
			cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(400), [&] { return job->queued; });
			if (!job->queued)
				return;
		}
		std::unique_lock<std::mutex> lk(job->mu);
		job->cv.wait(lk, [&] { return job->table_final || job->done; });
		if (job->table_final && job->table_ok) {
			*blocks = job->blocks.data();
			*n_blocks = job->blocks.size();
		}
	}

	// conga_reads_bgzf_next_go: the caller will bring nothing in front of these bytes -- they (and whatever was named before them)
	// start on their way now instead of when the next call begins.  A no-op for bytes already on their way or taken up.
	void go(uint64_t ticket)
	{
		std::lock_guard<std::mutex> g(mu);
		size_t upto = named.size();
		for (size_t k = 0; ticket != 0 && k < named.size(); k++)
			if (named[k]->ticket == ticket)
				upto = k;
		for (size_t k = 0; upto < named.size() && k <= upto; k++) {
			if (!named[k]->queued)
				trace("job %llu: told to go", (unsigned long long) named[k]->ticket);
			enqueue(named[k]); // (in the order they were named)
		}
	}

	// conga_reads_bgzf_forget
	void forget(uint64_t ticket)
	{
		JobPtr job;
		{
			std::lock_guard<std::mutex> g(mu);
			for (size_t k = 0; ticket != 0 && k < named.size(); k++)
				if (named[k]->ticket == ticket) {
					job = named[k];
					named.erase(named.begin() + (long) k);
					break;
				}
		}
		abandon(job); // (returns when the upload thread no longer reads from the descriptor)
	}

	// A call begins: the job that brings its bytes -- the one named for exactly these, already on its way (*ahead), or one of the
	// call's own, in front of named bytes whose upload has not begun.
	JobPtr adopt(const ByteSource &src, size_t n_bytes, bool *ahead)
	{
		// (the call before has returned behind its walks: nothing reads its compressed bytes any more)
		release(job_kept);
		job_kept.reset();
		JobPtr job;
		*ahead = false;
		{
			std::lock_guard<std::mutex> g(mu);
			in_call = true; // (until end_call)
			for (size_t k = 0; k < named.size() && !job; k++) {
				Job &p = *named[k];
				if (src.fd >= 0 && p.src.fd == src.fd && p.src.file_off == src.file_off && p.n_bytes == n_bytes) {
					job = named[k];
					*ahead = job->queued;
					enqueue(job); // (named between two calls: it starts now)
					named.erase(named.begin() + (long) k);
				}
			}
		}
		if (job) { // (one that failed before it was asked for is no reason to fail now: start over)
			std::unique_lock<std::mutex> lk(job->mu);
			if (job->failed) {
				lk.unlock();
				abandon(job);
				job.reset();
				*ahead = false;
			}
		}
		if (!job) {
			// Bytes named ahead that are NOT these belong to a later call (a cohort's planning thread may name sample k + 1 before
			// sample k's call gets here): they stay named; when their upload has not begun, this call's goes in front of it.
			// With two or more of them, though, both device buffers may be theirs: the ones behind the first are given up (their
			// calls bring them again) -- a caller that names in the order of its calls, as it must, gets here only at a run's start.
			for (;;) {
				JobPtr last;
				{
					std::lock_guard<std::mutex> g(mu);
					if (named.size() >= 2) {
						last = named.back();
						named.pop_back();
					}
				}
				if (!last)
					break;
				abandon(last);
			}
			std::lock_guard<std::mutex> g(mu);
			job = queue_job(src, n_bytes);
			// (in front of named bytes whose upload has not begun)
			for (size_t at = queue.size() - 1; at > 0 && queue[at - 1]->ticket != 0; at--)
				std::swap(queue[at - 1], queue[at]);
		}
		return job;
	}

	// the adopted job was inflated ahead with exactly this table?  Waits for its inflating thread if it has one.
	// -> 0: no inflating thread; 1: inflated ahead with this table (the caller swaps the output sets in, then calls spare_free);
	//    -1: the thread is through but its work is not this call's (the caller lets it drain, then spare_free)
	int take_inflated(const JobPtr &job, const conga_bgzf_block *blocks, size_t n_blocks, bool same_base)
	{
		bool asked;
		{
			std::lock_guard<std::mutex> g(job->mu);
			job->adopted = true; // (a job that has not started yet will not start inflating ahead now: this call launches its inflates)
			asked = job->inflate_asked;
		}
		if (!asked)
			return 0;
		{
			std::unique_lock<std::mutex> lk(job->mu);
			job->cv.wait(lk, [&] { return job->inflate_done; });
		}
		if (job->inflater.joinable())
			job->inflater.join();
		return job->inflate_ok && same_base && job->blocks.size() == n_blocks
				&& memcmp(job->blocks.data(), blocks, n_blocks * sizeof(conga_bgzf_block)) == 0 ? 1 : -1;
	}

	// the bytes of the calls after this one: behind this call's (and behind the swap of the output sets)
	void enqueue_later()
	{
		std::lock_guard<std::mutex> g(mu);
		for (const JobPtr &later : named)
			enqueue(later);
	}

	void end_call()
	{
		std::lock_guard<std::mutex> g(mu);
		in_call = false;
	}
};

} // namespace bz
