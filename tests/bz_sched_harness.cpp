// bz_sched_harness.cpp -- conga_amd/csrc/bz_sched.h (the upload pipeline of conga_reads_bgzf*: jobs, tickets, who owns which
// buffer) driven WITHOUT a GPU: a fake machine in ordinary memory, BGZF files of stored blocks written on the spot, and a caller
// that behaves like `conga --cohort` (read_bam_cohort, bam_data.cpp: planning threads that name the next two samples' bytes and
// sleep in wait_table; the thread of the calls adopts, swaps or launches, ends the call).  tests/test_bz_sched.py builds it with
// -fsanitize=thread and runs every scenario; a watchdog turns a standstill into exit code 3.
//
// The fake "inflate" copies a stored block's payload (BTYPE 00: one byte of header, LEN, NLEN, the bytes) to its place in the
// output stream -- enough to tell WHICH job's blocks lie in an output set when a call takes it up.
#include "../conga_amd/csrc/bz_sched.h"

#include <fcntl.h>
#include <sys/stat.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

namespace {

std::atomic<long> g_progress{0};
std::atomic<bool> g_finished{false};

void fail(const char *what)
{
	fprintf(stderr, "FAILED: %s\n", what);
	fflush(stderr);
	_exit(2);
}

struct Sample {
	std::string path;
	std::vector<uint8_t> file;             // the BGZF file
	std::vector<uint8_t> stream;           // its blocks' payloads one behind the other
	std::vector<conga_bgzf_block> blocks;  // the non-empty ones, file order
	std::vector<uint64_t> starts;          // where every block begins (the index knows some of them)
	int fd = -1;
};

// a BGZF file of stored blocks: payloads of 1 .. max_payload bytes, an empty block now and then, the EOF marker
Sample make_sample(const std::string &path, std::mt19937 &rng, size_t n_blocks, size_t max_payload)
{
	Sample s;
	s.path = path;
	for (size_t b = 0; b <= n_blocks; b++) {
		const bool eof = b == n_blocks;
		size_t len = eof || rng() % 17 == 0 ? 0 : 1 + rng() % max_payload;
		std::vector<uint8_t> payload(len);
		for (size_t i = 0; i < len; i++)
			payload[i] = (uint8_t) (rng() >> 7);
		const size_t deflate_len = len ? len + 5 : 2; // stored: 01 LEN NLEN bytes; empty: 03 00
		const size_t bsize = 18 + deflate_len + 8;
		const size_t at = s.file.size();
		s.starts.push_back(at);
		s.file.resize(at + bsize);
		uint8_t *h = s.file.data() + at;
		const uint8_t head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t) ((bsize - 1) & 255), (uint8_t) ((bsize - 1) >> 8)};
		memcpy(h, head, 18);
		if (len) {
			h[18] = 1;
			h[19] = (uint8_t) (len & 255), h[20] = (uint8_t) (len >> 8);
			h[21] = (uint8_t) (~len & 255), h[22] = (uint8_t) ((~len >> 8) & 255);
			memcpy(h + 23, payload.data(), len);
		} else
			h[18] = 3, h[19] = 0;
		const uint32_t crc = 0x1234u + (uint32_t) b, isize = (uint32_t) len;
		memcpy(h + bsize - 8, &crc, 4);
		memcpy(h + bsize - 4, &isize, 4);
		if (len) {
			conga_bgzf_block bl = {};
			bl.data_off = at + 18;
			bl.data_len = (uint32_t) deflate_len;
			bl.inflated_len = isize;
			bl.crc32 = crc;
			s.blocks.push_back(bl);
			s.stream.insert(s.stream.end(), payload.begin(), payload.end());
		}
	}
	FILE *f = fopen(path.c_str(), "wb");
	if (!f || fwrite(s.file.data(), 1, s.file.size(), f) != s.file.size())
		fail("cannot write a sample");
	fclose(f);
	s.fd = open(path.c_str(), O_RDONLY);
	if (s.fd < 0)
		fail("cannot open a sample");
	return s;
}

struct FakeMachine final : bz::Machine {
	size_t slot_bytes;
	int n_slots;
	std::vector<uint8_t> ring;
	std::vector<uint8_t> up[2];
	// two output sets: [0] = the context's, [1] = the spare one (the caller swaps them like conga_api.hip swaps its DevBufs)
	std::vector<uint8_t> out[2];
	std::mutex mu;
	std::map<void *, bool> events; // recorded?
	std::atomic<int> launches{0}, spare_users{0};
	std::atomic<bool> can_ahead{true};
	std::mt19937 jitter{7};

	FakeMachine(size_t slot, int slots) : slot_bytes(slot), n_slots(slots), ring(slot * (size_t) slots + 64) {}
	void nap()
	{
		unsigned r;
		{
			std::lock_guard<std::mutex> g(mu);
			r = jitter() % 8;
		}
		if (r == 0)
			std::this_thread::sleep_for(std::chrono::microseconds(200));
		else if (r < 3)
			std::this_thread::yield();
	}
	bool bind() override { return true; }
	uint8_t *ring_slot(int slot) override { return (uint8_t *) (((uintptr_t) ring.data() + 63) & ~(uintptr_t) 63) + (size_t) slot * slot_bytes; }
	bool slot_wait(int) override
	{
		nap();
		return true; // (copies are synchronous here: the slot is free when copy_up has returned)
	}
	uint8_t *up_buffer(int which, size_t bytes) override
	{
		if (up[which].size() < bytes)
			up[which].assign(bytes + bytes / 16, 0xEE);
		return up[which].data();
	}
	void *event_create() override
	{
		std::lock_guard<std::mutex> g(mu);
		void *e = malloc(1);
		events[e] = false;
		return e;
	}
	void event_destroy(void *ev) override
	{
		std::lock_guard<std::mutex> g(mu);
		events.erase(ev);
		free(ev);
	}
	bool copy_up(uint8_t *dst, int slot, size_t len) override
	{
		memcpy(dst, ring_slot(slot), len);
		g_progress++;
		nap();
		return true;
	}
	bool event_record(void *ev) override
	{
		std::lock_guard<std::mutex> g(mu);
		events[ev] = true;
		return true;
	}
	bool ahead_possible() override { return can_ahead.load(); }
	bool spare_reserve(size_t, uint64_t out_bytes) override
	{
		if (spare_users.fetch_add(1) != 0)
			fail("two jobs inflate into the spare output set at once");
		if (out[1].size() < out_bytes + 64)
			out[1].assign(out_bytes + 64, 0xDD);
		return true;
	}
	void inflate(std::vector<uint8_t> &dst, const uint8_t *d_bytes, const conga_bgzf_block *blocks, const uint64_t *out_off, size_t first, size_t n)
	{
		for (size_t b = first; b < first + n; b++) {
			const uint8_t *p = d_bytes + blocks[b].data_off;
			if (p[0] != 1 || (size_t) (p[1] | p[2] << 8) != blocks[b].inflated_len || out_off[b] + blocks[b].inflated_len > dst.size())
				fail("a block handed to the inflate is not the stored block its table says");
			memcpy(dst.data() + out_off[b], p + 5, blocks[b].inflated_len);
		}
	}
	bool ahead_launch(void *batch_event, const uint8_t *d_bytes, const conga_bgzf_block *blocks, const uint64_t *out_off, size_t first, size_t n, int) override
	{
		{
			std::lock_guard<std::mutex> g(mu);
			if (!events.count(batch_event) || !events[batch_event])
				fail("an inflate was launched behind a batch whose event was never recorded");
		}
		inflate(out[1], d_bytes, blocks, out_off, first, n);
		launches++;
		g_progress++;
		nap();
		return true;
	}
	bool ahead_mark() override { return true; }
	bool ahead_wait() override { return true; }
	void ahead_drain() override {}
	void prewarm_join() override {}
	// the adopting call is through with the spare set (it swapped it in, or let it drain): the next job may reserve it
	void spare_done() { spare_users.store(0); }
};

struct Run {
	FakeMachine machine;
	bz::Scheduler sched;
	Run(size_t slot, int slots, size_t piece, int copy_threads) : machine(slot, slots)
	{
		sched.m = &machine;
		sched.cfg.slot_bytes = slot;
		sched.cfg.n_slots = slots;
		sched.cfg.piece = piece;
		sched.cfg.pieces_per_launch_small = 4;
		sched.cfg.copy_threads = copy_threads;
		sched.cfg.cpus = 8;
	}
};

// one conga_reads_bgzf_fd call for sample s with table `blocks` (conga_api.hip: upload_and_inflate_overlapped): adopt, take what
// was inflated ahead or inflate here, end the call.  -> what happened: 'a' inflated ahead, 'c' in the call, 'f' the upload failed
char one_call(Run &r, const Sample &s, const std::vector<conga_bgzf_block> &blocks, bool expect_failure = false)
{
	bz::ByteSource src;
	src.fd = s.fd;
	{ // (reads_bgzf_from: the largest ratio of inflated to compressed bytes so far sizes the spare set)
		std::lock_guard<std::mutex> g(r.sched.mu);
		r.sched.ratio = std::max(r.sched.ratio, (double) s.stream.size() / (double) std::max<size_t>(s.file.size(), 1));
	}
	bool ahead = false;
	std::shared_ptr<bz::Job> job = r.sched.adopt(src, s.file.size(), &ahead);
	std::vector<uint64_t> out_off(blocks.size());
	uint64_t total = 0;
	for (size_t b = 0; b < blocks.size(); b++) {
		out_off[b] = total;
		total += blocks[b].inflated_len;
	}
	char how = 'c';
	const int took = r.sched.take_inflated(job, blocks.data(), blocks.size(), true);
	if (took > 0) {
		std::swap(r.machine.out[0], r.machine.out[1]);
		how = 'a';
	}
	if (took != 0) {
		r.machine.spare_done();
		r.sched.spare_free(job);
	}
	const bool held = took == 0 && r.sched.hold_spare(job); // (upload_and_inflate_overlapped: until the sample's compute is enqueued)
	r.sched.enqueue_later();
	bool failed = false;
	if (how == 'c') {
		if (r.machine.out[0].size() < total + 64)
			r.machine.out[0].assign(total + 64, 0xCC);
		size_t b_done = 0;
		for (size_t batch = 0; batch < job->n_batches && !failed; batch++) {
			{
				std::unique_lock<std::mutex> lk(job->mu);
				job->cv.wait(lk, [&] { return job->failed || job->batches_ready > batch; });
				failed = job->failed;
			}
			if (failed)
				break;
			const size_t have = std::min(s.file.size(), (batch + 1) * job->pieces_per_batch * job->piece);
			size_t b1 = b_done;
			while (b1 < blocks.size() && blocks[b1].data_off + blocks[b1].data_len <= have)
				b1++;
			if (batch + 1 == job->n_batches)
				b1 = blocks.size();
			if (b1 > b_done)
				r.machine.inflate(r.machine.out[0], job->d_bytes, blocks.data(), out_off.data(), b_done, b1 - b_done);
			b_done = b1;
		}
	}
	if (failed)
		r.sched.abandon(job);
	else {
		std::unique_lock<std::mutex> lk(job->mu);
		job->cv.wait(lk, [&] { return job->done; });
		failed = job->failed;
	}
	r.sched.job_kept = job;
	if (held)
		r.sched.spare_free(job); // ("conga_chrom_compute")
	r.sched.end_call();
	g_progress++;
	if (failed != expect_failure)
		fail(expect_failure ? "an upload that had to fail did not" : "an upload failed");
	if (!failed && (r.machine.out[0].size() < total || memcmp(r.machine.out[0].data(), s.stream.data(), total) != 0))
		fail(how == 'a' ? "the output set swapped in does not hold this sample's stream (another job's inflates went into the spare set?)"
				: "the stream inflated in the call is not the sample's");
	return failed ? 'f' : how;
}

std::vector<uint64_t> known_of(const Sample &s, std::mt19937 &rng)
{
	std::vector<uint64_t> k;
	for (size_t i = 0; i < s.starts.size(); i++)
		if (i == 0 || rng() % 3 == 0)
			k.push_back(s.starts[i]);
	return k;
}

// ---- scenario: a cohort, the next samples named `depth` deep from planning threads, as read_bam_cohort does it
void cohort(const char *dir, unsigned seed, size_t piece, int copy_threads, int depth, int n_samples, bool caller_tables, bool second_first)
{
	std::mt19937 rng(seed);
	std::unique_ptr<Run> run(new Run((size_t) 64 << 10, 6, piece, copy_threads)); // (on the heap: a mutex that comes back at the same stack address confuses the sanitizer)
	Run &r = *run;
	std::vector<Sample> samples;
	for (int k = 0; k < n_samples; k++)
		samples.push_back(make_sample(std::string(dir) + "/s" + std::to_string(seed) + "_" + std::to_string(k) + ".bgzf", rng, 40 + rng() % 200, 9000));
	if (second_first) // the standstill's order of events: the job named SECOND gets to the spare set's door first
		r.sched.hook_spare_wait = [](bz::Scheduler *, bz::Job *j) {
			if (j->ticket % 2 == 1)
				std::this_thread::sleep_for(std::chrono::milliseconds(60));
		};
	std::vector<std::thread> planners((size_t) n_samples);
	std::vector<std::vector<conga_bgzf_block>> tables((size_t) n_samples);
	std::unique_ptr<std::atomic<bool>[]> named(new std::atomic<bool>[(size_t) n_samples]);
	std::unique_ptr<std::atomic<uint64_t>[]> tickets(new std::atomic<uint64_t>[(size_t) n_samples]);
	std::unique_ptr<std::atomic<bool>[]> begun(new std::atomic<bool>[(size_t) n_samples]); // sample j's call is about to begin (bam_data.cpp: call_begins)
	for (int j = 0; j < n_samples; j++)
		named[j] = false, tickets[j] = 0, begun[j] = false;
	auto launch = [&](int j, int ahead_of) {
		if (j >= n_samples || planners[(size_t) j].joinable() || named[j].load())
			return;
		std::mt19937 prng(seed * 131u + (unsigned) j);
		planners[(size_t) j] = std::thread([&, j, ahead_of, prng]() mutable {
			while (j > 0 && !named[j - 1].load())
				std::this_thread::sleep_for(std::chrono::microseconds(100));
			const bool tell = j <= ahead_of || tickets[j - 1].load() != 0;
			uint64_t t = 0;
			if (tell) {
				const std::vector<uint64_t> known = caller_tables ? std::vector<uint64_t>() : known_of(samples[(size_t) j], prng);
				t = r.sched.name_next(samples[(size_t) j].fd, 0, samples[(size_t) j].file.size(), known.data(), known.size(), 0, true);
			}
			if (t && begun[j - 1].load())
				r.sched.go(t); // (the call in front has begun -- or is over: the next one is this sample's own, and it waits for this thread)
			tickets[j] = t;
			named[j] = true;
			tables[(size_t) j] = samples[(size_t) j].blocks; // (what walking the file gives)
			if (t && caller_tables)
				r.sched.bring_table(t, tables[(size_t) j].data(), tables[(size_t) j].size(), true);
			else if (t) {
				const conga_bgzf_block *b = nullptr;
				size_t n = 0;
				r.sched.wait_table(t, &b, &n);
				if (n) {
					if (n != samples[(size_t) j].blocks.size() || memcmp(b, samples[(size_t) j].blocks.data(), n * sizeof(conga_bgzf_block)) != 0)
						fail("the table read off the bytes is not the file's");
					tables[(size_t) j].assign(b, b + n);
				}
			}
		});
	};
	named[0] = true;
	tables[0] = samples[0].blocks;
	std::string how;
	for (int k = 0; k < n_samples; k++) {
		if (planners[(size_t) k].joinable())
			planners[(size_t) k].join();
		if (depth >= 1)
			launch(k + 1, k + 1);
		if (depth >= 2 && k > 0)
			launch(k + 2, k + 1);
		begun[k] = true;
		how += one_call(r, samples[(size_t) k], tables[(size_t) k]);
		if (tickets[k].load())
			r.sched.forget(tickets[k].load()); // (a no-op for a ticket taken up)
	}
	for (std::thread &t : planners)
		if (t.joinable())
			t.join();
	r.sched.quiesce(true);
	printf("cohort seed %u piece %zu threads %d depth %d%s%s: %s (%d launches ahead)\n", seed, piece, copy_threads, depth, caller_tables ? " caller's tables" : "",
			second_first ? " second-first" : "", how.c_str(), r.machine.launches.load());
	if (depth >= 1 && n_samples >= 4 && how.find('a') == std::string::npos)
		fail("no sample of the cohort was inflated ahead");
	for (Sample &s : samples) {
		close(s.fd);
		unlink(s.path.c_str());
	}
}

// ---- scenario: jobs given up in every state, a file that ends early, bytes named for nobody, the context's end with jobs named
void give_ups(const char *dir, unsigned seed)
{
	std::mt19937 rng(seed);
	std::unique_ptr<Run> run(new Run((size_t) 64 << 10, 6, 4096, 3));
	Run &r = *run;
	std::vector<Sample> s;
	for (int k = 0; k < 5; k++)
		s.push_back(make_sample(std::string(dir) + "/g" + std::to_string(seed) + "_" + std::to_string(k) + ".bgzf", rng, 60 + rng() % 100, 9000));
	one_call(r, s[0], s[0].blocks);
	// named between two calls and forgotten before anything started
	uint64_t t = r.sched.name_next(s[1].fd, 0, s[1].file.size(), nullptr, 0, 0, true);
	r.sched.forget(t);
	// named with known starts between calls, then a call for OTHER bytes: the named ones stay named, go up behind, and are taken up next
	std::vector<uint64_t> known = known_of(s[2], rng);
	t = r.sched.name_next(s[2].fd, 0, s[2].file.size(), known.data(), known.size(), 0, true);
	one_call(r, s[1], s[1].blocks);
	one_call(r, s[2], s[2].blocks);
	r.sched.forget(t);
	// three named, none asked for; a call for a fourth file: the ones behind the first are given up
	uint64_t ta = r.sched.name_next(s[0].fd, 0, s[0].file.size(), nullptr, 0, 0, true);
	uint64_t tb = r.sched.name_next(s[1].fd, 0, s[1].file.size(), nullptr, 0, 0, true);
	uint64_t tc = r.sched.name_next(s[2].fd, 0, s[2].file.size(), nullptr, 0, 0, true);
	if (!ta || !tb || !tc || r.sched.name_next(s[3].fd, 0, s[3].file.size(), nullptr, 0, 0, true) != 0)
		fail("three stretches are held named, a fourth is not");
	one_call(r, s[3], s[3].blocks);
	one_call(r, s[0], s[0].blocks); // (the first one named is still there)
	r.sched.forget(ta), r.sched.forget(tb), r.sched.forget(tc);
	// a file that ends early: named longer than it is -- the job fails, the call that adopts it starts over and fails the same way
	t = r.sched.name_next(s[4].fd, 0, s[4].file.size() + 5000, nullptr, 0, 0, true);
	{
		Sample longer = s[4];
		longer.file.resize(s[4].file.size() + 5000);
		one_call(r, longer, s[4].blocks, true);
	}
	r.sched.forget(t);
	one_call(r, s[4], s[4].blocks);
	// a table brought for a ticket that was taken up already, for ticket 0, twice: nothing happens
	r.sched.bring_table(0, s[4].blocks.data(), s[4].blocks.size(), true);
	t = r.sched.name_next(s[1].fd, 0, s[1].file.size(), nullptr, 0, 0, true);
	r.sched.bring_table(t, s[1].blocks.data(), s[1].blocks.size(), true);
	r.sched.bring_table(t, s[1].blocks.data(), s[1].blocks.size(), true);
	// ... and the wrong table for the bytes: the call finds the spare set is not its own and inflates by itself
	if (one_call(r, s[1], std::vector<conga_bgzf_block>(s[1].blocks.begin(), s[1].blocks.end() - 1)) == 'a')
		fail("a call with another table took the stream inflated ahead");
	// named between two calls, and the table asked for before any call begins -- a cohort's thread of the calls waits for the next
	// sample's plan, the plan for the table, the table for the bytes, and the bytes (named between calls) for a call: told to go,
	// they start by themselves and the table is there long before wait_table's 400 ms are over
	known = known_of(s[3], rng);
	t = r.sched.name_next(s[3].fd, 0, s[3].file.size(), known.data(), known.size(), 0, true);
	r.sched.go(t);
	r.sched.go(t); // (twice, and for a ticket nobody holds: nothing happens)
	r.sched.go(12345);
	{
		const auto t0 = std::chrono::steady_clock::now();
		const conga_bgzf_block *b = nullptr;
		size_t n = 0;
		r.sched.wait_table(t, &b, &n);
		const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		if (n != s[3].blocks.size() || memcmp(b, s[3].blocks.data(), n * sizeof(conga_bgzf_block)) != 0)
			fail("bytes told to go: no table, or not the file's");
		if (ms > 300.0)
			fail("bytes told to go waited for a call all the same");
	}
	one_call(r, s[3], s[3].blocks);
	r.sched.forget(t);
	// the context ends with jobs named and one inflating ahead
	known = known_of(s[2], rng);
	(void) r.sched.name_next(s[2].fd, 0, s[2].file.size(), known.data(), known.size(), 0, true);
	t = r.sched.name_next(s[3].fd, 0, s[3].file.size(), nullptr, 0, 0, true);
	r.sched.bring_table(t, s[3].blocks.data(), s[3].blocks.size(), true);
	r.sched.quiesce(true);
	printf("give-ups seed %u ok\n", seed);
	for (Sample &x : s) {
		close(x.fd);
		unlink(x.path.c_str());
	}
}

} // namespace

int main(int argc, char **argv)
{
	const char *dir = argc > 1 ? argv[1] : "/tmp";
	const std::string what = argc > 2 ? argv[2] : "all";
	const int patience = argc > 3 ? std::max(1, atoi(argv[3])) : 10; // seconds without a copy, a launch or a call
	std::thread watchdog([patience] {
		long last = -1;
		int still = 0;
		while (!g_finished.load()) {
			std::this_thread::sleep_for(std::chrono::milliseconds(250));
			const long now = g_progress.load();
			still = now == last ? still + 1 : 0;
			last = now;
			if (still >= 4 * patience) {
				fprintf(stderr, "STANDSTILL: nothing moved for %d seconds\n", patience);
				fflush(stderr);
				_exit(3);
			}
		}
	});
	if (what == "all" || what == "cohort") {
		unsigned seed = 1;
		for (size_t piece : {(size_t) 4096, (size_t) 6144, (size_t) 20000, (size_t) 0})
			for (int threads : {1, 3})
				for (int depth : {0, 1, 2})
					cohort(dir, seed++, piece, threads, depth, 6, false, false);
		cohort(dir, 90, 4096, 2, 1, 6, true, false); // the caller brings the tables (conga_reads_bgzf_next_blocks)
		cohort(dir, 91, 6144, 3, 2, 7, true, false);
	}
	if (what == "all" || what == "second_first") {
		cohort(dir, 81, 16384, 3, 2, 4, false, true); // round 3's standstill: two named ahead, uploads that take no time
		cohort(dir, 82, 6144, 1, 2, 6, false, true);
	}
	if (what == "all" || what == "brought_table") {
		cohort(dir, 95, 8192, 2, 1, 4, true, false); // a job whose table the caller brought
		cohort(dir, 96, 8192, 2, 2, 5, true, true);
	}
	if (what == "all" || what == "give_ups")
		for (unsigned seed = 1; seed <= 3; seed++)
			give_ups(dir, seed);
	g_finished = true;
	watchdog.join();
	puts("ok");
	return 0;
}
