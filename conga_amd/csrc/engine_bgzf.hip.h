// engine_bgzf.hip.h -- count_reads_bam with the BAM decode on the device (bam_data.c:192-221,253-259,293: htslib's BGZF reader and
// record iterator in the reference): the inflate launches, the pinned ring, the HIP side of the upload pipeline (bz_sched.h holds
// its scheduler, host code tested without a GPU), the record walks, and the conga_reads_bgzf* / conga_inflate_blocks entry points.
// Part of conga_api.hip's one translation unit.
#pragma once

namespace {

// the byte-wise CRC-32 table (polynomial 0xEDB88320), once per context
int ensure_crc_table(conga_ctx *ctx)
{
	if (ctx->d_bz_crc.p)
		return CONGA_OK;
	uint32_t table[256];
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i;
		for (int k = 0; k < 8; k++)
			c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
		table[i] = c;
	}
	TRY(upload(ctx, ctx->d_bz_crc, table, sizeof table));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (`table` is on the stack)
	return CONGA_OK;
}

// BGZF inflate of blocks [b0, b0 + n) of d_bz_blocks / d_bz_off on stream `st`: d_bz_in -> d_bz_out, one status byte per block.
// Default: one block per WAVE (inflate_wave.hip.h).  CONGA_BGZF_KERNEL=lane: the host decoder's source one block per
// lane (round 1's kernel, kept for comparison); `lanes` sizes its per-lane scratch.
bool lane_kernel_asked(const conga_ctx *ctx)
{
	return ctx->knobs.bgzf_lane_kernel;
}

constexpr size_t kBzTickets = 1024; // a counter per launch, taken in turn: a launch is long through when its counter comes round again

// a zeroed counter for one launch on stream `st` (any thread)
uint32_t *bz_ticket(conga_ctx *ctx, hipStream_t st)
{
	if (!ctx->d_bz_ticket.p)
		return nullptr;
	uint32_t *t = ptr<uint32_t>(ctx->d_bz_ticket) + (ctx->bz_ticket_next.fetch_add(1) % kBzTickets);
	if (hipMemsetAsync(t, 0, 4, st) != hipSuccess) {
		(void) hipGetLastError();
		return nullptr; // (round robin then)
	}
	return t;
}

int ensure_x2n(conga_ctx *ctx)
{
	if (ctx->d_bz_x2n.p)
		return CONGA_OK;
	// x^(2^k) mod P for the CRC-32 polynomial, reflected (bit 31 = x^0): the wave combines its lanes' partial CRCs with them
	uint32_t x2n[32];
	auto mul = [](uint32_t a, uint32_t b) {
		uint32_t p = 0;
		for (int k = 0; k < 32; k++) {
			if ((a >> (31 - k)) & 1u)
				p ^= b;
			b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
		}
		return p;
	};
	x2n[0] = 0x40000000u; // x
	for (int k = 1; k < 32; k++)
		x2n[k] = mul(x2n[k - 1], x2n[k - 1]);
	TRY(upload(ctx, ctx->d_bz_x2n, x2n, sizeof x2n));
	TRY(ensure(ctx, ctx->d_bz_ticket, kBzTickets * 4)); // the launches' block counters (inflate_wave.hip.h: `ticket`)
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (`x2n` is on the stack)
	return CONGA_OK;
}

int launch_inflate(conga_ctx *ctx, size_t n_blocks, uint32_t lanes, hipStream_t st = nullptr, size_t b0 = 0, const uint8_t *in = nullptr)
{
	if (!st)
		st = ctx->stream;
	if (!in)
		in = ptr<uint8_t>(ctx->d_bz_in);
	if (lane_kernel_asked(ctx)) {
		TRY(ensure(ctx, ctx->d_bz_scratch, (size_t) lanes * sizeof(InflateScratch)));
		hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(lanes / 64), dim3(64), 0, st, (uint32_t) n_blocks, in,
				ptr<conga_bgzf_block>(ctx->d_bz_blocks) + b0, ptr<uint64_t>(ctx->d_bz_off) + b0, ptr<uint8_t>(ctx->d_bz_out),
				ptr<InflateScratch>(ctx->d_bz_scratch), ptr<uint32_t>(ctx->d_bz_crc), ptr<uint8_t>(ctx->d_bz_status) + b0);
		return CONGA_OK;
	}
	TRY(ensure_x2n(ctx));
	// one resident round of workgroups (8 per CU), blocks round robin over their waves
	const size_t groups = std::min<size_t>((n_blocks + iw::kWavesPerGroup - 1) / iw::kWavesPerGroup, (size_t) ctx->n_cu * (size_t) ctx->knobs.bgzf_groups_per_cu);
	// CONGA_BGZF_KERNEL=wave1: round 2's symbol loop (every trip decodes its sixty-four candidates completely), for comparison
	const bool one_phase = ctx->knobs.bgzf_one_phase;
	uint32_t *ticket = ctx->knobs.bgzf_round_robin ? nullptr : bz_ticket(ctx, st);
	if (one_phase)
		hipLaunchKernelGGL(iw::bgzf_inflate_wave_kernel<false>, dim3((unsigned) groups), dim3(64 * iw::kWavesPerGroup), 0, st, (uint32_t) n_blocks,
				in, ptr<conga_bgzf_block>(ctx->d_bz_blocks) + b0, ptr<uint64_t>(ctx->d_bz_off) + b0,
				ptr<uint8_t>(ctx->d_bz_out), ptr<uint32_t>(ctx->d_bz_crc), ptr<uint32_t>(ctx->d_bz_x2n), ptr<uint8_t>(ctx->d_bz_status) + b0, ticket);
	else
		hipLaunchKernelGGL(iw::bgzf_inflate_wave_kernel<true>, dim3((unsigned) groups), dim3(64 * iw::kWavesPerGroup), 0, st, (uint32_t) n_blocks,
				in, ptr<conga_bgzf_block>(ctx->d_bz_blocks) + b0, ptr<uint64_t>(ctx->d_bz_off) + b0,
				ptr<uint8_t>(ctx->d_bz_out), ptr<uint32_t>(ctx->d_bz_crc), ptr<uint32_t>(ctx->d_bz_x2n), ptr<uint8_t>(ctx->d_bz_status) + b0, ticket);
	return CONGA_OK;
}

// The file's bytes to HBM and the inflate of their blocks, overlapped.  A pageable hipMemcpy of gigabytes runs at the rate
// of ONE staging thread inside the runtime (~18 GB/s measured); here host threads copy 16 MB pieces of the caller's bytes
// (the page cache behind an mmap) into a ring of pinned buffers, each piece goes up at the link's rate as soon as it is
// full, and every 128 MB of pieces the inflate of the blocks they complete is launched on one of three streams, so that
// copying in, copying up and inflating all run at once.  Ends with ctx->stream waiting for every launch.
// blocks[] must be in file order (data_off ascending); the caller falls back to the plain form otherwise.
// a slot of the ring (CONGA_BGZF_SLOT_MB: measurement switch; pieces are at most a slot)
#define kBzPiece (ctx->knobs.bgzf_slot_bytes)
constexpr int kBzPiecesPerLaunch = 16;
// how many of them are used (CONGA_BGZF_SLOTS / CONGA_BGZF_STREAMS: measurement switches)
#define bz_slots() (ctx->knobs.bgzf_slots)
#define bz_streams_wanted() (ctx->knobs.bgzf_streams)

// the pinned ring, its events and the streams of the overlapped upload (96 MB of pinned memory take ~50 ms to get: with
// CONGA_FLAG_EXPECT_BGZF conga_create() does this, and a caller that creates its context on a thread of its own -- the
// conga executable does, while it reads the BAM's block table -- never waits for it)
void make_bz_ring(conga_ctx *ctx)
{
	int prio_low = 0, prio_high = 0; // (numerically lower = more urgent)
	(void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
	const bool prio = !ctx->knobs.bgzf_no_priority;
	bool ok = hipSetDevice(ctx->device) == hipSuccess
			&& hipHostMalloc((void **) &ctx->h_bz_ring, kBzPiece * (size_t) bz_slots(), hipHostMallocDefault) == hipSuccess;
	if (ok && ctx->bz_copy) // (the ring was given back, conga_release_staging: streams and events are still there)
		return;
	ok = ok && hipStreamCreateWithPriority(&ctx->bz_copy, hipStreamNonBlocking, prio ? prio_high : 0) == hipSuccess;
	if (ok && ctx->knobs.bgzf_copy_streams > 1)
		ok = hipStreamCreateWithPriority(&ctx->bz_copy2, hipStreamNonBlocking, prio ? prio_high : 0) == hipSuccess
				&& hipEventCreateWithFlags(&ctx->ev_bz_copy2, hipEventDisableTiming) == hipSuccess;
	for (int k = 0; ok && k < bz_slots(); k++)
		ok = hipEventCreateWithFlags(&ctx->ev_bz_slot[k], hipEventDisableTiming) == hipSuccess;
	// The inflate launches need streams BELOW the copy stream's priority (equal priorities: the pieces go up at 22 GB/s beside
	// the kernels instead of 50; copy high / kernels normal: the stage takes 116 ms instead of 93).  A stream costs 15-20 ms to
	// make on this platform, so the context's own two streams -- made with the lowest priority, conga_create -- take the
	// launches (two streams instead of three dedicated ones: +3 ms for the stage, -50 ms for the creation).
	// CONGA_STREAMS_NORMAL=1: the context's streams at the default priority and three streams of their own for the inflate.
	if (ctx->bz_shared) {
		ctx->bz_kernel[0] = ctx->stream2;
		ctx->bz_kernel[1] = ctx->stream;
		ctx->n_bz_streams = std::min(2, bz_streams_wanted());
	} else
		ctx->n_bz_streams = bz_streams_wanted();
	for (int k = 0; ok && k < ctx->n_bz_streams; k++)
		ok = (ctx->bz_shared || hipStreamCreateWithPriority(&ctx->bz_kernel[k], hipStreamNonBlocking, prio ? prio_low : 0) == hipSuccess)
				&& hipEventCreateWithFlags(&ctx->ev_bz_kernel[k], hipEventDisableTiming) == hipSuccess;
	if (!ok) {
		(void) hipGetLastError();
		ctx->bz_ring_failed = true;
	}
}

// cores this process may use: the affinity mask and the cgroup's CPU quota (a container's 16 of the machine's 256)
unsigned cpus_allowed()
{
	static const unsigned n = [] {
		unsigned c = std::max(1u, std::thread::hardware_concurrency());
		cpu_set_t set;
		if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
			c = std::min(c, (unsigned) CPU_COUNT(&set));
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
			char q[32] = "";
			long long period = 0;
			if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
				c = std::min(c, (unsigned) std::max(1LL, (atoll(q) + period - 1) / period));
			fclose(f);
		}
		return c;
	}();
	return n;
}

using bz::ByteSource;

bool quiet_ensure(DevBuf &b, size_t bytes);
void bz_prewarm_join(conga_ctx *ctx);

// What the upload pipeline (bz_sched.h: jobs, tickets, who owns which buffer -- host code, tested without a GPU) asks of the
// machine, in HIP: the pinned ring and its copy stream, the two device buffers for compressed bytes, the spare output set and the
// launches that fill it ahead.
struct HipMachine final : bz::Machine {
	conga_ctx *ctx;
	explicit HipMachine(conga_ctx *c) : ctx(c) {}
	bool bind() override { return hipSetDevice(ctx->device) == hipSuccess; }
	uint8_t *ring_slot(int slot) override { return ctx->h_bz_ring + (size_t) slot * kBzPiece; }
	bool slot_wait(int slot) override
	{
		if (!ctx->knobs.bgzf_slot_spin)
			return hipEventSynchronize(ctx->ev_bz_slot[slot]) == hipSuccess;
		for (;;) { // (measurement switch)
			const hipError_t e = hipEventQuery(ctx->ev_bz_slot[slot]);
			if (e == hipSuccess)
				return true;
			if (e != hipErrorNotReady) {
				(void) hipGetLastError();
				return false;
			}
			for (int k = 0; k < 64; k++)
				__builtin_ia32_pause();
		}
	}
	uint8_t *up_buffer(int which, size_t bytes) override
	{ // (grown only: a cohort's samples are of a size)
		std::lock_guard<std::mutex> g(ctx->prewarm_mu); // (bz_prealloc, on the caller's thread, may be at the same buffers)
		if (ctx->bz_up_cap[which] < bytes) {
			if (ctx->bz_up_buf[which])
				(void) hipFree(ctx->bz_up_buf[which]);
			ctx->bz_up_buf[which] = nullptr;
			ctx->bz_up_cap[which] = 0;
			const size_t want = bytes + bytes / 16;
			if (hipMalloc((void **) &ctx->bz_up_buf[which], want) != hipSuccess) {
				(void) hipGetLastError();
				return nullptr;
			}
			ctx->bz_up_cap[which] = want;
		}
		return ctx->bz_up_buf[which];
	}
	void *event_create() override
	{
		hipEvent_t e = nullptr;
		return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? (void *) e : nullptr;
	}
	void event_destroy(void *ev) override { (void) hipEventDestroy((hipEvent_t) ev); }
	bool copy_up(uint8_t *dst, int slot, size_t len) override
	{
		hipStream_t cs = (ctx->bz_copy2 && (slot & 1)) ? ctx->bz_copy2 : ctx->bz_copy;
		return hipMemcpyAsync(dst, ring_slot(slot), len, hipMemcpyHostToDevice, cs) == hipSuccess && hipEventRecord(ctx->ev_bz_slot[slot], cs) == hipSuccess;
	}
	bool event_record(void *ev) override
	{
		if (ctx->bz_copy2 // (what the event says -- every piece so far is up -- holds for both streams)
				&& (hipEventRecord(ctx->ev_bz_copy2, ctx->bz_copy2) != hipSuccess || hipStreamWaitEvent(ctx->bz_copy, ctx->ev_bz_copy2, 0) != hipSuccess))
			return false;
		return hipEventRecord((hipEvent_t) ev, ctx->bz_copy) == hipSuccess;
	}
	// inflating ahead: when a call of this context has inflated something (the CRC tables are on the device), no chromosome holds
	// reference text (split reads are mapped on the inflated stream where it lies: no spare set) and the kernel is the usual one
	bool ahead_possible() override
	{
		return !ctx->sr_layout.load() && ctx->d_bz_x2n.p && ctx->d_bz_crc.p && !ctx->knobs.bgzf_other_kernel;
	}
	bool spare_reserve(size_t n_blocks, uint64_t out_bytes) override
	{
		const bool grows = n_blocks * sizeof(conga_bgzf_block) > ctx->d_bz_blocks2.cap || n_blocks * 8 > ctx->d_bz_off2.cap
				|| (size_t) out_bytes + 64 > ctx->d_bz_out2.cap || n_blocks > ctx->d_bz_status2.cap;
		if (grows)
			bz::trace("the spare output set grows: %zu blocks, %.0f MB of output (it holds %.0f MB)", n_blocks, (double) out_bytes / 1e6, (double) ctx->d_bz_out2.cap / 1e6);
		bool ok = quiet_ensure(ctx->d_bz_blocks2, n_blocks * sizeof(conga_bgzf_block)) && quiet_ensure(ctx->d_bz_off2, n_blocks * 8)
				&& quiet_ensure(ctx->d_bz_out2, (size_t) out_bytes + 64) && quiet_ensure(ctx->d_bz_status2, n_blocks);
		if (grows)
			bz::trace("the spare output set has grown");
		if (ok && !ctx->bz_ahead[0]) { // the launch streams of the inflate ahead (lowest priority), made by its first thread
			int lo = 0, hi = 0;
			ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
			for (int k = 0; ok && k < 2; k++)
				ok = hipStreamCreateWithPriority(&ctx->bz_ahead[k], hipStreamNonBlocking, lo) == hipSuccess
						&& hipEventCreateWithFlags(&ctx->ev_bz_ahead[k], hipEventDisableTiming) == hipSuccess;
		}
		if (!ok)
			(void) hipGetLastError();
		return ok;
	}
	bool ahead_launch(void *batch_event, const uint8_t *d_bytes, const conga_bgzf_block *blocks, const uint64_t *out_off, size_t first, size_t n,
			int launch) override
	{
		hipStream_t ks = ctx->bz_ahead[ctx->knobs.bgzf_ahead_one_stream ? 0 : launch % 2];
		// A launch ahead is up to a whole sample's blocks on every wave slot of the machine, resident until its last block is through
		// (27-30 ms of a 1x genome), and the compute of the sample in front is a CHAIN of small launches: once these waves are in, the
		// chain's next link finds no slot and that compute -- 0.3 ms of work -- ends when the inflate does (a run at 41 ms per sample
		// instead of 32, profiles/r04b_cohort_one_launch.log).  The set was handed on when that compute was enqueued: its launches go
		// behind the compute's END, in stream order, without the host in between.
		if (launch < 2 && ctx->computed_once.load(std::memory_order_acquire) && hipStreamWaitEvent(ks, ctx->ev_done, 0) != hipSuccess) {
			(void) hipGetLastError();
			return false;
		}
		bz::trace("launch ahead %d: its tables go up", launch);
		bool ok = hipMemcpyAsync(ptr<conga_bgzf_block>(ctx->d_bz_blocks2) + first, blocks + first, n * sizeof(conga_bgzf_block), hipMemcpyHostToDevice, ks)
						== hipSuccess
				&& hipMemcpyAsync(ptr<uint64_t>(ctx->d_bz_off2) + first, out_off + first, n * 8, hipMemcpyHostToDevice, ks) == hipSuccess
				&& hipMemsetAsync(ptr<uint8_t>(ctx->d_bz_status2) + first, 0xFF, n, ks) == hipSuccess
				&& hipStreamWaitEvent(ks, (hipEvent_t) batch_event, 0) == hipSuccess;
		if (ok) {
			const size_t groups = std::min<size_t>((n + iw::kWavesPerGroup - 1) / iw::kWavesPerGroup, (size_t) ctx->n_cu * (size_t) ctx->knobs.bgzf_groups_per_cu);
			hipLaunchKernelGGL(iw::bgzf_inflate_wave_kernel<true>, dim3((unsigned) groups), dim3(64 * iw::kWavesPerGroup), 0, ks, (uint32_t) n, d_bytes,
					ptr<conga_bgzf_block>(ctx->d_bz_blocks2) + first, ptr<uint64_t>(ctx->d_bz_off2) + first, ptr<uint8_t>(ctx->d_bz_out2),
					ptr<uint32_t>(ctx->d_bz_crc), ptr<uint32_t>(ctx->d_bz_x2n), ptr<uint8_t>(ctx->d_bz_status2) + first,
					ctx->knobs.bgzf_round_robin ? nullptr : bz_ticket(ctx, ks));
			ok = hipGetLastError() == hipSuccess;
		}
		if (!ok)
			(void) hipGetLastError();
		return ok;
	}
	bool ahead_mark() override
	{
		bool ok = ctx->bz_ahead[0] != nullptr;
		for (int k = 0; ok && k < 2; k++)
			ok = hipEventRecord(ctx->ev_bz_ahead[k], ctx->bz_ahead[k]) == hipSuccess;
		if (!ok)
			(void) hipGetLastError();
		return ok;
	}
	bool ahead_wait() override
	{
		bool through = true;
		for (int k = 0; k < 2; k++)
			through = hipEventSynchronize(ctx->ev_bz_ahead[k]) == hipSuccess && through;
		return through;
	}
	void ahead_drain() override
	{
		if (ctx->bz_ahead[0])
			for (int k = 0; k < 2; k++)
				(void) hipStreamSynchronize(ctx->bz_ahead[k]);
	}
	void prewarm_join() override { bz_prewarm_join(ctx); }
};

// CONGA_FLAG_EXPECT_COHORT: what the pipeline of a cohort needs besides the first sample's own buffers -- the second device buffer
// for compressed bytes and the spare output set, ~3.6 bytes of HBM per byte of file -- is allocated by a thread of its own
// while the first sample is inflated, indexed and computed: 45 GB take the runtime 1.3 s, which the second and third sample
// would otherwise wait for (profiles/r03e_cohort_depth.log).
constexpr double kSetRoom = 1.25 * 1.1; // an output set of the pipeline: what a job asks for (1.25 x ratio x bytes), for bytes a tenth more than the first input's

// ... or rather, since round 4's last day, by the first call itself BEFORE anything of it is on the device: hipMalloc is a
// millisecond for gigabytes on an idle device and hundreds of milliseconds beside running kernels and copies, with every launch
// of the process queued behind it meanwhile -- the thread below made a cohort's FIRST sample 0.45 s longer than a sample that
// is alone in its process (its own call: "upload + inflate 624 ms" instead of 38), which was most of `fixed_cost_ms`
// (profiles/r04j_cohort_first_samples.log).  The thread stays for a context whose first call was not told its size this way.
void bz_prealloc(conga_ctx *ctx, size_t n_bytes, double ratio)
{
	if (ctx->bz_prewarmed || !(ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) || ctx->knobs.bgzf_no_inflate_ahead)
		return;
	ctx->bz_prewarmed = true;
	const auto t0 = std::chrono::steady_clock::now();
	{
		std::lock_guard<std::mutex> g(ctx->prewarm_mu);
		const size_t want = n_bytes + n_bytes / 10 + n_bytes / 16 + 512;
		for (int which = 0; which < 2; which++)
			if (ctx->bz_up_cap[which] < want) {
				uint8_t *p = nullptr;
				if (hipMalloc((void **) &p, want) == hipSuccess) {
					if (ctx->bz_up_buf[which])
						(void) hipFree(ctx->bz_up_buf[which]);
					ctx->bz_up_buf[which] = p;
					ctx->bz_up_cap[which] = want;
				} else
					(void) hipGetLastError();
			}
	}
	if (!ctx->sr_layout.load()) { // (split reads: named bytes are brought up ahead, not inflated ahead -- no spare output set)
		const size_t cap_blocks = (n_bytes + n_bytes / 10) / 4096 + 65536;
		const uint64_t cap_out = (uint64_t) ((double) n_bytes * ratio * kSetRoom) + ((uint64_t) 64 << 20);
		(void) (quiet_ensure(ctx->d_bz_blocks2, cap_blocks * sizeof(conga_bgzf_block)) && quiet_ensure(ctx->d_bz_off2, cap_blocks * 8)
				&& quiet_ensure(ctx->d_bz_out2, (size_t) cap_out + 64) && quiet_ensure(ctx->d_bz_status2, cap_blocks));
	}
	bz::trace("the pipeline's buffers (the second one for compressed bytes, the spare output set) in %.1f ms, before the first piece goes up",
			std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

void bz_prewarm_start(conga_ctx *ctx, size_t n_bytes)
{
	if (ctx->bz_prewarmed || !(ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) || ctx->knobs.bgzf_no_inflate_ahead)
		return;
	ctx->bz_prewarmed = true;
	double ratio;
	{
		std::lock_guard<std::mutex> g(ctx->sched.mu);
		ratio = ctx->sched.ratio;
	}
	const int device = ctx->device;
	std::lock_guard<std::mutex> g(ctx->prewarm_mu);
	ctx->bz_prewarm = std::thread([ctx, device, n_bytes, ratio] {
		if (hipSetDevice(device) != hipSuccess)
			return;
		const size_t want = n_bytes + n_bytes / 16 + 512;
		if (ctx->bz_up_cap[1] < want) {
			uint8_t *p = nullptr;
			if (hipMalloc((void **) &p, want) == hipSuccess) {
				if (ctx->bz_up_buf[1])
					(void) hipFree(ctx->bz_up_buf[1]);
				ctx->bz_up_buf[1] = p;
				ctx->bz_up_cap[1] = want;
			} else
				(void) hipGetLastError();
		}
		if (ctx->sr_layout.load()) // (split reads: named bytes are brought up ahead, not inflated ahead -- no spare output set)
			return;
		// (a job asks for room for 1.25 x the largest ratio seen x ITS bytes: the set holds a tenth more than the first input would
		// ask for, so that a sample a few per cent larger than the first -- another individual at the same depth -- does not make it grow)
		const size_t cap_blocks = (n_bytes + n_bytes / 10) / 4096 + 65536;
		const uint64_t cap_out = (uint64_t) ((double) n_bytes * ratio * kSetRoom) + ((uint64_t) 64 << 20);
		(void) (quiet_ensure(ctx->d_bz_blocks2, cap_blocks * sizeof(conga_bgzf_block)) && quiet_ensure(ctx->d_bz_off2, cap_blocks * 8)
				&& quiet_ensure(ctx->d_bz_out2, (size_t) cap_out + 64) && quiet_ensure(ctx->d_bz_status2, cap_blocks));
	});
}

// (before anything else touches what it allocates: a named job's start, the inflate ahead, the context's end)
void bz_prewarm_join(conga_ctx *ctx)
{
	std::thread t;
	{
		std::lock_guard<std::mutex> g(ctx->prewarm_mu);
		t.swap(ctx->bz_prewarm);
	}
	if (t.joinable())
		t.join();
}

// a device buffer of the spare set, grown without a word to the context (this runs beside the caller's thread)
bool quiet_ensure(DevBuf &b, size_t bytes)
{
	if (bytes <= b.cap)
		return true;
	size_t want = std::max(bytes, b.cap + b.cap / 16);
	want = (want + 255) & ~(size_t) 255;
	void *np = nullptr;
	if (hipMalloc(&np, want) != hipSuccess) {
		(void) hipGetLastError();
		return false;
	}
	if (b.p)
		(void) hipFree(b.p); // (nothing uses the spare set: the call that swapped it out has returned behind its walks)
	b.p = np;
	b.cap = want;
	return true;
}

// the spare output set, held back by the call that swapped it in (see there), goes to the next named job that waits for it
void hand_spare_on(conga_ctx *ctx)
{
	// (conga_release_staging may run on a thread of the caller's beside conga_chrom_compute -- the executable does that behind its
	// last BAM --, and both hand the set on: the pointer changes hands under a lock)
	std::shared_ptr<bz::Job> held;
	{
		std::lock_guard<std::mutex> g(ctx->spare_mu);
		held.swap(ctx->spare_held);
	}
	if (held) {
		bz::trace("the spare output set is handed on (behind job %llu)", (unsigned long long) held->ticket);
		ctx->sched.spare_free(held);
	}
}

// *inflated: the bytes named ahead came with their block table and are inflated (the launches are enqueued) in what is now the
// context's output set -- the caller goes straight to its walks
int upload_and_inflate_overlapped(conga_ctx *ctx, const ByteSource &src, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		uint64_t base)
{
	const bool timing = ctx->knobs.timing;
	const auto t0 = std::chrono::steady_clock::now();
	auto ms_since = [](std::chrono::steady_clock::time_point t) {
		return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
	};
	bz::Scheduler &sched = ctx->sched;
	hand_spare_on(ctx); // (a call without a compute behind it -- a sample that goes up chromosome by chromosome)
	if (!ctx->h_bz_ring && !ctx->bz_ring_failed) {
		sched.quiesce(false); // (nothing of ours is in the ring: it is not there)
		make_bz_ring(ctx);
	}
	if (ctx->bz_ring_failed || !ctx->h_bz_ring)
		return fail(ctx, CONGA_ERR_NOMEM, "conga_reads_bgzf: no pinned staging ring");
	TRY(ensure_x2n(ctx));
	// the third launch stream, if the thread the first call left behind has made it (before anything below counts streams)
	if (ctx->bz_shared && ctx->n_bz_streams == 2 && ctx->bz_third_ready.load(std::memory_order_acquire) && bz_streams_wanted() > 2) {
		if (ctx->bz_third_maker.joinable())
			ctx->bz_third_maker.join();
		ctx->bz_kernel[2] = ctx->bz_third;
		ctx->ev_bz_kernel[2] = ctx->ev_bz_third;
		ctx->n_bz_streams = 3;
	}
	const double ms_ring = ms_since(t0);
	// The bytes: already on their way when conga_reads_bgzf_next_fd named exactly these, otherwise a job of this call's own
	// (bz_sched.h: adopt -- the scheduler is `in a call` from here until reads_bgzf_from returns)
	bool ahead = false;
	bz::trace("call: begins (%zu bytes)", n_bytes);
	std::shared_ptr<bz::Job> job = sched.adopt(src, n_bytes, &ahead);
	const double ms_head_start = ahead ? ms_since(job->t_queued) : 0.0;
	if ((ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) && !ctx->bz_prewarmed) {
		{ // (once this call's job has taken its buffer: the thread below allocates the other one)
			std::unique_lock<std::mutex> lk(job->mu);
			job->cv.wait(lk, [&] { return job->started || job->failed || job->done; });
		}
		bz_prewarm_start(ctx, n_bytes);
	}
	// Named ahead WITH the block table: the inflates are launched (or being launched) into the spare output set by the job's own
	// thread.  When that went well and the table is this call's, the sets change places and nothing is left to launch.
	bool inflated_ahead = false;
	{
		const int took = sched.take_inflated(job, blocks, n_blocks, base == 0);
		bz::trace("call: job %llu taken up (%s)", (unsigned long long) job->ticket, took > 0 ? "inflated ahead: the output sets change places" : took < 0 ? "its launches ahead failed" : "to be launched here");
		if (took > 0) {
			std::swap(ctx->d_bz_out, ctx->d_bz_out2);
			std::swap(ctx->d_bz_blocks, ctx->d_bz_blocks2);
			std::swap(ctx->d_bz_off, ctx->d_bz_off2);
			std::swap(ctx->d_bz_status, ctx->d_bz_status2);
			for (int k = 0; k < 2; k++)
				HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_bz_ahead[k], 0));
			inflated_ahead = true;
		} else if (took < 0) // (whatever it launched writes the spare set: let it finish before that set is used again)
			ctx->machine->ahead_drain();
		// The set that was swapped out is the next named job's to fill -- and its inflates are a sample's worth of launches (30 ms of
		// a 1x genome) that go to the GPU the moment the set is free.  This sample's record walks and its compute are a tenth of
		// that and not enqueued yet: handed the set now, the next sample's launches would be in front of them in the GPU's queues
		// and this call (or the compute behind it) would wait a whole inflate for 3 ms of work -- round 3's "walks + checks 30 ms"
		// mode (profiles/r03m_cohort_1x_hw_queues.log).  The set is handed on when this sample's compute is enqueued
		// (conga_chrom_compute), or when the next call of this kind begins: order in the queues, not priorities, keeps the walks in
		// front.
		if (took > 0) {
			std::lock_guard<std::mutex> g(ctx->spare_mu);
			ctx->spare_held = job;
		}
		else if (took < 0)
			sched.spare_free(job);
		else if (ctx->machine->ahead_possible() && sched.hold_spare(job)) { // (bz_sched.h: hold_spare)
			std::lock_guard<std::mutex> g(ctx->spare_mu);
			ctx->spare_held = job;
		}
	}
	sched.enqueue_later();
	if (inflated_ahead) {
		{
			std::unique_lock<std::mutex> lk(job->mu);
			job->cv.wait(lk, [&] { return job->done; });
		}
		if (timing)
			fprintf(stderr, "\n[timing] overlapped upload: named ahead with its block table %d ms before this call: %zu pieces by %d threads enqueued after "
					"%.1f ms (threads: %.1f ms copying, %.1f ms waiting for a free slot, each), %d inflate launches made ahead (the first one: %zu blocks; their "
					"thread: %.1f ms)\n",
					(int) ms_head_start, job->n_pieces, job->n_threads, job->ms_enqueued, job->ms_copy, job->ms_wait, job->launches_ahead,
					job->first_launch_blocks, job->ms_inflate_ahead);
		ctx->bz_in_now = job->d_bytes;
		sched.job_kept = job;
		return CONGA_OK;
	}
	// Launch size: a launch lasts at least one block's 4.4 ms and the launches of a stream follow one another, so with two
	// launch streams (the first call of a context, make_bz_ring) 128 MB per launch -- 3 440 blocks, 42 % of the waves the
	// machine holds -- left it half empty: 97 ms for the stage against 86-93 with 256 MB (32 MB: 248 ms, 64: 143, 384: 96);
	// with three streams 128 MB fill it.  (CONGA_BGZF_LAUNCH_MB: measurement switch, in batches of 128 MB.)
	size_t batches_per_launch = job->piece < kBzPiece ? 1 : (ctx->n_bz_streams >= 3 ? 1 : 2);
	if (ctx->knobs.bgzf_launch_mb > 0)
		batches_per_launch = std::max<size_t>(1, (size_t) ctx->knobs.bgzf_launch_mb / 128);
	// everything enqueued on ctx->stream so far (the block table, buffers grown) comes before the launches
	HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
	for (int k = 0; k < ctx->n_bz_streams; k++)
		HIP_TRY(ctx, hipStreamWaitEvent(ctx->bz_kernel[k], ctx->ev_fork, 0));

	int rc = CONGA_OK;
	size_t b_done = 0; // blocks launched so far
	int launches = 0;
	for (size_t batch = 0; batch < job->n_batches && rc == CONGA_OK;) {
		const size_t last_batch = std::min(job->n_batches, batch + batches_per_launch) - 1;
		{
			std::unique_lock<std::mutex> lk(job->mu);
			job->cv.wait(lk, [&] { return job->failed || job->batches_ready > last_batch; });
			if (job->failed)
				rc = job->short_read ? fail(ctx, CONGA_ERR_DATA, job->src.io_error.load() ? "conga_reads_bgzf: reading the file failed (pread)"
								: "conga_reads_bgzf: the file ends inside the piece that was named")
						: fail(ctx, CONGA_ERR_HIP, "conga_reads_bgzf: the upload failed: " + job->error);
		}
		if (rc != CONGA_OK)
			break;
		const bool last = last_batch + 1 == job->n_batches;
		const size_t have = std::min(n_bytes, (last_batch + 1) * job->pieces_per_batch * job->piece);
		size_t b1 = b_done; // the blocks that are complete with the bytes up to here
		while (b1 < n_blocks && blocks[b1].data_off + blocks[b1].data_len <= have)
			b1++;
		if (last)
			b1 = n_blocks;
		if (b1 > b_done) {
			hipStream_t ks = ctx->bz_kernel[launches % ctx->n_bz_streams];
			const hipError_t e = hipStreamWaitEvent(ks, (hipEvent_t) job->ev_batch[last_batch], 0);
			if (e != hipSuccess)
				rc = fail(ctx, CONGA_ERR_HIP, std::string("conga_reads_bgzf: ") + hipGetErrorString(e));
			else if (!ctx->knobs.bgzf_upload_only) // (measurement switch: the copy up alone; the call then fails its checks)
				rc = launch_inflate(ctx, b1 - b_done, 0, ks, b_done, job->d_bytes);
			launches++;
			b_done = b1;
		}
		batch = last_batch + 1;
	}
	if (rc != CONGA_OK)
		sched.abandon(job);
	else { // (every batch is ready: the job is through but for its bookkeeping)
		std::unique_lock<std::mutex> lk(job->mu);
		job->cv.wait(lk, [&] { return job->done; });
	}
	if (timing)
		fprintf(stderr, "\n[timing] overlapped upload: pinned ring + streams %.1f ms, %zu pieces by %d threads enqueued after %.1f ms (threads: %.1f ms "
				"copying, %.1f ms waiting for a free slot, each), %d inflate launches%s\n", ms_ring, job->n_pieces, job->n_threads, job->ms_enqueued,
				job->ms_copy, job->ms_wait, launches,
				ahead ? (", named ahead: on its way " + std::to_string((int) ms_head_start) + " ms before this call").c_str() : "");
	// ctx->stream goes on behind every launch (and behind the last piece's copy, for the case of no launch at all)
	for (int k = 0; k < ctx->n_bz_streams; k++) {
		(void) hipEventRecord(ctx->ev_bz_kernel[k], ctx->bz_kernel[k]);
		(void) hipStreamWaitEvent(ctx->stream, ctx->ev_bz_kernel[k], 0);
	}
	if (rc == CONGA_OK && job->n_batches)
		(void) hipStreamWaitEvent(ctx->stream, (hipEvent_t) job->ev_batch[job->n_batches - 1], 0);
	ctx->bz_in_now = job->d_bytes;
	sched.job_kept = job; // (its events are waited for by work still in flight)
	if (ctx->bz_shared && ctx->n_bz_streams == 2 && !ctx->bz_third_maker.joinable() && !ctx->bz_third_ready.load(std::memory_order_acquire)
			&& bz_streams_wanted() > 2) {
		const int device = ctx->device;
		ctx->bz_third_maker = std::thread([ctx, device] {
			int lo = 0, hi = 0;
			if (hipSetDevice(device) == hipSuccess && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess
					&& hipStreamCreateWithPriority(&ctx->bz_third, hipStreamNonBlocking, lo) == hipSuccess
					&& hipEventCreateWithFlags(&ctx->ev_bz_third, hipEventDisableTiming) == hipSuccess)
				ctx->bz_third_ready.store(true, std::memory_order_release);
		});
	}
	if (rc == CONGA_OK)
		HIP_TRY(ctx, hipGetLastError());
	return rc;
}

} // namespace

namespace {
int reads_bgzf_from(conga_ctx *ctx, const ByteSource &src, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		const conga_bam_segment *segments, size_t n_segments, uint64_t *reads_per_chrom);
}

extern "C" {

int conga_reads_bgzf(conga_ctx *ctx, const uint8_t *bytes, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		const conga_bam_segment *segments, size_t n_segments, uint64_t *reads_per_chrom)
{
	if (!ctx || !bytes)
		return CONGA_ERR_INVALID;
	ByteSource src;
	src.bytes = bytes;
	return reads_bgzf_from(ctx, src, n_bytes, blocks, n_blocks, segments, n_segments, reads_per_chrom);
}

int conga_reads_bgzf_fd(conga_ctx *ctx, int fd, uint64_t file_off, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		const conga_bam_segment *segments, size_t n_segments, uint64_t *reads_per_chrom)
{
	if (!ctx || fd < 0)
		return CONGA_ERR_INVALID;
	ByteSource src;
	src.fd = fd;
	src.file_off = file_off;
	return reads_bgzf_from(ctx, src, n_bytes, blocks, n_blocks, segments, n_segments, reads_per_chrom);
}

int conga_reads_bgzf_next_fd(conga_ctx *ctx, int fd, uint64_t file_off, size_t n_bytes, const uint64_t *known_starts, size_t n_known,
		uint64_t stop_at, uint64_t *ticket)
{
	if (ticket)
		*ticket = 0;
	if (!ctx || fd < 0 || !ticket || (n_known && !known_starts))
		return CONGA_ERR_INVALID;
	for (size_t k = 0; k < n_known; k++)
		if (known_starts[k] >= n_bytes || (k && known_starts[k] <= known_starts[k - 1]))
			return CONGA_ERR_INVALID;
	// (only what the overlapped route would take, and only with the ring in place: this call allocates nothing and touches
	// nothing but the upload thread's queue -- it may come from another thread than the one inside conga_reads_bgzf_fd)
	const bool overlapped = ctx->knobs.bgzf_overlap >= 0 ? ctx->knobs.bgzf_overlap != 0 : n_bytes >= ((size_t) 96 << 20);
	if (lane_kernel_asked(ctx) || !overlapped || n_bytes == 0 || ctx->knobs.bgzf_no_ahead)
		return CONGA_OK;
	*ticket = ctx->sched.name_next(fd, file_off, n_bytes, known_starts, n_known, stop_at,
			ctx->h_bz_ring != nullptr && !ctx->bz_ring_failed && ctx->bz_copy != nullptr);
	return CONGA_OK;
}

int conga_reads_bgzf_next_blocks(conga_ctx *ctx, uint64_t ticket, const conga_bgzf_block *blocks, size_t n_blocks)
{
	if (!ctx || !blocks || n_blocks == 0 || n_blocks > (size_t) 1 << 28)
		return CONGA_ERR_INVALID;
	ctx->sched.bring_table(ticket, blocks, n_blocks, ctx->d_bz_x2n.p != nullptr && ctx->d_bz_crc.p != nullptr);
	return CONGA_OK;
}

int conga_reads_bgzf_next_table(conga_ctx *ctx, uint64_t ticket, const conga_bgzf_block **blocks, size_t *n_blocks)
{
	if (!ctx || !blocks || !n_blocks)
		return CONGA_ERR_INVALID;
	*blocks = nullptr;
	*n_blocks = 0;
	ctx->sched.wait_table(ticket, blocks, n_blocks);
	return CONGA_OK;
}

int conga_reads_bgzf_next_go(conga_ctx *ctx, uint64_t ticket)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	ctx->sched.go(ticket);
	return CONGA_OK;
}

int conga_reads_bgzf_forget(conga_ctx *ctx, uint64_t ticket)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	ctx->sched.forget(ticket);
	return CONGA_OK;
}

} // extern "C"

namespace {

int reads_bgzf_from(conga_ctx *ctx, const ByteSource &src, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		const conga_bam_segment *segments, size_t n_segments, uint64_t *reads_per_chrom)
{
	if (!blocks || !segments || n_blocks == 0 || n_segments == 0 || n_blocks > (size_t) 1 << 28 || n_segments > (size_t) 1 << 28)
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty())
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_bgzf: no chromosome open");
	if (ctx->staging_cur >= 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_bgzf: a staging buffer is handed out and not committed");
	bz::trace("call: entered");
	const int n_chrom = (int) ctx->slots.size();
	const int first_chrom = segments[0].chrom;
	if (first_chrom < 0 || first_chrom >= n_chrom)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_bgzf: no such chromosome");
	for (int c = first_chrom; c < n_chrom; c++)
		if (ctx->slots[(size_t) c].n_reads != 0)
			return fail(ctx, CONGA_ERR_INVALID, "conga_reads_bgzf: a chromosome from the first named one on already has reads");
	// Split reads (a chromosome named here has a reference sequence): the walk also notes where every kept record starts, and
	// the split-read launch reads pos / qual / flag / l_qseq, the packed sequence and the qualities (split_read.c:206-354)
	// where they lie in the inflated stream -- the records never exist on the host.  The stream then has to outlive this
	// call, so it goes behind what earlier calls left for their chromosomes.
	bool want_rec = false;
	for (int c = first_chrom; c < n_chrom; c++)
		want_rec = want_rec || !ctx->slots[(size_t) c].ref.empty();
	const uint64_t base = ctx->bz_keep_bytes;
	// the inflated stream: the blocks' payloads one behind the other
	std::vector<uint64_t> out_off(n_blocks);
	uint64_t total = 0;
	for (size_t b = 0; b < n_blocks; b++) {
		const conga_bgzf_block &bl = blocks[b];
		if (bl.data_off > n_bytes || (uint64_t) bl.data_len > n_bytes - bl.data_off || bl.inflated_len == 0 || bl.inflated_len > 65536u)
			return fail(ctx, CONGA_ERR_INVALID, "conga_reads_bgzf: block outside the byte range, empty or larger than 64 KiB");
		out_off[b] = base + total;
		total += bl.inflated_len;
	}
	for (size_t k = 0; k < n_segments; k++) {
		const conga_bam_segment &sg = segments[k];
		const bool same = k && sg.chrom == segments[k - 1].chrom;
		if (sg.start > total || sg.pos_lo > sg.pos_hi || sg.chrom < first_chrom || sg.chrom >= n_chrom || sg.ref_id < 0
				|| (k && sg.chrom < segments[k - 1].chrom)
				|| (same && (sg.pos_lo != segments[k - 1].pos_hi || sg.start < segments[k - 1].start || sg.ref_id != segments[k - 1].ref_id))
				|| (!same && sg.pos_lo != 0) || (int64_t) sg.pos_hi > ctx->slots[(size_t) sg.chrom].L)
			return fail(ctx, CONGA_ERR_INVALID, "conga_reads_bgzf: segments must be grouped by chromosome, ordered, and tile each one");
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	// (the pinned block of the walks' results, below: made here, before anything of this call is on the device -- an allocation
	// beside running launches waits for them)
	const size_t r8 = (n_segments + 7) & ~(size_t) 7;
	{
		const size_t walk_bytes = r8 * (8 + 8 + 8 + 4 + 1) + ((n_blocks + 7) & ~(size_t) 7) + 64;
		if (walk_bytes > ctx->h_walk_cap) {
			if (ctx->h_walk)
				(void) hipHostFree(ctx->h_walk);
			ctx->h_walk = nullptr;
			ctx->h_walk_cap = 0;
			const size_t cap = walk_bytes + walk_bytes / 4; // (a cohort's samples differ a little)
			HIP_TRY(ctx, hipHostMalloc(&ctx->h_walk, cap, hipHostMallocDefault));
			ctx->h_walk_cap = cap;
		}
	}
	{
		std::lock_guard<std::mutex> g(ctx->sched.mu);
		ctx->sched.ratio = std::max(ctx->sched.ratio, (double) total / (double) std::max<size_t>(n_bytes, 1));
	}
	const bool timing = ctx->knobs.timing;
	const auto t_begin = std::chrono::steady_clock::now();
	auto ms_since = [](std::chrono::steady_clock::time_point t) {
		return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
	};
	// one decoder scratch (17 KB) per lane; a lane takes several blocks only beyond 131 072 of them (2.2 GB of scratch)
	uint32_t lanes = (uint32_t) std::min<size_t>((n_blocks + 63) & ~(size_t) 63, 131072);
	if (ctx->knobs.bgzf_lanes > 0) // (tests: few lanes, several blocks each)
		lanes = std::min(lanes, (uint32_t) std::max(64, ctx->knobs.bgzf_lanes & ~63));
	// the bytes and their inflate: overlapped (pinned pieces, several launches) for a piece of the file worth it and blocks in
	// file order; otherwise one copy, one launch
	bool in_order = true;
	for (size_t b = 1; b < n_blocks && in_order; b++)
		in_order = blocks[b].data_off >= blocks[b - 1].data_off + blocks[b - 1].data_len;
	// (CONGA_BGZF_OVERLAP = 0 / 1 forces; tests run both forms on small files)
	const bool overlapped = !lane_kernel_asked(ctx) && in_order && (ctx->knobs.bgzf_overlap >= 0 ? ctx->knobs.bgzf_overlap != 0 : n_bytes >= ((size_t) 96 << 20));
	if (!overlapped) // (the overlapped form has device buffers of its own for the compressed bytes: the upload jobs')
		TRY(ensure(ctx, ctx->d_bz_in, n_bytes + 512)); // (the decoders read ahead of their position: up to 64 dwords)
	// A cohort's pipeline swaps this output set with the spare one sample by sample: sized for THIS sample alone it is too small for
	// the bound the next job is given room for (a quarter more than the largest ratio seen, bz_prewarm_start) and was grown -- 6.5 GB
	// allocated and 5.2 GB freed by the job's inflating thread, with every other thread's HIP call waiting behind the runtime's lock:
	// 0.1 - 0.65 s in which a cohort's second or third sample stood still (profiles/r04j_cohort_first_samples.log).  With
	// CONGA_FLAG_EXPECT_COHORT the set is the spare's size from the start.
	size_t room_blocks = n_blocks, room_out = (size_t) (base + total) + 64;
	if ((ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) && overlapped && base == 0 && !ctx->sr_layout.load() && !ctx->knobs.bgzf_no_inflate_ahead
			&& ctx->d_bz_out.cap == 0) { // (the first sizing: later a set is grown for what a sample needs, not for what the next one may)
		double ratio;
		{
			std::lock_guard<std::mutex> g(ctx->sched.mu);
			ratio = ctx->sched.ratio;
		}
		room_blocks = std::max(room_blocks, (n_bytes + n_bytes / 10) / 4096 + 65536);
		room_out = std::max(room_out, (size_t) ((double) n_bytes * ratio * kSetRoom) + ((size_t) 64 << 20) + 64);
	}
	if (overlapped && base == 0) {
		double ratio;
		{
			std::lock_guard<std::mutex> g(ctx->sched.mu);
			ratio = ctx->sched.ratio;
		}
		bz_prealloc(ctx, n_bytes, ratio); // (the device is idle: see there)
	}
	TRY(ensure(ctx, ctx->d_bz_blocks, room_blocks * sizeof(conga_bgzf_block)));
	TRY(ensure(ctx, ctx->d_bz_off, room_blocks * 8));
	TRY(ensure(ctx, ctx->d_bz_out, room_out, base > 0));
	TRY(ensure(ctx, ctx->d_bz_status, room_blocks));
	// (a cohort: room for a sample with an eighth more start points than the first -- growing beside the pipeline's launches waits for them)
	const size_t seg_room = ctx->d_bz_seg.cap == 0 && (ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) ? n_segments + n_segments / 8 : n_segments;
	TRY(ensure(ctx, ctx->d_bz_seg, seg_room * sizeof(conga_bam_segment)));
	TRY(ensure(ctx, ctx->d_bz_cnt, seg_room * 4));
	TRY(ensure(ctx, ctx->d_bz_first, seg_room * 8));
	TRY(ensure(ctx, ctx->d_bz_stop, seg_room * 8));
	TRY(ensure(ctx, ctx->d_bz_bad, seg_room));
	TRY(ensure(ctx, ctx->d_bz_at, seg_room * 8));
	TRY(ensure(ctx, ctx->d_bz_flag, 4));
	TRY(ensure_crc_table(ctx));
	const double ms_buffers = ms_since(t_begin);
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_blocks.p, blocks, n_blocks * sizeof(conga_bgzf_block), hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_off.p, out_off.data(), n_blocks * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_seg.p, segments, n_segments * sizeof(conga_bam_segment), hipMemcpyHostToDevice, st));
	double ms_alloc_upload = 0, ms_inflate = 0;
	auto t_inflate = std::chrono::steady_clock::now();
	std::vector<uint8_t> whole; // (a small piece of a file: read in one go)
	struct InCall { // (bytes named ahead while this call runs go up right behind this call's)
		conga_ctx *c;
		~InCall() { c->sched.end_call(); }
	} in_call{ctx};
	if (overlapped) {
		TRY(upload_and_inflate_overlapped(ctx, src, n_bytes, blocks, n_blocks, base));
	} else {
		const uint8_t *bytes = src.bytes;
		if (!bytes) {
			whole.resize(n_bytes);
			if (!src.fetch(0, whole.data(), n_bytes))
				return fail(ctx, CONGA_ERR_DATA, "conga_reads_bgzf: the file ends inside the piece that was named");
			bytes = whole.data();
		}
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_in.p, bytes, n_bytes, hipMemcpyHostToDevice, st));
		if (!src.bytes)
			HIP_TRY(ctx, hipStreamSynchronize(st)); // (`whole` must outlive the copy)
		if (timing) {
			HIP_TRY(ctx, hipStreamSynchronize(st));
			ms_alloc_upload = ms_since(t_begin);
			t_inflate = std::chrono::steady_clock::now();
		}
		TRY(launch_inflate(ctx, n_blocks, lanes));
	}
	BamWalkArgs w;
	w.stream = ptr<uint8_t>(ctx->d_bz_out) + base;
	w.stream_len = total;
	w.rec_off = nullptr;
	w.rec_base = base;
	w.check_body = want_rec ? 1u : 0u;
	w.segments = ptr<conga_bam_segment>(ctx->d_bz_seg);
	w.n_segments = (uint32_t) n_segments;
	w.count = ptr<uint32_t>(ctx->d_bz_cnt);
	w.v_first = ptr<uint64_t>(ctx->d_bz_first);
	w.v_stop = ptr<uint64_t>(ctx->d_bz_stop);
	w.bad = ptr<uint8_t>(ctx->d_bz_bad);
	w.write_at = ptr<uint64_t>(ctx->d_bz_at);
	w.pos = nullptr;
	w.mapq = nullptr;
	const int wgrid = (int) ((n_segments + 63) / 64);
	if (timing) {
		HIP_TRY(ctx, hipStreamSynchronize(st));
		ms_inflate = ms_since(t_inflate);
	}
	const auto t_walk = std::chrono::steady_clock::now();
	hipLaunchKernelGGL(bam_walk_kernel<false>, dim3(wgrid), dim3(64), 0, st, w);
	// What the counting walk found comes down into ONE pinned block of the context's and the places of the writing walk go up out
	// of it: five copies into vectors made on the spot (pageable: each staged through the runtime's own buffer) and one back were
	// most of the 0.7-0.9 ms between the two walks of a 1x genome.
	uint64_t *const v_first = static_cast<uint64_t *>(ctx->h_walk), *const v_stop = v_first + r8, *const write_at = v_stop + r8;
	uint32_t *const count = reinterpret_cast<uint32_t *>(write_at + r8);
	uint8_t *const bad = reinterpret_cast<uint8_t *>(count + r8), *const status = bad + r8;
	HIP_TRY(ctx, hipMemcpyAsync(status, ctx->d_bz_status.p, n_blocks, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(bad, ctx->d_bz_bad.p, n_segments, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(count, ctx->d_bz_cnt.p, n_segments * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(v_first, ctx->d_bz_first.p, n_segments * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(v_stop, ctx->d_bz_stop.p, n_segments * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	for (size_t b = 0; b < n_blocks; b++)
		if (status[b] != kBgzfOk)
			return fail(ctx, CONGA_ERR_DATA, status[b] == kBgzfCrc ? "conga_reads_bgzf: a block fails its CRC32"
					: "conga_reads_bgzf: a block does not inflate to its recorded size");
	std::vector<int64_t> per_chrom((size_t) n_chrom, 0);
	uint64_t n_new = 0;
	for (size_t k = 0; k < n_segments; k++) {
		if (bad[k])
			return fail(ctx, CONGA_ERR_DATA, bad[k] == 2 ? "conga_reads_bgzf: the records of a target are not in position order"
					: "conga_reads_bgzf: a start point does not lead along whole BAM records");
		// the record that ends a segment is the next segment's first -- inside a chromosome, and from a target to the target
		// that follows it in the file (whatever ends target t is the first record behind it: the first of target t + 1 if that
		// one has records, and what ends that one's empty walk if it has none)
		if (k + 1 < n_segments && (segments[k + 1].chrom == segments[k].chrom || segments[k + 1].ref_id == segments[k].ref_id + 1)
				&& v_stop[k] != v_first[k + 1])
			return fail(ctx, CONGA_ERR_DATA, "conga_reads_bgzf: the start points do not line up with the records");
		write_at[k] = (uint64_t) ctx->n_reads_total + n_new;
		n_new += count[k];
		per_chrom[(size_t) segments[k].chrom] += count[k];
	}
	if ((uint64_t) ctx->n_reads_total + n_new >= 0xFFFFFFF0ull)
		return fail(ctx, CONGA_ERR_RANGE, "conga_reads_bgzf: more than 2^32 reads in one context");
	if (n_new) {
		const size_t total_reads = (size_t) ctx->n_reads_total + (size_t) n_new;
		if (total_reads * 4 > ctx->d_pos.cap || total_reads > ctx->d_mapq.cap) {
			// (a cohort's samples differ a little: the next one must not make these grow beside the pipeline's launches -- bz_prealloc)
			const size_t want = std::max(total_reads + ((ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) ? total_reads / 8 : 0), (size_t) 1 << 22);
			TRY(ensure(ctx, ctx->d_pos, want * 4, true));
			TRY(ensure(ctx, ctx->d_mapq, want, true));
		}
		if (want_rec) {
			TRY(ensure(ctx, ctx->d_sr_recoff, std::max(total_reads, (size_t) 1 << 22) * 8, true));
			w.rec_off = ptr<uint64_t>(ctx->d_sr_recoff);
		}
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_at.p, write_at, n_segments * 8, hipMemcpyHostToDevice, st));
		w.pos = ptr<int32_t>(ctx->d_pos);
		w.mapq = ptr<uint8_t>(ctx->d_mapq);
		hipLaunchKernelGGL(bam_walk_kernel<true>, dim3(wgrid), dim3(64), 0, st, w);
		// more than 32767 read starts on one base would wrap the reference's `short`: only the dense formulation
		// reproduces that (same guard as note_equal_runs, exact here)
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_bz_flag.p, 0, 4, st));
		if (n_new >= 32768) {
			const int egrid = (int) ((n_new + 255) / 256);
			hipLaunchKernelGGL(equal_run_kernel, dim3(egrid), dim3(256), 0, st, ptr<int32_t>(ctx->d_pos) + ctx->n_reads_total, n_new, 32768u,
					ptr<uint32_t>(ctx->d_bz_flag));
		}
		uint32_t flag = 0;
		HIP_TRY(ctx, hipMemcpyAsync(&flag, ctx->d_bz_flag.p, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(ctx, hipStreamSynchronize(st));
		if (flag)
			ctx->wrap_risk = true;
	}
	bz::trace("call: the walks are through");
	if (timing)
		fprintf(stderr, "\n[timing] conga_reads_bgzf: %zu blocks, %.1f MB -> %.1f MB, %zu start points, %llu reads: %s %.1f ms, "
				"%s %.1f ms, walks + checks %.1f ms\n", n_blocks, n_bytes / 1e6, total / 1e6, n_segments, (unsigned long long) n_new,
				overlapped ? "buffers" : "buffers + upload", overlapped ? ms_buffers : ms_alloc_upload,
				overlapped ? "upload + inflate (overlapped)" : "inflate", ms_inflate, ms_since(t_walk));
	// the tuples of a context lie in chromosome order: every chromosome from the first named one on gets its place
	{
		int64_t at = ctx->n_reads_total;
		for (int c = first_chrom; c < n_chrom; c++) {
			HostSlot &hc = ctx->slots[(size_t) c];
			hc.read_off = at;
			hc.n_reads = per_chrom[(size_t) c];
			hc.device_fed = hc.device_fed || per_chrom[(size_t) c] > 0;
			if (!hc.ref.empty()) { // its split-read records are its tuples' records, in place
				hc.sr_inplace = true;
				hc.sr_off = at;
				hc.n_sr = hc.n_reads;
			}
			at += hc.n_reads;
		}
	}
	if (want_rec)
		ctx->bz_keep_bytes = (base + total + 255) & ~(uint64_t) 255;
	ctx->n_reads_total += (int64_t) n_new;
	ctx->sample_dirty = true;
	ctx->computed = false;
	if (reads_per_chrom)
		for (int c = 0; c < n_chrom; c++)
			reads_per_chrom[c] = (uint64_t) ctx->slots[(size_t) c].n_reads;
	return CONGA_OK;
}

} // namespace

extern "C" {

int conga_inflate_blocks(conga_ctx *ctx, const uint8_t *bytes, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		uint8_t *out, size_t out_bytes, uint8_t *status, double *kernel_ms)
{
	if (!ctx || !bytes || !blocks || !status || n_blocks == 0 || n_blocks > (size_t) 1 << 28)
		return CONGA_ERR_INVALID;
	std::vector<uint64_t> out_off(n_blocks);
	uint64_t total = 0;
	for (size_t b = 0; b < n_blocks; b++) {
		const conga_bgzf_block &bl = blocks[b];
		if (bl.data_off > n_bytes || (uint64_t) bl.data_len > n_bytes - bl.data_off || bl.inflated_len == 0 || bl.inflated_len > 65536u)
			return fail(ctx, CONGA_ERR_INVALID, "conga_inflate_blocks: block outside the byte range, empty or larger than 64 KiB");
		out_off[b] = total;
		total += bl.inflated_len;
	}
	if (out && out_bytes < total)
		return fail(ctx, CONGA_ERR_INVALID, "conga_inflate_blocks: output buffer too small");
	if (ctx->bz_keep_bytes)
		return fail(ctx, CONGA_ERR_INVALID, "conga_inflate_blocks: the context holds BAM records in place");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	hipStream_t st = ctx->stream;
	uint32_t lanes = (uint32_t) std::min<size_t>((n_blocks + 63) & ~(size_t) 63, 131072);
	TRY(ensure(ctx, ctx->d_bz_in, n_bytes + 512));
	TRY(ensure(ctx, ctx->d_bz_blocks, n_blocks * sizeof(conga_bgzf_block)));
	TRY(ensure(ctx, ctx->d_bz_off, n_blocks * 8));
	TRY(ensure(ctx, ctx->d_bz_out, (size_t) total + 16));
	TRY(ensure(ctx, ctx->d_bz_status, n_blocks));
	TRY(ensure_crc_table(ctx));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_in.p, bytes, n_bytes, hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_blocks.p, blocks, n_blocks * sizeof(conga_bgzf_block), hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_off.p, out_off.data(), n_blocks * 8, hipMemcpyHostToDevice, st));
	HIP_TRY(ctx, hipMemsetAsync(ctx->d_bz_status.p, 0xFF, n_blocks, st));
	HIP_TRY(ctx, hipEventRecord(ctx->ev_k0[0], st));
	TRY(launch_inflate(ctx, n_blocks, lanes));
	HIP_TRY(ctx, hipEventRecord(ctx->ev_k1[0], st));
	HIP_TRY(ctx, hipMemcpyAsync(status, ctx->d_bz_status.p, n_blocks, hipMemcpyDeviceToHost, st));
	if (out)
		HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_bz_out.p, (size_t) total, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	HIP_TRY(ctx, hipGetLastError());
	if (kernel_ms) {
		float ms = 0.0f;
		HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_k0[0], ctx->ev_k1[0]));
		*kernel_ms = ms;
	}
	return CONGA_OK;
}

} // extern "C"
