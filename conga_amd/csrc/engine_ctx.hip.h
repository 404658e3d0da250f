// engine_ctx.hip.h -- the engine's state: device buffers, the host side of a chromosome, the context (part of conga_api.hip's
// one translation unit; see there).  Included first: everything else works on a conga_ctx.
#pragma once

namespace {

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
};

constexpr size_t kStagingTuples = (size_t) 1 << 22; // per pinned buffer
constexpr int kStagingRing = 2;

struct Staging {
	int32_t *pos = nullptr;
	uint8_t *mapq = nullptr;
	hipEvent_t copied = nullptr; // H2D of the last commit from this buffer
	bool in_flight = false;
};

// Host side of one chromosome.
struct HostSlot {
	int64_t L = 0, n_win = 0, n_tiles = 0;
	int64_t read_off = 0, n_reads = 0;
	bool device_fed = false; // its tuples came from conga_reads_bgzf (not to be mixed with conga_reads_commit)
	int32_t tail_val = 0;  // position of the last committed tuple and the length of the run of equal
	int64_t tail_len = 0;  // positions that ends there (capped): see wrap_risk
	std::vector<uint8_t> gc_hist, gc_like; // gc_like empty = same as gc_hist
	std::vector<int32_t> iv_start[2], iv_end[2], iv_support[2]; // [0] = dels, [1] = dups
	bool has_map = false, map_sorted = false;
	std::vector<int32_t> map_start, map_end;
	std::vector<float> map_val;
	// split-read inputs
	std::vector<uint8_t> ref;               // upper-cased chromosome sequence
	std::vector<int32_t> sat_start, sat_end; // sorted, disjoint
	int64_t sr_off = 0, n_sr = 0;           // this chromosome's records in the split-read arrays (in place: in d_sr_recoff = its tuples' indices)
	bool sr_inplace = false;                // its records lie in the inflated BAM stream of conga_reads_bgzf
	int64_t refn_off = 0, kpos_off = 0, sat_off = 0; // where its packed reference / 10-mer index / satellites lie (prepare_layout)
	int64_t pres_off = 0;                           // ... and its {solo, echo} bits (in uint2: 32 positions each)
	int kidx = -1;                          // its offset table; -1: no reference, no part in the split-read launch
	uint64_t ref_version = 0;               // stamps every conga_reference(): the 10-mer index is rebuilt only for new text
	// filled by prepare()
	int64_t rd_off = 0, gc_off = 0, tile0 = 0, tidx_off = 0, iv0 = 0, map_row_off = 0, row_tile_off = 0;
};

} // namespace

struct conga_ctx {
	int device = 0;
	int n_cu = 256;
	int depth_blocks_per_cu = 8; // resident depth_tile workgroups per CU (occupancy query)
	int tuple_blocks_per_cu = 8; // resident workgroups per CU of the tuple pass: its grid is exactly one resident wave of them
	int split_blocks_per_cu = 8; // ... and of the split-read launch, whose workgroups take work units round robin
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr; // runs interval_reduce beside the float chain (both are latency-bound)
	hipEvent_t ev_fork = nullptr, ev_fork2 = nullptr, ev_counted = nullptr, ev_join = nullptr;
	conga_opts opts{};
	std::string err;

	int32_t step = 100, tile_len = 0;
	std::vector<HostSlot> slots;
	int cur = -1; // selected slot
	bool layout_dirty = true; // chromosomes, GC arrays, intervals, tracks or split-read inputs changed: prepare_layout()
	bool sample_dirty = true; // only the read tuples changed (another sample behind the same layout): prepare_sample()
	int read_target = -1;     // chromosome conga_reads_commit() appends to; -1: the one begun last

	// reads
	int64_t n_reads_total = 0;
	// d_small holds TWO accumulator arenas, each Small[n_slots] (padded to small_bytes) followed by int32 observed[n_iv]
	// (arena_bytes in all).  A compute adds into arena `small_cur`; the chain launch clears the other one on the
	// side, so the next compute starts on a zeroed arena without a memset launch of its own.
	size_t small_bytes = 0, arena_bytes = 0;
	uint32_t tuple_chunks = 0, tuple_chunks_per_block = 1; // geometry of the tuple pass (prepare)
	int small_cur = 0, small_cur_next = 0;
	bool arena_zeroed[2] = {false, false};
	bool layout_dense = false;   // formulation prepare_layout() laid the tracks out for (track_painted depends on it)
	bool wrap_risk = false;      // some position may hold more than 32767 reads: only the dense kernels reproduce the `short` wrap
	bool depth_resident = false; // read_depth[] of the last compute is in d_rd
	Staging staging[kStagingRing];
	int staging_next = 0; // buffer the next conga_reads_staging() hands out
	int staging_cur = -1; // buffer handed out and not yet committed

	// layout totals (prepare)
	int64_t total_L = 0, total_tiles = 0, total_gc = 0, n_iv = 0, n_items = 0, n_chain_x = 0, n_chain_a = 0, n_chain_b = 0, n_depth_blocks = 0;
	bool gc_like_distinct = false, any_map = false, support_given = false;
	bool any_ref = false;                     // some chromosome has a reference sequence (per layout): the support column exists
	bool any_sr = false;                      // ... and split-read records (per sample): the split-read launch runs
	int n_sr_slots = 0;                       // chromosomes with split-read records and a reference (one SplitSlot each)
	uint32_t sr_units = 0;                    // work units of the split-read launch
	int64_t refn_words = 0, kpos_total = 0, sat_total = 0; // layout totals of the split-read inputs
	int64_t pres_words = 0;
	bool pres_built = false;      // the solo / echo bits of the resident indexes are made (enqueue_compute: before the second split-read launch)
	int sr_launches_on_index = 0; // split-read launches since the indexes were built
	uint64_t bz_keep_bytes = 0;               // bytes of d_bz_out that hold records in place: the next conga_reads_bgzf goes behind them
	uint64_t ref_stamp = 0;                   // source of HostSlot::ref_version
	std::vector<uint64_t> index_sig;          // what the resident 10-mer indexes were built from (slot, length, version)
	bool any_map_painted = false; // some chromosome's track is painted into d_map by compute (dense formulation / unsorted rows)
	bool any_map_rows = false;    // some chromosome's track is summed in row space (sorted rows, tuple-space formulation)
	int64_t n_sr_total = 0, sr_bytes_total = 0;
	conga_split_staging sr_stage{}; // one pinned set (the split-read path is not the bench line)
	bool sr_staged = false;

	// device buffers
	DevBuf d_head; // per sample: [Slot table | TupleBlockHome table], one upload from h_head; d_slots / d_block_home point into it
	void *h_head = nullptr;
	size_t h_head_cap = 0;
	hipEvent_t ev_head = nullptr; // the upload from h_head
	bool head_in_flight = false;
	DevBuf d_pos, d_mapq, d_tile_start, d_small_scratch, d_item_slot, d_item_row0, d_item_row1, d_item_rt_off, d_block_home, d_item_lo, d_rd, d_gc_hist, d_gc_like, d_slots, d_small, d_map, d_winner,
			d_map_start, d_map_end, d_map_val, d_iv_start, d_iv_end, d_iv_type, d_iv_slot, d_iv_has_map, d_order,
			d_expected, d_item_off, d_item_len, d_item_iv, d_item_has_map, d_item_first, d_map_part,
			d_support, d_results, d_bases, d_row_tile, d_depth_blocks, d_support_base, d_ref, d_sat_start, d_sat_end, d_sr_pos,
			d_sr_mapq, d_sr_flag, d_sr_lq, d_sr_off, d_sr_data, d_sr_recoff, d_refn, d_kmer_keys, d_kmer_sorted, d_kmer_tmp, d_kmer_offset, d_kmer_pos, d_kmer_pres, d_sr_slots,
			// conga_reads_bgzf: compressed blocks, their table, the inflated stream, the decoders' scratch, the walk's per-segment results
			d_bz_in, d_bz_blocks, d_bz_off, d_bz_out, d_bz_status, d_bz_scratch, d_bz_crc, d_bz_seg, d_bz_cnt, d_bz_first, d_bz_stop,
			d_bz_bad, d_bz_at, d_bz_flag, d_bz_x2n, d_bz_ticket,
			// the spare output set: bytes named ahead WITH their block table (conga_reads_bgzf_next_blocks) are inflated into it
			// while the sample in front is still walked and computed; the call that takes them up swaps the sets
			d_bz_out2, d_bz_blocks2, d_bz_off2, d_bz_status2;

	// conga_reads_bgzf: the file's bytes go up through a ring of pinned pieces filled by host threads, inflate launches follow
	uint8_t *h_bz_ring = nullptr;
	hipEvent_t ev_bz_slot[12] = {};
	bool bz_ring_failed = false;
	hipStream_t bz_copy = nullptr, bz_kernel[3] = {};
	hipStream_t bz_copy2 = nullptr;   // (measurement switch CONGA_BGZF_COPY_STREAMS=2: the ring's odd slots go up on a stream of their own)
	hipEvent_t ev_bz_copy2 = nullptr;
	hipEvent_t ev_bz_kernel[3] = {};
	bool bz_shared = false; // the inflate launches go to `stream2` and `stream` (made with the lowest priority for that)
	int n_bz_streams = 0;
	// ... and to a third stream of their own from the second call on: made by a thread that the first call leaves behind
	// (15-20 ms that no caller waits for)
	std::thread bz_third_maker;
	std::atomic<bool> bz_third_ready{false};
	hipStream_t bz_third = nullptr;
	hipEvent_t ev_bz_third = nullptr;
	// The upload is a JOB run by a thread of the scheduler's own (bz_sched.h): conga_reads_bgzf* starts one and launches the
	// inflates behind its batches; conga_reads_bgzf_next_fd queues the NEXT sample's behind it, into the other of two device
	// buffers, so that sample k + 1 is on its way up while sample k is walked, computed and written out.  The scheduler's state
	// (queue, tickets, who owns which buffer) is host code of its own; `machine` is what it asks of the GPU.
	conga::Knobs knobs;
	std::unique_ptr<bz::Machine> machine; // (in front of the scheduler: its jobs give their events back through it when they go)
	bz::Scheduler sched;
	std::atomic<uint32_t> bz_ticket_next{0}; // which of d_bz_ticket's counters the next inflate launch takes
	std::atomic<bool> computed_once{false}; // ev_done has been recorded at least once (the inflate-ahead thread waits for it on ITS streams)
	std::mutex spare_mu; // (spare_held: the caller's thread and a thread that releases the staging may both hand it on)
	std::shared_ptr<bz::Job> spare_held; // the job whose inflated-ahead set the last call swapped in: the set that went out is handed on behind this sample's compute
	std::mutex prewarm_mu;
	std::thread bz_prewarm; // CONGA_FLAG_EXPECT_COHORT: gets the second buffer of compressed bytes and the spare output set while the first sample is on
	bool bz_prewarmed = false;
	std::atomic<bool> sr_layout{false}; // a chromosome has its reference text (conga_reference): split reads will be mapped on the records
	                                    // where the inflate leaves them -- the inflated stream of a sample is in use until its compute is
	                                    // through, so bytes named ahead are only brought up, not inflated ahead (no spare output set)
	uint8_t *bz_up_buf[2] = {nullptr, nullptr};
	size_t bz_up_cap[2] = {0, 0};
	const uint8_t *bz_in_now = nullptr; // the compressed bytes the last overlapped upload brought
	hipStream_t bz_ahead[2] = {nullptr, nullptr}; // launch streams of the inflate ahead (lowest priority), made by its thread
	hipEvent_t ev_bz_ahead[2] = {nullptr, nullptr};

	// pinned read-back
	Small *h_small = nullptr;
	size_t h_small_cap = 0;
	conga_result *h_results = nullptr;
	size_t h_results_cap = 0;
	std::vector<int32_t> order_pos;    // position of interval iv in the chain kernel's processing order
	bool host_results_by_order = false; // h_results of the last compute is laid out in that order (fused scoring)
	bool host_results_valid = false;    // h_results holds the records of the last compute (CONGA_FLAG_RESULTS_ON_DEVICE: not until fetched)

	bool computed = false;
	// conga_sample_reads() is double-buffered: the next sample's tuples go into the OTHER pair of buffers on stream2 while the
	// last compute (which reads d_pos / d_mapq) and its fetch are still under way.  `computed_reads` is what that compute ran
	// on (per chromosome: first tuple, count), for the fetch's statistics and for settle_wrap_risk's second compute.
	DevBuf d_pos_alt, d_mapq_alt;
	// conga_sample_reads_d16: the differences as they came up and the exceptions (one set per pair of tuple buffers: the copy stream
	// carries nothing but copies, back to back), the scan's scratch; what the next compute has to expand first
	DevBuf d_delta[2], d_delta_esc[2], d_delta_agg;
	bool expand_pending = false;
	int expand_width = 16;
	size_t expand_esc_at = (size_t) -1; // the exceptions lie behind the differences at this offset of d_delta (-1: in d_delta_esc)
	uint64_t expand_total = 0;
	size_t expand_n_esc = 0;
	hipEvent_t ev_reads = nullptr;     // the copies of the last conga_sample_reads (on stream2)
	hipEvent_t ev_pair[2] = {};        // the last compute that read buffer pair 0 / 1 (on stream)
	bool used_recorded[2] = {false, false};
	int pos_buf = 0;                   // which pair d_pos / d_mapq currently are
	bool reads_on_stream2 = false;     // the next compute has to wait for ev_reads
	bool reads_ahead = false;          // the HostSlots describe a newer sample than the one last computed
	std::vector<std::pair<int64_t, int64_t>> computed_reads;
	int64_t computed_total = 0;
	hipGraphExec_t graph_exec = nullptr; // the captured step; dropped whenever the layout changes
	bool graph_dense = false;
	int computes_on_layout = 0;          // computes since the layout last changed
	hipEvent_t ev_done = nullptr;      // the end of the LATEST compute (also waited for by the inflate-ahead thread on its streams)
	// conga_chrom_compute_ahead: TWO computes in flight.  What a compute leaves for its fetch -- the pinned read-back blocks, the
	// records in HBM, the event behind its last launch, what it ran on -- exists twice; the set of the compute BEFORE the latest
	// one is the `_prev` half, and the two change places (swap_result_sets) whenever the previous compute's results are asked for
	// or a compute goes ahead.  A small step's time is mostly the host's (the wait's wake-up, the fetch, the next step's launches:
	// 0.27 ms against 0.12 ms of kernels on one chromosome); with the next compute already in the queue the GPU does not wait for it.
	hipEvent_t ev_set = nullptr, ev_set_prev = nullptr; // the end of the compute whose results the set holds
	Small *h_small_prev = nullptr;
	size_t h_small_prev_cap = 0;
	conga_result *h_results_prev = nullptr;
	size_t h_results_prev_cap = 0;
	DevBuf d_results_prev;
	bool prev_by_order = false, prev_valid = false, prev_depth_resident = false;
	std::vector<std::pair<int64_t, int64_t>> prev_computed_reads;
	int64_t prev_computed_total = 0;
	bool have_previous = false;    // the `_prev` half holds the results of the compute before the latest one (same layout)
	bool previous_settled = false; // ... and its wrap guard has been looked at (settle_previous)
	void *h_walk = nullptr;        // pinned: what the counting walk of conga_reads_bgzf* found, and the places of the writing one
	size_t h_walk_cap = 0;
	bool rd_clobbered = false;     // d_rd holds the depth of an OLDER sample than the latest compute's (settle_previous computed it again)
	hipEvent_t ev_k0[CONGA_K_COUNT] = {}, ev_k1[CONGA_K_COUNT] = {};
	bool ev_used[CONGA_K_COUNT] = {};
};

namespace {

int fail(conga_ctx *ctx, int status, const std::string &msg)
{
	if (ctx)
		ctx->err = msg;
	return status;
}

int enqueue_compute(conga_ctx *ctx, bool dense);
int settle_previous(conga_ctx *ctx); // (engine_compute.hip.h: the wrap guard of the compute before the latest one)

// Formulation: tuple / row space unless the arrays were asked for, the reads may be unsorted, or a `short` may wrap.
bool dense_formulation(const conga_ctx *ctx)
{
	return (ctx->opts.flags & (CONGA_FLAG_READS_UNSORTED | CONGA_FLAG_MATERIALIZE_DEPTH)) != 0 || ctx->wrap_risk;
}

// A chromosome's track is painted into mappability[L] (and summed from there) in the dense formulation and whenever
// its rows are not sorted-and-at-most-abutting; otherwise interval_map_rows sums straight from the rows.
bool track_painted(const conga_ctx *ctx, const HostSlot &h)
{
	return h.has_map && (dense_formulation(ctx) || !h.map_sorted);
}

char *arena_of(conga_ctx *ctx, int which)
{
	return static_cast<char *>(ctx->d_small.p) + (size_t) which * ctx->arena_bytes;
}

int32_t *observed_of(conga_ctx *ctx)
{
	return reinterpret_cast<int32_t *>(arena_of(ctx, ctx->small_cur) + ctx->small_bytes);
}

void drop_graph(conga_ctx *ctx)
{
	if (ctx->graph_exec)
		(void) hipGraphExecDestroy(ctx->graph_exec);
	ctx->graph_exec = nullptr;
	ctx->computes_on_layout = 0;
}

#define HIP_TRY(ctx, call)                                                                              \
	do {                                                                                                \
		hipError_t e_ = (call);                                                                         \
		if (e_ != hipSuccess)                                                                           \
			return fail((ctx), (e_ == hipErrorOutOfMemory) ? CONGA_ERR_NOMEM : CONGA_ERR_HIP,          \
					std::string(#call) + ": " + hipGetErrorString(e_));                                \
	} while (0)

#define TRY(expr)                \
	do {                         \
		int rc_ = (expr);        \
		if (rc_ != CONGA_OK)     \
			return rc_;          \
	} while (0)

// Grow a device buffer.  keep = preserve the old contents (device-to-device copy on the stream).
int ensure(conga_ctx *ctx, DevBuf &b, size_t bytes, bool keep = false)
{
	if (bytes <= b.cap)
		return CONGA_OK;
	size_t want = std::max(bytes, b.cap + b.cap / 2);
	want = (want + 255) & ~(size_t) 255;
	void *np = nullptr;
	HIP_TRY(ctx, hipMalloc(&np, want));
	hipError_t e = hipSuccess;
	if (keep && b.p && b.cap)
		e = hipMemcpyAsync(np, b.p, b.cap, hipMemcpyDeviceToDevice, ctx->stream);
	if (e == hipSuccess && b.p)
		e = hipStreamSynchronize(ctx->stream); // nothing in flight may still use the old block
	if (e != hipSuccess) {
		(void) hipFree(np);
		return fail(ctx, CONGA_ERR_HIP, std::string("grow: ") + hipGetErrorString(e));
	}
	if (b.p)
		(void) hipFree(b.p);
	b.p = np;
	b.cap = want;
	return CONGA_OK;
}

template <typename T> T *ptr(const DevBuf &b)
{
	return static_cast<T *>(b.p);
}

int upload(conga_ctx *ctx, DevBuf &b, const void *src, size_t bytes)
{
	TRY(ensure(ctx, b, bytes ? bytes : 1));
	if (bytes)
		HIP_TRY(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
	return CONGA_OK;
}

void free_buf(DevBuf &b)
{
	if (b.p)
		(void) hipFree(b.p);
	b.p = nullptr;
	b.cap = 0;
}

int type_index(char type)
{
	if (type == CONGA_DELETION)
		return 0;
	if (type == CONGA_DUPLICATION)
		return 1;
	return -1;
}

bool batch_mode(const conga_ctx *ctx)
{
	return (ctx->opts.flags & CONGA_FLAG_BATCH) != 0;
}


} // namespace
