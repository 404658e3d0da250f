// conga_api.hip -- C-ABI of include/conga_hip.h over the gfx950 kernels in kernels.hip.h.
//
// One context = one GPU and one HIP stream.  A context holds one chromosome (the reference's
// sequential per-chromosome loop, bam_data.c:269-339) or, with CONGA_FLAG_BATCH, any number of
// them ("slots"): every kernel then covers the whole batch in ONE launch, which is what keeps an
// MI355X busy -- a single chromosome's interval kernels are far too small to fill 256 CUs.
// Everything the kernels need stays resident in HBM, so conga_chrom_compute() can be replayed on
// the same inputs (bench.py times exactly that).
//
// There is deliberately no CPU fallback anywhere in this file: without a HIP device
// conga_create() returns NULL / CONGA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include <unistd.h>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <memory>
#include <new>
#include <numeric>
#include <type_traits>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/conga_hip.h"
#include "kernels.hip.h"
#include "delta16.hip.h"
#include "kernels_bam.hip.h"
#include "kmer_sort.hip.h"
#include "split_map.hip.h"
#include "pack_host.h"

using namespace conga;

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
			d_sr_mapq, d_sr_flag, d_sr_lq, d_sr_off, d_sr_data, d_sr_recoff, d_refn, d_kmer_keys, d_kmer_sorted, d_kmer_tmp, d_kmer_offset, d_kmer_pos, d_sr_slots,
			// conga_reads_bgzf: compressed blocks, their table, the inflated stream, the decoders' scratch, the walk's per-segment results
			d_bz_in, d_bz_blocks, d_bz_off, d_bz_out, d_bz_status, d_bz_scratch, d_bz_crc, d_bz_seg, d_bz_cnt, d_bz_first, d_bz_stop,
			d_bz_bad, d_bz_at, d_bz_flag, d_bz_x2n,
			// the spare output set: bytes named ahead WITH their block table (conga_reads_bgzf_next_blocks) are inflated into it
			// while the sample in front is still walked and computed; the call that takes them up swaps the sets
			d_bz_out2, d_bz_blocks2, d_bz_off2, d_bz_status2;

	// conga_reads_bgzf: the file's bytes go up through a ring of pinned pieces filled by host threads, inflate launches follow
	uint8_t *h_bz_ring = nullptr;
	hipEvent_t ev_bz_slot[12] = {};
	bool bz_ring_failed = false;
	hipStream_t bz_copy = nullptr, bz_kernel[3] = {};
	hipEvent_t ev_bz_kernel[3] = {};
	bool bz_shared = false; // the inflate launches go to `stream2` and `stream` (made with the lowest priority for that)
	int n_bz_streams = 0;
	// ... and to a third stream of their own from the second call on: made by a thread that the first call leaves behind
	// (15-20 ms that no caller waits for)
	std::thread bz_third_maker;
	std::atomic<bool> bz_third_ready{false};
	hipStream_t bz_third = nullptr;
	hipEvent_t ev_bz_third = nullptr;
	// The upload is a JOB run by a thread of the context's own (BzJob below): conga_reads_bgzf* starts one and launches the
	// inflates behind its batches; conga_reads_bgzf_next_fd queues the NEXT sample's behind it, into the other of two device
	// buffers, so that sample k + 1 is on its way up while sample k is walked, computed and written out.
	std::thread bz_up_thread;
	std::mutex bz_up_mu;
	std::condition_variable bz_up_cv;
	std::deque<std::shared_ptr<struct BzJob>> bz_up_queue;
	bool bz_up_quit = false, bz_up_busy = false;
	// named ahead, not yet taken up by a conga_reads_bgzf_fd call, in the order of their calls: at most three (a cohort names two
	// samples ahead, and its planning thread may do so before the call for the sample in front has taken ITS bytes up)
	std::vector<std::shared_ptr<struct BzJob>> bz_named;
	bool bz_in_call = false; // a conga_reads_bgzf* call is between queueing its bytes and its return
	double bz_ratio = 0;     // inflated bytes per compressed byte of the largest call so far: sizes the spare output buffer
	std::thread bz_prewarm; // CONGA_FLAG_EXPECT_COHORT: gets the second buffer of compressed bytes and the spare output set while the first sample is on
	bool bz_prewarmed = false;
	std::atomic<bool> sr_layout{false}; // a chromosome has its reference text (conga_reference): split reads will be mapped on the records
	                                    // where the inflate leaves them -- the inflated stream of a sample is in use until its compute is
	                                    // through, so bytes named ahead are only brought up, not inflated ahead (no spare output set)
	std::shared_ptr<struct BzJob> bz_spare_owner; // the named job whose inflates fill the spare output set (until the call that takes it up swaps the sets)
	std::set<uint64_t> bz_spare_waiting;          // tickets of the named jobs that will inflate ahead and have not got the set yet: it goes to the
	                                              // OLDEST of them (bz_up_mu).  Whoever wakes first took it until tests/soak.py --bam (seed 81, case 38):
	                                              // the job named second behind the call got the set, the call in front waited for the job named first
	                                              // to be inflated, that one for the set, the set for the call behind -- a standstill
	uint8_t *bz_up_buf[2] = {nullptr, nullptr};
	size_t bz_up_cap[2] = {0, 0};
	std::shared_ptr<struct BzJob> bz_buf_owner[2]; // a buffer is its job's until the call that took the bytes up is through with them
	uint64_t bz_up_tickets = 0;
	bool bz_slot_used[12] = {};
	const uint8_t *bz_in_now = nullptr; // the compressed bytes the last overlapped upload brought
	hipStream_t bz_ahead[2] = {nullptr, nullptr}; // launch streams of the inflate ahead (lowest priority), made by its thread
	hipEvent_t ev_bz_ahead[2] = {nullptr, nullptr};
	std::shared_ptr<struct BzJob> bz_job_kept;

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
	hipEvent_t ev_done = nullptr;
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

// What depends on the read tuples of the sample and on nothing else: where each chromosome's tuples lie, and the
// geometry of the pass over them.  One small upload from a pinned block, no wait: this is all that stands between
// "another sample's tuples are in HBM" and the kernels when the layout is unchanged (a cohort against one call set).
int prepare_sample(conga_ctx *ctx)
{
	const int n_slots = (int) ctx->slots.size();
	drop_graph(ctx); // sizes and grids below are baked into a captured step
	// the tuples lie in chromosome order (both producers append that way)
	{
		int64_t at = 0;
		for (HostSlot &h : ctx->slots) {
			h.read_off = at;
			at += h.n_reads;
		}
	}
	ctx->tuple_chunks = (uint32_t) ((ctx->n_reads_total + kTupleChunk - 1) / kTupleChunk);
	int blocks = ctx->n_cu * ctx->tuple_blocks_per_cu;
	if (const char *e = getenv("CONGA_TUPLE_BLOCKS_PER_CU")) // tuning knob
		blocks = ctx->n_cu * std::max(1, atoi(e));
	ctx->tuple_chunks_per_block = std::max<uint32_t>(1, (ctx->tuple_chunks + (uint32_t) blocks - 1) / (uint32_t) blocks);
	const uint32_t grid = (ctx->tuple_chunks + ctx->tuple_chunks_per_block - 1) / ctx->tuple_chunks_per_block;
	const size_t n_homes = std::max<uint32_t>(grid, 1);
	const size_t homes_at = ((size_t) std::max(n_slots, 1) * sizeof(Slot) + 255) & ~(size_t) 255;
	// the split-read launch's table: one SplitSlot per chromosome that has a reference AND records of this sample
	const size_t sr_at = (homes_at + n_homes * sizeof(TupleBlockHome) + 255) & ~(size_t) 255;
	const size_t bytes = sr_at + (size_t) std::max(n_slots, 1) * sizeof(SplitSlot);
	if (bytes > ctx->h_head_cap) {
		if (ctx->head_in_flight)
			HIP_TRY(ctx, hipEventSynchronize(ctx->ev_head));
		ctx->head_in_flight = false;
		if (ctx->h_head)
			(void) hipHostFree(ctx->h_head);
		ctx->h_head = nullptr;
		ctx->h_head_cap = 0;
		const size_t cap = bytes + bytes / 2 + 4096;
		HIP_TRY(ctx, hipHostMalloc(&ctx->h_head, cap, hipHostMallocDefault));
		ctx->h_head_cap = cap;
	}
	TRY(ensure(ctx, ctx->d_head, bytes));
	if (ctx->head_in_flight) { // the previous sample's upload still reads the pinned block
		HIP_TRY(ctx, hipEventSynchronize(ctx->ev_head));
		ctx->head_in_flight = false;
	}
	Slot *dslots = static_cast<Slot *>(ctx->h_head);
	for (int s = 0; s < n_slots; s++) {
		const HostSlot &h = ctx->slots[s];
		Slot &d = dslots[s];
		d.L = h.L;
		d.rd_off = h.rd_off;
		d.read_off = h.read_off;
		d.n_reads = h.n_reads;
		d.gc_off = h.gc_off;
		d.n_win = h.n_win;
		d.tile0 = h.tile0;
		d.n_tiles = h.n_tiles;
		d.tidx_off = h.tidx_off;
	}
	// tuple pass: contiguous runs of 1024-tuple chunks per workgroup, and the chromosome each run starts in
	TupleBlockHome *homes = reinterpret_cast<TupleBlockHome *>(static_cast<char *>(ctx->h_head) + homes_at);
	int s = 0;
	for (uint32_t b = 0; b < (uint32_t) n_homes; b++) {
		TupleBlockHome &bh = homes[b];
		memset(&bh, 0, sizeof bh);
		bh.sl.r0 = 1; // empty range
		bh.slot = -1;
		const int64_t base = (int64_t) b * ctx->tuple_chunks_per_block * kTupleChunk;
		while (s + 1 < n_slots && ctx->slots[s + 1].read_off <= base)
			s++;
		const HostSlot &h = ctx->slots[s];
		if (base >= h.read_off && base + kTupleChunk <= h.read_off + h.n_reads) { // first chunk inside one chromosome
			bh.sl.r0 = (uint32_t) h.read_off;
			bh.sl.r1 = (uint32_t) (h.read_off + h.n_reads);
			bh.sl.L = (int32_t) h.L;
			bh.sl.gc_off = (uint32_t) h.gc_off;
			bh.slot = s;
		}
	}
	{
		SplitSlot *ss = reinterpret_cast<SplitSlot *>(static_cast<char *>(ctx->h_head) + sr_at);
		int k = 0;
		uint64_t units = 0;
		for (int c = 0; c < n_slots; c++) {
			const HostSlot &h = ctx->slots[(size_t) c];
			if (h.kidx < 0 || h.n_sr <= 0)
				continue;
			SplitSlot &sl = ss[k++];
			memset(&sl, 0, sizeof sl);
			sl.sr_off = h.sr_off;
			sl.n_sr = h.n_sr;
			sl.refn_off = h.refn_off;
			sl.L = h.L;
			sl.kpos_off = h.kpos_off;
			sl.kidx = h.kidx;
			sl.sat_off = (int32_t) h.sat_off;
			sl.n_sat = (int32_t) h.sat_start.size();
			sl.iv0 = (int32_t) h.iv0;
			sl.n_dels = (int32_t) h.iv_start[0].size();
			sl.n_dups = (int32_t) h.iv_start[1].size();
			sl.slot = c;
			sl.unit0 = (uint32_t) units;
			sl.inplace = h.sr_inplace ? 1 : 0;
			units += (uint64_t) ((h.n_sr + kSplitUnitReads - 1) / kSplitUnitReads);
		}
		ctx->n_sr_slots = k;
		ctx->sr_units = (uint32_t) units; // (fewer than 2^32 reads in a context: far fewer units)
		ctx->any_sr = k > 0;
	}
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_head.p, ctx->h_head, bytes, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipEventRecord(ctx->ev_head, ctx->stream));
	ctx->head_in_flight = true;
	ctx->d_slots.p = ctx->d_head.p;
	ctx->d_block_home.p = static_cast<char *>(ctx->d_head.p) + homes_at;
	ctx->d_sr_slots.p = static_cast<char *>(ctx->d_head.p) + sr_at;
	ctx->sample_dirty = false;
	return CONGA_OK;
}

// Lay the batch out in the concatenated buffers and upload everything that is not a read tuple and does not depend on
// the tuples: with the same chromosomes, intervals and tracks it is done once for any number of samples.
int prepare_layout(conga_ctx *ctx)
{
	const int n_slots = (int) ctx->slots.size();
	drop_graph(ctx); // buffers, sizes and grids below are baked into the captured step

	// ---- geometry
	int64_t rd_off = 0, gc_off = 0, tile0 = 0, iv0 = 0, map_rows = 0;
	ctx->gc_like_distinct = false;
	ctx->any_map = false;
	ctx->any_map_painted = false;
	ctx->any_map_rows = false;
	ctx->support_given = false;
	// split-read inputs: every chromosome with a reference sequence gets its packed reference (8 bases per dword, kRefPadBases
	// of code 0 behind it), its 10-mer index (one position per base) and its satellites, whatever records this sample has
	int64_t refn_words = 0, kpos_total = 0, sat_total = 0;
	int n_ref = 0;
	for (int s = 0; s < n_slots; s++) {
		HostSlot &h = ctx->slots[s];
		h.kidx = -1;
		if (!h.ref.empty()) {
			h.kidx = n_ref++;
			h.refn_off = refn_words;
			h.kpos_off = kpos_total;
			h.sat_off = sat_total;
			refn_words += ((h.L + kRefPadBases + 7) / 8 + 63) & ~(int64_t) 63;
			kpos_total += (h.L + 63) & ~(int64_t) 63;
			sat_total += (int64_t) h.sat_start.size();
		}
	}
	ctx->any_ref = n_ref > 0;
	ctx->refn_words = refn_words;
	ctx->kpos_total = kpos_total;
	ctx->sat_total = sat_total;
	for (int s = 0; s < n_slots; s++) {
		HostSlot &h = ctx->slots[s];
		h.rd_off = rd_off;
		h.gc_off = gc_off;
		h.tile0 = tile0;
		h.tidx_off = tile0 + s;
		h.iv0 = iv0;
		h.map_row_off = map_rows;
		rd_off += (h.L + kDepthMaxTile - 1) & ~(int64_t) (kDepthMaxTile - 1); // whole tiles: 4 KiB-aligned regions
		gc_off += (h.n_win + 15) & ~(int64_t) 15;
		tile0 += h.n_tiles;
		iv0 += (int64_t) (h.iv_start[0].size() + h.iv_start[1].size());
		map_rows += (int64_t) h.map_start.size();
		if (!h.gc_like.empty())
			ctx->gc_like_distinct = true;
		if (h.has_map)
			ctx->any_map = true;
		if (h.has_map && h.iv_start[0].size() + h.iv_start[1].size() > 0) {
			if (track_painted(ctx, h))
				ctx->any_map_painted = true;
			else
				ctx->any_map_rows = true;
		}
		if (!h.iv_support[0].empty() || !h.iv_support[1].empty())
			ctx->support_given = true;
	}
	ctx->total_L = rd_off;
	ctx->total_gc = gc_off;
	ctx->total_tiles = tile0;
	ctx->n_iv = iv0;
	ctx->layout_dense = dense_formulation(ctx);
	ctx->sample_dirty = true; // the Slot table carries layout offsets too
	TRY(prepare_sample(ctx)); // (gc_bases_kernel below reads the Slot table)

	{
		// depth workgroups: contiguous tile ranges that never cross a chromosome, dispatched in genome order
		std::vector<DepthBlock> blocks;
		int64_t tiles_per_block = kDepthTilesPerBlock;
		if (const char *e = getenv("CONGA_DEPTH_TILES_PER_BLOCK")) // tuning knob
			tiles_per_block = std::max(1, atoi(e));
		for (int s = 0; s < n_slots; s++) {
			const HostSlot &h = ctx->slots[s];
			for (int64_t t = 0; t < h.n_tiles; t += tiles_per_block) {
				DepthBlock b;
				b.slot = s;
				b.n_tiles = (int32_t) std::min<int64_t>(tiles_per_block, h.n_tiles - t);
				b.first_tile = h.tile0 + t;
				blocks.push_back(b);
			}
		}
		ctx->n_depth_blocks = (int64_t) blocks.size();
		TRY(upload(ctx, ctx->d_depth_blocks, blocks.data(), blocks.size() * sizeof(DepthBlock)));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	// one arena, one memset per compute: the per-chromosome blocks, then observed[n_iv]
	ctx->small_bytes = (std::max<size_t>(n_slots, 1) * sizeof(Small) + 255) & ~(size_t) 255;
	ctx->arena_bytes = (ctx->small_bytes + std::max<size_t>((size_t) ctx->n_iv, 1) * 4 + 255) & ~(size_t) 255;
	TRY(ensure(ctx, ctx->d_small, 2 * ctx->arena_bytes));
	ctx->small_cur = 0;
	ctx->arena_zeroed[0] = ctx->arena_zeroed[1] = false;
	// d_rd / d_tile_start (6 GB for a human genome) are allocated by the first compute that materialises read_depth
	if ((size_t) n_slots > ctx->h_small_cap) {
		if (ctx->h_small)
			(void) hipHostFree(ctx->h_small);
		ctx->h_small = nullptr;
		ctx->h_small_cap = 0;
		const size_t cap = (size_t) n_slots + 8;
		HIP_TRY(ctx, hipHostMalloc((void **) &ctx->h_small, cap * sizeof(Small), hipHostMallocDefault));
		ctx->h_small_cap = cap;
	}

	// ---- GC bytes (padded to 16 per slot)
	{
		std::vector<uint8_t> gh((size_t) ctx->total_gc, 0), gl;
		if (ctx->gc_like_distinct)
			gl.assign((size_t) ctx->total_gc, 0);
		for (int s = 0; s < n_slots; s++) {
			const HostSlot &h = ctx->slots[s];
			memcpy(gh.data() + h.gc_off, h.gc_hist.data(), (size_t) h.n_win);
			if (ctx->gc_like_distinct)
				memcpy(gl.data() + h.gc_off, h.gc_like.empty() ? h.gc_hist.data() : h.gc_like.data(), (size_t) h.n_win);
		}
		TRY(upload(ctx, ctx->d_gc_hist, gh.data(), gh.size()));
		if (ctx->gc_like_distinct)
			TRY(upload(ctx, ctx->d_gc_like, gl.data(), gl.size()));
		// window_per_gc depends on the annotation only: computed once per layout
		TRY(ensure(ctx, ctx->d_bases, (size_t) n_slots * kGcBins * 8));
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_bases.p, 0, (size_t) n_slots * kGcBins * 8, ctx->stream));
		hipLaunchKernelGGL(gc_bases_kernel, dim3(64, n_slots), dim3(256), 0, ctx->stream, ptr<uint8_t>(ctx->d_gc_hist),
				ptr<Slot>(ctx->d_slots), n_slots, ctx->step, ptr<unsigned long long>(ctx->d_bases));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}

	// ---- mappability rows
	if (ctx->any_map) {
		std::vector<int32_t> ms((size_t) map_rows), me((size_t) map_rows);
		std::vector<float> mv((size_t) map_rows);
		int64_t max_L = 0;
		bool any_unsorted = false;
		for (const HostSlot &h : ctx->slots) {
			if (!h.has_map)
				continue;
			std::copy(h.map_start.begin(), h.map_start.end(), ms.begin() + h.map_row_off);
			std::copy(h.map_end.begin(), h.map_end.end(), me.begin() + h.map_row_off);
			std::copy(h.map_val.begin(), h.map_val.end(), mv.begin() + h.map_row_off);
			if (!h.map_sorted) {
				any_unsorted = true;
				max_L = std::max(max_L, h.L);
			}
		}
		TRY(upload(ctx, ctx->d_map_start, ms.data(), ms.size() * 4));
		TRY(upload(ctx, ctx->d_map_end, me.data(), me.size() * 4));
		TRY(upload(ctx, ctx->d_map_val, mv.data(), mv.size() * 4));
		if (ctx->any_map_painted) // 11.5 GB for a human genome: only when some track really is painted
			TRY(ensure(ctx, ctx->d_map, std::max<size_t>((size_t) ctx->total_L, 8) * 4));
		if (any_unsorted)
			TRY(ensure(ctx, ctx->d_winner, (size_t) max_L * 4));
		// per-tile first-row index of every sorted track (rows do not change between computes)
		int64_t rt = 0;
		for (HostSlot &h : ctx->slots) {
			h.row_tile_off = rt;
			if (h.has_map && h.map_sorted)
				rt += (h.L + kPaintTile - 1) / kPaintTile + 2;
		}
		TRY(ensure(ctx, ctx->d_row_tile, std::max<size_t>((size_t) rt, 1) * 4));
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_row_tile.p, 0xFF, std::max<size_t>((size_t) rt, 1) * 4, ctx->stream));
		for (const HostSlot &h : ctx->slots) {
			if (!h.has_map || !h.map_sorted || h.map_start.empty())
				continue;
			const int64_t mrows = (int64_t) h.map_start.size();
			const int grid = (int) std::min<int64_t>((mrows + 255) / 256, (int64_t) ctx->n_cu * 8);
			hipLaunchKernelGGL(row_tile_index_kernel, dim3(grid), dim3(256), 0, ctx->stream,
					ptr<int32_t>(ctx->d_map_start) + h.map_row_off, mrows, kPaintTile, (h.L + kPaintTile - 1) / kPaintTile,
					ptr<uint32_t>(ctx->d_row_tile) + h.row_tile_off);
		}
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}

	// ---- split-read inputs: satellite intervals, packed references and the 10-mer indexes.  An index depends on the
	// chromosome's sequence only: it is built here, once per layout (and again only for new text), and stays resident
	// (4 bytes per base) -- a compute only maps reads against it.
	if (ctx->any_ref) {
		TRY(ensure(ctx, ctx->d_sat_start, std::max<size_t>((size_t) ctx->sat_total, 1) * 4));
		TRY(ensure(ctx, ctx->d_sat_end, std::max<size_t>((size_t) ctx->sat_total, 1) * 4));
		std::vector<uint64_t> sig;
		int64_t max_L = 0;
		for (int s = 0; s < n_slots; s++) {
			const HostSlot &h = ctx->slots[s];
			if (h.kidx < 0)
				continue;
			max_L = std::max(max_L, h.L);
			sig.push_back((uint64_t) s);
			sig.push_back((uint64_t) h.ref.size());
			sig.push_back(h.ref_version);
			if (!h.sat_start.empty()) {
				HIP_TRY(ctx, hipMemcpyAsync(ptr<int32_t>(ctx->d_sat_start) + h.sat_off, h.sat_start.data(), h.sat_start.size() * 4,
						hipMemcpyHostToDevice, ctx->stream));
				HIP_TRY(ctx, hipMemcpyAsync(ptr<int32_t>(ctx->d_sat_end) + h.sat_off, h.sat_end.data(), h.sat_end.size() * 4,
						hipMemcpyHostToDevice, ctx->stream));
			}
		}
		if (sig != ctx->index_sig) {
			const auto t_index = std::chrono::steady_clock::now();
			const size_t n_idx = sig.size() / 3;
			TRY(ensure(ctx, ctx->d_refn, (size_t) ctx->refn_words * 4 + 256));
			TRY(ensure(ctx, ctx->d_kmer_pos, (size_t) ctx->kpos_total * 4 + 256));
			TRY(ensure(ctx, ctx->d_kmer_offset, n_idx * ((size_t) kKmerBuckets + 2) * 4));
			// scratch of the build, sized for the longest chromosome: its text, a sort key per position, the keys in sorted
			// order, and what the sort asks for
			DevBuf text;
			// the sort's scratch: two (key, value) buffers to go back and forth between (the third pass writes the keys into the
			// first one's keys and the positions where they stay), the (digit, tile) counts, the digits' totals and bases
			const size_t max_n = ((size_t) max_L + 63) & ~(size_t) 63;
			const size_t max_tiles = (max_n + kRadixTile - 1) / kRadixTile;
			const size_t tmp_bytes = 3 * max_n * 4 + (size_t) kRadixBins * max_tiles * 4 + 2 * kRadixBins * 4;
			int rc = ensure(ctx, text, (size_t) max_L + 64);
			if (rc == CONGA_OK)
				rc = ensure(ctx, ctx->d_kmer_keys, (size_t) max_L * 4 + 256);
			if (rc == CONGA_OK)
				rc = ensure(ctx, ctx->d_kmer_sorted, (size_t) max_L * 4 + 256);
			if (rc == CONGA_OK)
				rc = ensure(ctx, ctx->d_kmer_tmp, tmp_bytes + 256);
			for (int s = 0; s < n_slots && rc == CONGA_OK; s++) {
				const HostSlot &h = ctx->slots[s];
				if (h.kidx < 0)
					continue;
				hipStream_t st = ctx->stream;
				uint32_t *refn = ptr<uint32_t>(ctx->d_refn) + h.refn_off;
				const int64_t n_words = (h.L + kRefPadBases + 7) / 8;
				hipError_t e = hipMemcpyAsync(text.p, h.ref.data(), (size_t) h.L, hipMemcpyHostToDevice, st);
				if (e == hipSuccess) {
					const int gp = (int) std::min<int64_t>((n_words + 255) / 256, (int64_t) ctx->n_cu * 16);
					hipLaunchKernelGGL(ref_pack_kernel, dim3(gp), dim3(256), 0, st, ptr<uint8_t>(text), h.L, refn, n_words);
					const int gk = (int) std::min<int64_t>(((h.L + 7) / 8 + 255) / 256, (int64_t) ctx->n_cu * 16);
					hipLaunchKernelGGL(kmer_key_kernel, dim3(gk), dim3(256), 0, st, refn, h.L, ptr<uint32_t>(ctx->d_kmer_keys));
					{
						// three stable passes of 7 bits over the 21-bit keys (kmer_sort.hip.h)
						const uint32_t n = (uint32_t) h.L, n_tiles = (uint32_t) ((h.L + kRadixTile - 1) / kRadixTile);
						uint32_t *k0 = ptr<uint32_t>(ctx->d_kmer_keys), *kA = ptr<uint32_t>(ctx->d_kmer_sorted);
						int32_t *vA = ptr<int32_t>(ctx->d_kmer_tmp);
						uint32_t *kB = reinterpret_cast<uint32_t *>(vA + max_n);
						int32_t *vB = reinterpret_cast<int32_t *>(kB + max_n);
						uint32_t *counts = reinterpret_cast<uint32_t *>(vB + max_n), *totals = counts + (size_t) kRadixBins * max_tiles, *base = totals + kRadixBins;
						const unsigned g = (n_tiles + kRadixWaves - 1) / kRadixWaves;
						auto pass = [&](const uint32_t *ki, const int32_t *vi, int shift, uint32_t *ko, int32_t *vo) {
							hipLaunchKernelGGL(radix_hist_kernel, dim3(g), dim3(64 * kRadixWaves), 0, st, ki, n, shift, n_tiles, counts);
							hipLaunchKernelGGL(radix_scan_kernel, dim3(kRadixBins), dim3(1024), 0, st, counts, n_tiles, totals);
							hipLaunchKernelGGL(radix_base_kernel, dim3(1), dim3(kRadixBins), 0, st, totals, base);
							hipLaunchKernelGGL(radix_scatter_kernel, dim3(g), dim3(64 * kRadixWaves), 0, st, ki, vi, n, shift, n_tiles, counts, base, ko, vo);
						};
						pass(k0, nullptr, 0, kA, vA);
						pass(kA, vA, kRadixBits, kB, vB);
						pass(kB, vB, 2 * kRadixBits, kA, ptr<int32_t>(ctx->d_kmer_pos) + h.kpos_off);
					}
					const int gb = (int) std::min<int64_t>((h.L + 256) / 256, (int64_t) ctx->n_cu * 16);
					hipLaunchKernelGGL(kmer_bounds_kernel, dim3(gb), dim3(256), 0, st, ptr<uint32_t>(ctx->d_kmer_sorted), h.L,
							ptr<uint32_t>(ctx->d_kmer_offset) + (size_t) h.kidx * ((size_t) kKmerBuckets + 2));
					e = hipGetLastError();
				}
				if (e != hipSuccess && rc == CONGA_OK)
					rc = fail(ctx, CONGA_ERR_HIP, std::string("10-mer index: ") + hipGetErrorString(e));
			}
			(void) hipStreamSynchronize(ctx->stream);
			free_buf(text);
			// (the sort's scratch is a few bytes per base of the longest chromosome: given back, the index is built once)
			free_buf(ctx->d_kmer_keys);
			free_buf(ctx->d_kmer_sorted);
			free_buf(ctx->d_kmer_tmp);
			TRY(rc);
			ctx->index_sig = sig;
			if (getenv("CONGA_TIMING"))
				fprintf(stderr, "[timing] 10-mer indexes of %zu chromosomes (%.0f Mb) built in %.1f ms (once per reference)\n", n_idx,
						ctx->kpos_total / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_index).count());
		}
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (the satellite vectors are read by the uploads)
	}

	// ---- intervals: slot order, dels then dups inside a slot
	const size_t n = (size_t) ctx->n_iv;
	ctx->n_items = 0;
	ctx->n_chain_x = 0;
	ctx->n_chain_a = 0;
	ctx->n_chain_b = 0;
	if (n > 0) {
		std::vector<int32_t> start(n), end(n), iv_slot(n), order(n), item_first(n + 1), support;
		std::vector<uint8_t> type(n), iv_has_map(n);
		size_t k = 0;
		for (int s = 0; s < n_slots; s++) {
			const HostSlot &h = ctx->slots[s];
			for (int t = 0; t < 2; t++)
				for (size_t i = 0; i < h.iv_start[t].size(); i++, k++) {
					start[k] = h.iv_start[t][i];
					end[k] = h.iv_end[t][i];
					type[k] = t == 0 ? CONGA_DELETION : CONGA_DUPLICATION;
					iv_slot[k] = s;
					iv_has_map[k] = !h.has_map ? 0 : track_painted(ctx, h) ? 2 : 1;
				}
		}
		std::vector<int32_t> n_windows(n);
		for (size_t i = 0; i < n; i++)
			n_windows[i] = (end[i] <= start[i]) ? 0
					: (int32_t) (((int64_t) end[i] - 1) / ctx->step - (int64_t) start[i] / ctx->step + 1);
		// longest chains first: the lanes / groups of a wave in interval_chain_kernel then retire together, and the
		// four classes of that kernel are contiguous ranges of order[]
		std::iota(order.begin(), order.end(), 0);
		std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return n_windows[x] > n_windows[y]; });
		int32_t long_min = kChainLongWindows, serial_max = kChainSerialWindows;
		if (const char *e = getenv("CONGA_CHAIN_LONG_WINDOWS")) // tuning knobs
			long_min = std::max(1, atoi(e));
		if (const char *e = getenv("CONGA_CHAIN_SERIAL_WINDOWS"))
			serial_max = std::max(0, atoi(e));
		int32_t block_min = kChainBlockWindows;
		if (const char *e = getenv("CONGA_CHAIN_BLOCK_WINDOWS"))
			block_min = std::max(1, atoi(e));
		size_t nx = 0;
		while (nx < n && n_windows[order[nx]] > std::max(block_min, long_min))
			nx++;
		size_t na = nx;
		while (na < n && n_windows[order[na]] > long_min)
			na++;
		size_t nb = na;
		while (nb < n && n_windows[order[nb]] > serial_max)
			nb++;
		ctx->n_chain_x = (int64_t) nx;
		ctx->n_chain_a = (int64_t) (na - nx);
		ctx->n_chain_b = (int64_t) (nb - na);
		ctx->order_pos.assign(n, 0);
		for (size_t k2 = 0; k2 < n; k2++)
			ctx->order_pos[(size_t) order[k2]] = (int32_t) k2;

		// reduce work items: [start, min(end, L)) cut into kItemLen pieces
		std::vector<int64_t> item_off;
		std::vector<int32_t> item_len, item_iv, item_lo;
		std::vector<int32_t> item_slot;
		std::vector<uint32_t> item_row0, item_row1, item_rt_off;
		std::vector<uint8_t> item_has_map;
		item_off.reserve(n + n / 2);
		item_len.reserve(n + n / 2);
		item_iv.reserve(n + n / 2);
		item_has_map.reserve(n + n / 2);
		for (size_t i = 0; i < n; i++) {
			item_first[i] = (int32_t) item_off.size();
			const HostSlot &h = ctx->slots[iv_slot[i]];
			const int64_t s = start[i], e = std::min<int64_t>(end[i], h.L);
			for (int64_t a = s; a < e; a += kItemLen) {
				item_off.push_back(h.rd_off + a);
				item_len.push_back((int32_t) std::min<int64_t>(kItemLen, e - a));
				item_iv.push_back((int32_t) i);
				item_has_map.push_back(iv_has_map[i]);
				item_lo.push_back((int32_t) a);
				item_slot.push_back(iv_slot[i]);
				item_row0.push_back((uint32_t) h.map_row_off);
				item_row1.push_back((uint32_t) (h.map_row_off + (int64_t) h.map_start.size()));
				item_rt_off.push_back((uint32_t) h.row_tile_off);
			}
		}
		item_first[n] = (int32_t) item_off.size();
		ctx->n_items = (int64_t) item_off.size();

		TRY(upload(ctx, ctx->d_iv_start, start.data(), n * 4));
		TRY(upload(ctx, ctx->d_iv_end, end.data(), n * 4));
		TRY(upload(ctx, ctx->d_iv_type, type.data(), n));
		TRY(upload(ctx, ctx->d_iv_slot, iv_slot.data(), n * 4));
		TRY(upload(ctx, ctx->d_iv_has_map, iv_has_map.data(), n));
		TRY(upload(ctx, ctx->d_order, order.data(), n * 4));
		TRY(upload(ctx, ctx->d_item_first, item_first.data(), (n + 1) * 4));
		TRY(upload(ctx, ctx->d_item_off, item_off.data(), item_off.size() * 8));
		TRY(upload(ctx, ctx->d_item_len, item_len.data(), item_len.size() * 4));
		TRY(upload(ctx, ctx->d_item_iv, item_iv.data(), item_iv.size() * 4));
		TRY(upload(ctx, ctx->d_item_has_map, item_has_map.data(), item_has_map.size()));
		TRY(upload(ctx, ctx->d_item_lo, item_lo.data(), item_lo.size() * 4));
		TRY(upload(ctx, ctx->d_item_slot, item_slot.data(), item_slot.size() * 4));
		TRY(upload(ctx, ctx->d_item_row0, item_row0.data(), item_row0.size() * 4));
		TRY(upload(ctx, ctx->d_item_row1, item_row1.data(), item_row1.size() * 4));
		TRY(upload(ctx, ctx->d_item_rt_off, item_rt_off.data(), item_rt_off.size() * 4));
		TRY(ensure(ctx, ctx->d_expected, n * 4));
		TRY(ensure(ctx, ctx->d_map_part, std::max<size_t>(item_off.size(), 1) * 8));
		TRY(ensure(ctx, ctx->d_results, n * sizeof(conga_result)));
		if (ctx->support_given) {
			support.assign(n, 0);
			for (int s = 0; s < n_slots; s++) {
				const HostSlot &h = ctx->slots[s];
				size_t base = (size_t) h.iv0;
				for (int t = 0; t < 2; t++) {
					for (size_t i = 0; i < h.iv_support[t].size() && i < h.iv_start[t].size(); i++)
						support[base + i] = h.iv_support[t][i];
					base += h.iv_start[t].size();
				}
			}
			TRY(upload(ctx, ctx->d_support_base, support.data(), n * 4));
		}
		if (ctx->support_given || ctx->any_ref)
			TRY(ensure(ctx, ctx->d_support, n * 4));
		if (n > ctx->h_results_cap) {
			if (ctx->h_results)
				(void) hipHostFree(ctx->h_results);
			ctx->h_results = nullptr;
			ctx->h_results_cap = 0;
			const size_t cap = n + n / 2 + 64;
			HIP_TRY(ctx, hipHostMalloc((void **) &ctx->h_results, cap * sizeof(conga_result), hipHostMallocDefault));
			ctx->h_results_cap = cap;
		}
		// the uploads above read from vectors that die at return
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	}
	ctx->layout_dirty = false;
	return CONGA_OK;
}

struct KernelTimer {
	conga_ctx *ctx;
	int k;
	bool on;
	KernelTimer(conga_ctx *c, int kernel) : ctx(c), k(kernel), on((c->opts.flags & CONGA_FLAG_PROFILE) != 0)
	{
		if (on) {
			(void) hipEventRecord(ctx->ev_k0[k], ctx->stream);
			ctx->ev_used[k] = true;
		}
	}
	~KernelTimer()
	{
		if (on)
			(void) hipEventRecord(ctx->ev_k1[k], ctx->stream);
	}
};

void reset_slots(conga_ctx *ctx)
{
	ctx->slots.clear();
	ctx->sr_layout.store(false);
	ctx->cur = -1;
	ctx->n_reads_total = 0;
	ctx->wrap_risk = false;
	ctx->depth_resident = false;
	ctx->n_sr_total = 0;
	ctx->sr_bytes_total = 0;
	ctx->sr_staged = false;
	ctx->bz_keep_bytes = 0;
	ctx->expand_pending = false;
	ctx->staging_cur = -1;
	ctx->read_target = -1;
	ctx->layout_dirty = true;
	ctx->sample_dirty = true;
	ctx->computed = false;
}

HostSlot *current(conga_ctx *ctx)
{
	if (ctx->cur < 0 || ctx->cur >= (int) ctx->slots.size())
		return nullptr;
	return &ctx->slots[ctx->cur];
}

// read_depth[] is a `short` (common.h:91): the 32768th read starting at one base wraps it.  The tuple-space
// formulation counts reads and cannot reproduce that, so the commit path looks for runs of equal positions
// (reads are position-sorted, so the reads of one base are consecutive) and flags the batch for the dense kernels
// when one may reach kWrapRun.  Conservative and cheap: inside a batch it probes every 1024th tuple against the
// one kWrapRun - 1 behind it -- any run of 32768 contains such a pair -- and it carries the run that ends a batch
// into the next one.  All tuples count here, whatever their MAPQ.
constexpr int64_t kWrapProbeStride = 1024;
constexpr int64_t kWrapRun = 32768 - kWrapProbeStride + 1; // 31745

void note_equal_runs(conga_ctx *ctx, HostSlot &h, const int32_t *pos, size_t n)
{
	if (ctx->wrap_risk || n == 0)
		return;
	const int64_t N = (int64_t) n;
	int64_t lead = 0;
	if (h.n_reads > 0) {
		while (lead < N && pos[lead] == h.tail_val)
			lead++;
		if (h.tail_len + lead >= kWrapRun)
			ctx->wrap_risk = true;
	}
	for (int64_t i = 0; i + (kWrapRun - 1) < N; i += kWrapProbeStride)
		if (pos[i] == pos[i + (kWrapRun - 1)])
			ctx->wrap_risk = true;
	if (lead == N)
		h.tail_len = std::min<int64_t>(h.tail_len + N, kWrapRun);
	else {
		int64_t len = 1;
		while (len < N && len < kWrapRun && pos[N - 1 - len] == pos[N - 1])
			len++;
		h.tail_val = pos[N - 1];
		h.tail_len = len;
	}
}

// The dense formulation's front end on the sorted tuples: K0 tile index, then K1 + K2 (read_depth[] in d_rd, the
// GC sums and the read counters into `small`).  Also used to materialise read_depth after a tuple-space compute.
int launch_dense_depth(conga_ctx *ctx, Small *small, bool timed)
{
	hipStream_t st = ctx->stream;
	const int n_slots = (int) ctx->slots.size();
	const Slot *dslots = ptr<Slot>(ctx->d_slots);
	TRY(ensure(ctx, ctx->d_rd, std::max<size_t>((size_t) ctx->total_L, 8) * 2));
	TRY(ensure(ctx, ctx->d_tile_start, ((size_t) ctx->total_tiles + 2) * 4));
	HIP_TRY(ctx, hipMemsetAsync(ctx->d_tile_start.p, 0xFF, ((size_t) ctx->total_tiles + 2) * 4, st));
	{
		std::unique_ptr<KernelTimer> t(timed ? new KernelTimer(ctx, CONGA_K_INGEST) : nullptr);
		if (ctx->n_reads_total > 0) {
			const int grid = (int) std::min<int64_t>((ctx->n_reads_total + 255) / 256, (int64_t) ctx->n_cu * 8);
			hipLaunchKernelGGL(ingest_kernel, dim3(grid), dim3(256), 0, st, ptr<int32_t>(ctx->d_pos),
					ctx->n_reads_total, dslots, n_slots, ctx->tile_len, ptr<uint32_t>(ctx->d_tile_start), small);
		}
	}
	{
		std::unique_ptr<KernelTimer> t(timed ? new KernelTimer(ctx, CONGA_K_DEPTH) : nullptr);
		DepthArgs a;
		a.pos = ptr<int32_t>(ctx->d_pos);
		a.mapq = ptr<uint8_t>(ctx->d_mapq);
		a.tile_first = ptr<uint32_t>(ctx->d_tile_start);
		a.n_total = (uint32_t) ctx->n_reads_total;
		a.rd = ptr<int16_t>(ctx->d_rd);
		a.gc_hist = ptr<uint8_t>(ctx->d_gc_hist);
		a.slots = dslots;
		a.blocks = ptr<DepthBlock>(ctx->d_depth_blocks);
		a.small = small;
		a.step = ctx->step;
		a.step_magic = (uint32_t) (0x100000000ull / (uint64_t) ctx->step) + 1u;
		a.tile_len = ctx->tile_len;
		a.mq_threshold = ctx->opts.mq_threshold;
		a.total_tiles = ctx->total_tiles;
		const int grid = (int) ctx->n_depth_blocks;
		hipLaunchKernelGGL(depth_tile_kernel, dim3(grid), dim3(kDepthBlock), 0, st, a);
	}
	return CONGA_OK;
}

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
bool lane_kernel_asked()
{
	const char *which = getenv("CONGA_BGZF_KERNEL");
	return which && strcmp(which, "lane") == 0;
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
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (`x2n` is on the stack)
	return CONGA_OK;
}

int launch_inflate(conga_ctx *ctx, size_t n_blocks, uint32_t lanes, hipStream_t st = nullptr, size_t b0 = 0, const uint8_t *in = nullptr)
{
	if (!st)
		st = ctx->stream;
	if (!in)
		in = ptr<uint8_t>(ctx->d_bz_in);
	if (lane_kernel_asked()) {
		TRY(ensure(ctx, ctx->d_bz_scratch, (size_t) lanes * sizeof(InflateScratch)));
		hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(lanes / 64), dim3(64), 0, st, (uint32_t) n_blocks, in,
				ptr<conga_bgzf_block>(ctx->d_bz_blocks) + b0, ptr<uint64_t>(ctx->d_bz_off) + b0, ptr<uint8_t>(ctx->d_bz_out),
				ptr<InflateScratch>(ctx->d_bz_scratch), ptr<uint32_t>(ctx->d_bz_crc), ptr<uint8_t>(ctx->d_bz_status) + b0);
		return CONGA_OK;
	}
	TRY(ensure_x2n(ctx));
	// one resident round of workgroups (8 per CU), blocks round robin over their waves
	const size_t groups = std::min<size_t>((n_blocks + iw::kWavesPerGroup - 1) / iw::kWavesPerGroup, (size_t) ctx->n_cu * 8);
	// CONGA_BGZF_KERNEL=wave1: round 2's symbol loop (every trip decodes its sixty-four candidates completely), for comparison
	const char *which = getenv("CONGA_BGZF_KERNEL");
	const bool one_phase = which && strcmp(which, "wave1") == 0;
	if (one_phase)
		hipLaunchKernelGGL(iw::bgzf_inflate_wave_kernel<false>, dim3((unsigned) groups), dim3(64 * iw::kWavesPerGroup), 0, st, (uint32_t) n_blocks,
				in, ptr<conga_bgzf_block>(ctx->d_bz_blocks) + b0, ptr<uint64_t>(ctx->d_bz_off) + b0,
				ptr<uint8_t>(ctx->d_bz_out), ptr<uint32_t>(ctx->d_bz_crc), ptr<uint32_t>(ctx->d_bz_x2n), ptr<uint8_t>(ctx->d_bz_status) + b0);
	else
		hipLaunchKernelGGL(iw::bgzf_inflate_wave_kernel<true>, dim3((unsigned) groups), dim3(64 * iw::kWavesPerGroup), 0, st, (uint32_t) n_blocks,
				in, ptr<conga_bgzf_block>(ctx->d_bz_blocks) + b0, ptr<uint64_t>(ctx->d_bz_off) + b0,
				ptr<uint8_t>(ctx->d_bz_out), ptr<uint32_t>(ctx->d_bz_crc), ptr<uint32_t>(ctx->d_bz_x2n), ptr<uint8_t>(ctx->d_bz_status) + b0);
	return CONGA_OK;
}

// The file's bytes to HBM and the inflate of their blocks, overlapped.  A pageable hipMemcpy of gigabytes runs at the rate
// of ONE staging thread inside the runtime (~18 GB/s measured); here host threads copy 16 MB pieces of the caller's bytes
// (the page cache behind an mmap) into a ring of pinned buffers, each piece goes up at the link's rate as soon as it is
// full, and every 128 MB of pieces the inflate of the blocks they complete is launched on one of three streams, so that
// copying in, copying up and inflating all run at once.  Ends with ctx->stream waiting for every launch.
// blocks[] must be in file order (data_off ascending); the caller falls back to the plain form otherwise.
// a slot of the ring (CONGA_BGZF_SLOT_MB: measurement switch; pieces are at most a slot)
size_t bz_slot_bytes()
{
	static const size_t n = getenv("CONGA_BGZF_SLOT_MB") ? (size_t) std::max(1, std::min(atoi(getenv("CONGA_BGZF_SLOT_MB")), 64)) << 20 : (size_t) 8 << 20;
	return n;
}
#define kBzPiece bz_slot_bytes()
constexpr int kBzMaxSlots = 12, kBzMaxStreams = 3, kBzPiecesPerLaunch = 16;
// how many of them are used (CONGA_BGZF_SLOTS / CONGA_BGZF_STREAMS: measurement switches)
int bz_slots()
{
	static const int n = getenv("CONGA_BGZF_SLOTS") ? std::max(2, std::min(atoi(getenv("CONGA_BGZF_SLOTS")), kBzMaxSlots)) : kBzMaxSlots;
	return n;
}
int bz_streams_wanted()
{
	static const int n = getenv("CONGA_BGZF_STREAMS") ? std::max(1, std::min(atoi(getenv("CONGA_BGZF_STREAMS")), kBzMaxStreams)) : kBzMaxStreams;
	return n;
}

// the pinned ring, its events and the streams of the overlapped upload (96 MB of pinned memory take ~50 ms to get: with
// CONGA_FLAG_EXPECT_BGZF conga_create() does this, and a caller that creates its context on a thread of its own -- the
// conga executable does, while it reads the BAM's block table -- never waits for it)
void make_bz_ring(conga_ctx *ctx)
{
	int prio_low = 0, prio_high = 0; // (numerically lower = more urgent)
	(void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
	const bool prio = getenv("CONGA_BGZF_NO_PRIORITY") == nullptr;
	bool ok = hipSetDevice(ctx->device) == hipSuccess
			&& hipHostMalloc((void **) &ctx->h_bz_ring, kBzPiece * (size_t) bz_slots(), hipHostMallocDefault) == hipSuccess;
	if (ok && ctx->bz_copy) // (the ring was given back, conga_release_staging: streams and events are still there)
		return;
	ok = ok && hipStreamCreateWithPriority(&ctx->bz_copy, hipStreamNonBlocking, prio ? prio_high : 0) == hipSuccess;
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

} // namespace

// One sample's compressed bytes on their way to HBM (the context's upload thread runs it): host threads copy pieces of the
// file into the ring of pinned buffers, each piece goes up on the copy stream as soon as it is full, and behind every batch of
// pieces an event is recorded that the inflate launches of that batch wait for.
struct BzJob {
	ByteSource src;
	size_t n_bytes = 0, piece = 0, n_pieces = 0, pieces_per_batch = 0, n_batches = 0;
	int buf = -1;        // which of the context's two device buffers (taken when the job starts)
	std::atomic<bool> released{false}; // nothing reads the buffer any more
	uint64_t ticket = 0; // conga_reads_bgzf_next_fd's
	std::mutex mu;
	std::condition_variable cv;
	std::vector<uint8_t> filled;
	size_t issued = 0;        // pieces whose copy up has been enqueued
	size_t batches_ready = 0; // batches whose event has been recorded
	bool failed = false, short_read = false, done = false, started = false;
	bool queued = false; // handed to the upload thread (bytes named ahead wait for the call in front of theirs to queue its own)
	bool adopted = false; // a conga_reads_bgzf_fd call has taken the job up (mu): no inflating ahead begins behind its back
	std::atomic<bool> cancel{false};
	std::vector<hipEvent_t> ev_batch;
	const uint8_t *d_bytes = nullptr; // where the bytes go (set when the job starts)
	std::string error;
	std::chrono::steady_clock::time_point t_queued, t_started;
	double ms_enqueued = 0, ms_copy = 0, ms_wait = 0; // (CONGA_TIMING)
	int n_threads = 0;
	// inflate ahead (conga_reads_bgzf_next_blocks): a thread launches the batches' inflates into the context's spare output set
	std::vector<conga_bgzf_block> blocks;
	std::vector<uint64_t> out_off;
	uint64_t total_out = 0;
	std::thread inflater;
	bool inflate_asked = false, inflate_done = false, inflate_ok = false; // (mu)
	int launches_ahead = 0;
	double ms_inflate_ahead = 0;
	// The block table read off the bytes as they pass through the pinned ring (conga_reads_bgzf_next_fd with the block starts the
	// caller knows from the index): every copying thread walks the chain of headers inside its piece from the first known start
	// on, the upload thread joins the pieces' findings in file order (what straddles two pieces it reads from the file: a
	// trailer, now and then a header) and publishes the table batch by batch -- the inflate-ahead thread needs no more.
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
	uint64_t stop_at = 0;        // 0, or: the table ends with the first block that starts at or behind this offset
	std::vector<PieceTable> piece_tables;
	uint64_t expect = 0;         // where the chain goes on (upload thread)
	bool table_failed = false, table_stopped = false; // (upload thread; published with the counters below)
	size_t cap_blocks = 0;
	uint64_t cap_out = 0;        // 0: no inflate ahead (only the table)
	size_t table_n = 0;          // blocks published (mu)
	bool table_final = false, table_ok = false; // (mu)
	~BzJob()
	{
		if (inflater.joinable())
			inflater.join();
		for (hipEvent_t e : ev_batch)
			if (e)
				(void) hipEventDestroy(e);
	}
};

namespace {

void bz_inflate_ahead(conga_ctx *ctx, const std::shared_ptr<BzJob> &self);
void bz_prewarm_join(conga_ctx *ctx);

// BGZF header of the usual form at h[0..18) -> BSIZE + 1 (the block's length), or 0
inline uint32_t bgzf_block_len(const uint8_t *h)
{
	if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[10] != 6 || h[11] != 0 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0)
		return 0;
	const uint32_t len = (uint32_t) (h[16] | (h[17] << 8)) + 1u;
	return len >= 26u ? len : 0u;
}

// one piece's share of the table: buf = the piece's bytes [begin, begin + len) of the job's stretch
void bz_walk_piece(BzJob &job, size_t c, const uint8_t *buf, uint64_t begin, size_t len)
{
	BzJob::PieceTable &pt = job.piece_tables[c];
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
void bz_join_piece(BzJob &job, size_t c, const uint8_t *buf, uint64_t begin, size_t len)
{
	if (job.table_failed || job.table_stopped)
		return;
	const uint64_t end = begin + len;
	BzJob::PieceTable &pt = job.piece_tables[c];
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
		if (job.stop_at && b.data_off - 18 >= job.stop_at)
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

void bz_run_job(conga_ctx *ctx, const std::shared_ptr<BzJob> &self)
{
	BzJob &job = *self;
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
	if (hipSetDevice(ctx->device) != hipSuccess)
		return give_up("hipSetDevice", false);
	bz_prewarm_join(ctx); // (the buffers it allocates are about to be looked at)
	{ // one of the two device buffers: the one no job owns, or whose job's bytes nobody reads any more
		std::unique_lock<std::mutex> lk(ctx->bz_up_mu);
		auto free_buf = [&]() {
			for (int b = 0; b < 2; b++)
				if (!ctx->bz_buf_owner[b] || ctx->bz_buf_owner[b]->released.load())
					return b;
			return -1;
		};
		ctx->bz_up_cv.wait(lk, [&] { return free_buf() >= 0 || job.cancel.load(); });
		if (job.cancel.load()) {
			lk.unlock();
			return give_up("given up", false);
		}
		job.buf = free_buf();
		ctx->bz_buf_owner[job.buf] = self;
	}
	// the device buffer (grown only: a cohort's samples are of a size) and the batches' events
	if (ctx->bz_up_cap[job.buf] < job.n_bytes + 512) {
		if (ctx->bz_up_buf[job.buf])
			(void) hipFree(ctx->bz_up_buf[job.buf]);
		ctx->bz_up_buf[job.buf] = nullptr;
		ctx->bz_up_cap[job.buf] = 0;
		const size_t want = job.n_bytes + job.n_bytes / 16 + 512;
		if (hipMalloc((void **) &ctx->bz_up_buf[job.buf], want) != hipSuccess) {
			(void) hipGetLastError();
			return give_up("no device memory for the file's bytes", false);
		}
		ctx->bz_up_cap[job.buf] = want;
	}
	uint8_t *const d_dst = ctx->bz_up_buf[job.buf];
	job.ev_batch.assign(job.n_batches, nullptr);
	for (size_t k = 0; k < job.n_batches; k++)
		if (hipEventCreateWithFlags(&job.ev_batch[k], hipEventDisableTiming) != hipSuccess)
			return give_up("hipEventCreate", false);
	// The table is made here, batch by batch: the inflates can follow it into the spare output set (its thread waits until the set
	// is free: the named job in front of this one owns it until the call that takes THAT one up has swapped it in) -- when a
	// call of this context has shown how much such a file inflates to, and the job is a named one.
	bool ahead = false;
	if (job.build_table && job.ticket != 0 && !getenv("CONGA_BGZF_NO_INFLATE_AHEAD") && !ctx->sr_layout.load()
			&& !(getenv("CONGA_BGZF_KERNEL") && strcmp(getenv("CONGA_BGZF_KERNEL"), "wave") != 0)) {
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		if (ctx->bz_ratio > 0 && ctx->d_bz_x2n.p && ctx->d_bz_crc.p) {
			job.cap_out = (uint64_t) ((double) job.n_bytes * ctx->bz_ratio * 1.25) + ((uint64_t) 64 << 20);
			ahead = true;
		}
	}
	if (ahead) { // (in the order the jobs begin, which is the order they were named in: the spare set goes to the oldest ticket)
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		ctx->bz_spare_waiting.insert(job.ticket);
	}
	bool inflating = false;
	{
		std::lock_guard<std::mutex> g(job.mu);
		job.d_bytes = d_dst;
		job.started = true;
		if (ahead && !job.inflate_asked && !job.adopted && !job.cancel.load()) {
			job.inflate_asked = inflating = true;
			job.inflater = std::thread(bz_inflate_ahead, ctx, self);
		}
	}
	if (ahead && !inflating) { // (taken up by its call, or given up, in the meantime: it will not ask for the set)
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		ctx->bz_spare_waiting.erase(job.ticket);
		ctx->bz_up_cv.notify_all();
	}
	job.cv.notify_all();

	const int n_slots = bz_slots();
	const size_t piece = job.piece, n_pieces = job.n_pieces, n_bytes = job.n_bytes;
	std::atomic<size_t> next_piece{0};
	std::atomic<long long> us_copy{0}, us_wait{0};
	const int device = ctx->device;
	auto worker = [&]() {
		(void) hipSetDevice(device);
		for (;;) {
			const size_t c = next_piece.fetch_add(1);
			if (c >= n_pieces || job.cancel.load())
				return;
			const auto tw = std::chrono::steady_clock::now();
			const int slot = (int) (c % (size_t) n_slots);
			if (c >= (size_t) n_slots) { // the slot still holds piece c - n_slots until that one's copy up is done
				std::unique_lock<std::mutex> lk(job.mu);
				job.cv.wait(lk, [&] { return job.failed || job.issued > c - (size_t) n_slots; });
				if (job.failed)
					return;
			}
			// (a slot's first use in this job: the job before may have left its last pieces in the ring)
			if ((c >= (size_t) n_slots || ctx->bz_slot_used[slot]) && hipEventSynchronize(ctx->ev_bz_slot[slot]) != hipSuccess) {
				std::lock_guard<std::mutex> g(job.mu);
				job.failed = true;
				job.cv.notify_all();
				return;
			}
			const size_t at = c * piece, len = std::min(piece, n_bytes - at);
			const auto tc = std::chrono::steady_clock::now();
			const bool got = getenv("CONGA_BGZF_PLAIN_PREAD") ? job.src.fetch(at, ctx->h_bz_ring + (size_t) slot * kBzPiece, len) // (measurement switch)
					: job.src.fetch_into_ring(at, ctx->h_bz_ring + (size_t) slot * kBzPiece, len);
			if (got && job.build_table)
				bz_walk_piece(job, c, ctx->h_bz_ring + (size_t) slot * kBzPiece, at, len);
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
	int n_threads = (int) std::min<size_t>(n_pieces, std::min<unsigned>(std::max(2u, cpus_allowed() / 2), (unsigned) n_slots));
	if (const char *e = getenv("CONGA_BGZF_COPY_THREADS"))
		n_threads = std::max(1, std::min(atoi(e), n_slots));
	job.n_threads = n_threads;
	std::vector<std::thread> threads;
	for (int t = 0; t < n_threads; t++)
		threads.emplace_back(worker);

	std::string why;
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
		if (job.build_table)
			bz_join_piece(job, c, ctx->h_bz_ring + (size_t) slot * kBzPiece, at, len);
		hipError_t e = hipMemcpyAsync(d_dst + at, ctx->h_bz_ring + (size_t) slot * kBzPiece, len, hipMemcpyHostToDevice, ctx->bz_copy);
		if (e == hipSuccess)
			e = hipEventRecord(ctx->ev_bz_slot[slot], ctx->bz_copy);
		ctx->bz_slot_used[slot] = true;
		const bool batch_end = (c + 1) % job.pieces_per_batch == 0 || c + 1 == n_pieces;
		const size_t batch = c / job.pieces_per_batch;
		if (e == hipSuccess && batch_end)
			e = hipEventRecord(job.ev_batch[batch], ctx->bz_copy);
		if (e != hipSuccess) {
			why = std::string("copy up: ") + hipGetErrorString(e);
			break;
		}
		{
			std::lock_guard<std::mutex> g(job.mu);
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

void bz_upload_loop(conga_ctx *ctx)
{
	for (;;) {
		std::shared_ptr<BzJob> job;
		{
			std::unique_lock<std::mutex> lk(ctx->bz_up_mu);
			ctx->bz_up_busy = false;
			ctx->bz_up_cv.notify_all();
			ctx->bz_up_cv.wait(lk, [&] { return ctx->bz_up_quit || !ctx->bz_up_queue.empty(); });
			if (ctx->bz_up_queue.empty())
				return; // (quit, nothing left)
			job = ctx->bz_up_queue.front();
			ctx->bz_up_queue.pop_front();
			ctx->bz_up_busy = true;
		}
		if (job->cancel.load()) {
			std::lock_guard<std::mutex> g(job->mu);
			job->failed = job->done = true;
			job->error = "given up";
			job->cv.notify_all();
			continue;
		}
		bz_run_job(ctx, job);
	}
}

void bz_enqueue(conga_ctx *ctx, const std::shared_ptr<BzJob> &job) // (bz_up_mu held by the caller)
{
	if (job->queued)
		return;
	job->queued = true;
	job->t_queued = std::chrono::steady_clock::now();
	if (!ctx->bz_up_thread.joinable())
		ctx->bz_up_thread = std::thread(bz_upload_loop, ctx);
	ctx->bz_up_queue.push_back(job);
	ctx->bz_up_cv.notify_all();
}

// a job for n_bytes of `src`, queued behind whatever the upload thread is doing if `now` (bz_up_mu held by the caller)
std::shared_ptr<BzJob> bz_queue_job(conga_ctx *ctx, const ByteSource &src, size_t n_bytes, bool now = true)
{
	std::shared_ptr<BzJob> job = std::make_shared<BzJob>();
	job->src = src;
	job->n_bytes = n_bytes;
	size_t piece = kBzPiece; // (tests: small pieces, so that a small file goes through every part of this)
	if (const char *e = getenv("CONGA_BGZF_PIECE_KB"))
		piece = std::min(kBzPiece, std::max<size_t>(4096, (size_t) atol(e) << 10));
	job->piece = piece;
	job->n_pieces = (n_bytes + piece - 1) / piece;
	// a batch = what one inflate launch takes with three launch streams: 128 MB (small test pieces: sixteen of them)
	job->pieces_per_batch = piece < kBzPiece ? (size_t) kBzPiecesPerLaunch : std::max<size_t>(1, ((size_t) 128 << 20) / piece);
	job->n_batches = (job->n_pieces + job->pieces_per_batch - 1) / job->pieces_per_batch;
	job->filled.assign(job->n_pieces, 0);
	job->t_queued = std::chrono::steady_clock::now();
	if (now)
		bz_enqueue(ctx, job);
	return job;
}

// the job's device buffer may be written again
void bz_release(conga_ctx *ctx, const std::shared_ptr<BzJob> &job)
{
	if (!job)
		return;
	std::lock_guard<std::mutex> g(ctx->bz_up_mu);
	job->released.store(true);
	ctx->bz_up_cv.notify_all();
}

// the spare output set is nobody's again (when it was this job's)
void bz_spare_free(conga_ctx *ctx, const std::shared_ptr<BzJob> &job)
{
	std::lock_guard<std::mutex> g(ctx->bz_up_mu);
	if (ctx->bz_spare_owner == job) {
		ctx->bz_spare_owner.reset();
		ctx->bz_up_cv.notify_all();
	}
}

// gives a job up and waits until the upload thread is through with it
void bz_abandon(conga_ctx *ctx, const std::shared_ptr<BzJob> &job)
{
	if (!job)
		return;
	job->cancel.store(true);
	job->cv.notify_all();
	{
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		ctx->bz_up_cv.notify_all(); // (it may be waiting for a buffer)
		if (!job->queued) { // (never handed to the upload thread: nobody else will say it is done)
			std::lock_guard<std::mutex> g2(job->mu);
			job->failed = job->done = true;
		}
	}
	{
		std::unique_lock<std::mutex> lk(job->mu);
		job->cv.wait(lk, [&] { return job->done && (!job->inflate_asked || job->inflate_done); });
	}
	if (job->inflater.joinable())
		job->inflater.join();
	if (job->inflate_asked && ctx->bz_ahead[0])
		for (int k = 0; k < 2; k++)
			(void) hipStreamSynchronize(ctx->bz_ahead[k]); // (what it launched reads the job's bytes)
	bz_spare_free(ctx, job);
	bz_release(ctx, job);
}

// everything the upload thread has been given is through (conga_release_staging, conga_destroy)
void bz_upload_quiesce(conga_ctx *ctx, bool quit)
{
	std::vector<std::shared_ptr<BzJob>> pre;
	{
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		pre.swap(ctx->bz_named);
	}
	for (const std::shared_ptr<BzJob> &j : pre)
		bz_abandon(ctx, j);
	bz_prewarm_join(ctx);
	bz_release(ctx, ctx->bz_job_kept); // (the call that took those bytes up has returned: nothing reads them)
	{
		std::unique_lock<std::mutex> lk(ctx->bz_up_mu);
		ctx->bz_up_cv.wait(lk, [&] { return ctx->bz_up_queue.empty() && !ctx->bz_up_busy; });
		if (quit)
			ctx->bz_up_quit = true;
		ctx->bz_up_cv.notify_all();
	}
	if (quit && ctx->bz_up_thread.joinable())
		ctx->bz_up_thread.join();
}

bool quiet_ensure(DevBuf &b, size_t bytes);

// CONGA_FLAG_EXPECT_COHORT: what the pipeline of a cohort needs besides the first sample's own buffers -- the second device buffer
// for compressed bytes and the spare output set, ~3.6 bytes of HBM per byte of file -- is allocated by a thread of its own
// while the first sample is inflated, indexed and computed: 45 GB take the runtime 1.3 s, which the second and third sample
// would otherwise wait for (profiles/r03e_cohort_depth.log).
void bz_prewarm_start(conga_ctx *ctx, size_t n_bytes)
{
	if (ctx->bz_prewarmed || !(ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) || getenv("CONGA_BGZF_NO_INFLATE_AHEAD"))
		return;
	ctx->bz_prewarmed = true;
	double ratio;
	{
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		ratio = ctx->bz_ratio;
	}
	const int device = ctx->device;
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
		const size_t cap_blocks = n_bytes / 4096 + 65536;
		const uint64_t cap_out = (uint64_t) ((double) n_bytes * ratio * 1.25) + ((uint64_t) 64 << 20);
		(void) (quiet_ensure(ctx->d_bz_blocks2, cap_blocks * sizeof(conga_bgzf_block)) && quiet_ensure(ctx->d_bz_off2, cap_blocks * 8)
				&& quiet_ensure(ctx->d_bz_out2, (size_t) cap_out + 64) && quiet_ensure(ctx->d_bz_status2, cap_blocks));
	});
}

// (before anything else touches what it allocates: a named job's start, the inflate ahead, the context's end)
void bz_prewarm_join(conga_ctx *ctx)
{
	std::thread t;
	{
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
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

void bz_inflate_ahead(conga_ctx *ctx, const std::shared_ptr<BzJob> &self)
{
	BzJob &job = *self;
	bool ok = hipSetDevice(ctx->device) == hipSuccess;
	bz_prewarm_join(ctx);
	{ // the spare output set: one job at a time, in the order the jobs were named
		std::unique_lock<std::mutex> lk(ctx->bz_up_mu);
		ctx->bz_up_cv.wait(lk, [&] {
			return job.cancel.load() || (!ctx->bz_spare_owner && !ctx->bz_spare_waiting.empty() && *ctx->bz_spare_waiting.begin() == job.ticket);
		});
		ctx->bz_spare_waiting.erase(job.ticket);
		if (job.cancel.load())
			ok = false;
		else
			ctx->bz_spare_owner = self;
		ctx->bz_up_cv.notify_all(); // (the next ticket may be waiting for this one to be out of the way)
	}
	const auto t0 = std::chrono::steady_clock::now();
	// room for the table and the stream: the table's own size when the caller brought it, a bound when it grows with the upload
	const size_t room_blocks = std::max(job.cap_blocks, job.blocks.size());
	const uint64_t room_out = std::max<uint64_t>(job.cap_out, job.total_out);
	ok = ok && quiet_ensure(ctx->d_bz_blocks2, room_blocks * sizeof(conga_bgzf_block)) && quiet_ensure(ctx->d_bz_off2, room_blocks * 8)
			&& quiet_ensure(ctx->d_bz_out2, (size_t) room_out + 64) && quiet_ensure(ctx->d_bz_status2, room_blocks);
	if (ok && !ctx->bz_ahead[0]) {
		int lo = 0, hi = 0;
		ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
		for (int k = 0; ok && k < 2; k++)
			ok = hipStreamCreateWithPriority(&ctx->bz_ahead[k], hipStreamNonBlocking, lo) == hipSuccess
					&& hipEventCreateWithFlags(&ctx->ev_bz_ahead[k], hipEventDisableTiming) == hipSuccess;
	}
	size_t b_done = 0;
	int launches = 0;
	for (size_t batch = 0; ok && batch < job.n_batches; batch++) {
		size_t n_avail = 0;
		bool final = false;
		{
			std::unique_lock<std::mutex> lk(job.mu);
			job.cv.wait(lk, [&] { return job.failed || job.cancel.load() || job.batches_ready > batch; });
			if (job.failed || job.cancel.load() || (job.table_final && !job.table_ok)) {
				ok = false;
				break;
			}
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
			ok = job.out_off[b1 - 1] + job.blocks[b1 - 1].inflated_len <= room_out && b1 <= room_blocks;
			hipStream_t ks = ctx->bz_ahead[launches % 2];
			ok = ok && hipMemcpyAsync(ptr<conga_bgzf_block>(ctx->d_bz_blocks2) + b_done, job.blocks.data() + b_done, n * sizeof(conga_bgzf_block),
							hipMemcpyHostToDevice, ks) == hipSuccess
					&& hipMemcpyAsync(ptr<uint64_t>(ctx->d_bz_off2) + b_done, job.out_off.data() + b_done, n * 8, hipMemcpyHostToDevice, ks) == hipSuccess
					&& hipMemsetAsync(ptr<uint8_t>(ctx->d_bz_status2) + b_done, 0xFF, n, ks) == hipSuccess
					&& hipStreamWaitEvent(ks, job.ev_batch[batch], 0) == hipSuccess;
			if (ok) {
				const size_t groups = std::min<size_t>((n + iw::kWavesPerGroup - 1) / iw::kWavesPerGroup, (size_t) ctx->n_cu * 8);
				hipLaunchKernelGGL(iw::bgzf_inflate_wave_kernel<true>, dim3((unsigned) groups), dim3(64 * iw::kWavesPerGroup), 0, ks, (uint32_t) n,
						job.d_bytes, ptr<conga_bgzf_block>(ctx->d_bz_blocks2) + b_done, ptr<uint64_t>(ctx->d_bz_off2) + b_done,
						ptr<uint8_t>(ctx->d_bz_out2), ptr<uint32_t>(ctx->d_bz_crc), ptr<uint32_t>(ctx->d_bz_x2n), ptr<uint8_t>(ctx->d_bz_status2) + b_done);
				ok = hipGetLastError() == hipSuccess;
			}
			launches++;
			b_done = b1;
		}
	}
	for (int k = 0; ok && k < 2; k++)
		ok = hipEventRecord(ctx->ev_bz_ahead[k], ctx->bz_ahead[k]) == hipSuccess;
	if (!ok)
		(void) hipGetLastError();
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
		bool through = true;
		for (int k = 0; k < 2; k++)
			through = hipEventSynchronize(ctx->ev_bz_ahead[k]) == hipSuccess && through;
		bool uploaded;
		{
			std::unique_lock<std::mutex> lk(job.mu);
			job.cv.wait(lk, [&] { return job.done; });
			uploaded = !job.failed;
		}
		if (through && uploaded)
			bz_release(ctx, self);
	}
}

// *inflated: the bytes named ahead came with their block table and are inflated (the launches are enqueued) in what is now the
// context's output set -- the caller goes straight to its walks
int upload_and_inflate_overlapped(conga_ctx *ctx, const ByteSource &src, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		uint64_t base)
{
	const bool timing = getenv("CONGA_TIMING") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	auto ms_since = [](std::chrono::steady_clock::time_point t) {
		return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
	};
	if (!ctx->h_bz_ring && !ctx->bz_ring_failed) {
		bz_upload_quiesce(ctx, false); // (nothing of ours is in the ring: it is not there)
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
	// (the call before has returned behind its walks: nothing reads its compressed bytes any more)
	bz_release(ctx, ctx->bz_job_kept);
	ctx->bz_job_kept.reset();
	// The bytes: already on their way when conga_reads_bgzf_next_fd named exactly these, otherwise a job of this call's own.
	std::shared_ptr<BzJob> job;
	bool ahead = false;
	{
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		ctx->bz_in_call = true; // (until reads_bgzf_from returns)
		for (size_t k = 0; k < ctx->bz_named.size() && !job; k++) {
			BzJob &p = *ctx->bz_named[k];
			if (src.fd >= 0 && p.src.fd == src.fd && p.src.file_off == src.file_off && p.n_bytes == n_bytes) {
				job = ctx->bz_named[k];
				ahead = job->queued;
				bz_enqueue(ctx, job); // (named between two calls: it starts now)
				ctx->bz_named.erase(ctx->bz_named.begin() + (long) k);
			}
		}
	}
	auto enqueue_later = [&]() { // the bytes of the calls after this one: behind this call's (and behind the swap of the output sets below)
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		for (const std::shared_ptr<BzJob> &later : ctx->bz_named)
			bz_enqueue(ctx, later);
	};
	if (job) { // (one that failed before it was asked for is no reason to fail now: start over)
		std::unique_lock<std::mutex> lk(job->mu);
		if (job->failed) {
			lk.unlock();
			bz_abandon(ctx, job);
			job.reset();
			ahead = false;
		}
	}
	if (!job) {
		// Bytes named ahead that are NOT these belong to a later call (a cohort's planning thread may name sample k + 1 before
		// sample k's call gets here): they stay named; when their upload has not begun, this call's goes in front of it.
		// With two or more of them, though, both device buffers may be theirs: the ones behind the first are given up (their
		// calls bring them again) -- a caller that names in the order of its calls, as it must, gets here only at a run's start.
		for (;;) {
			std::shared_ptr<BzJob> last;
			{
				std::lock_guard<std::mutex> g(ctx->bz_up_mu);
				if (ctx->bz_named.size() >= 2) {
					last = ctx->bz_named.back();
					ctx->bz_named.pop_back();
				}
			}
			if (!last)
				break;
			bz_abandon(ctx, last);
		}
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		job = bz_queue_job(ctx, src, n_bytes);
		// (in front of named bytes whose upload has not begun)
		for (size_t at = ctx->bz_up_queue.size() - 1; at > 0 && ctx->bz_up_queue[at - 1]->ticket != 0; at--)
			std::swap(ctx->bz_up_queue[at - 1], ctx->bz_up_queue[at]);
	}
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
		bool asked;
		{
			std::lock_guard<std::mutex> g(job->mu);
			job->adopted = true; // (a job that has not started yet will not start inflating ahead now: this call launches its inflates)
			asked = job->inflate_asked;
		}
		if (asked) {
			{
				std::unique_lock<std::mutex> lk(job->mu);
				job->cv.wait(lk, [&] { return job->inflate_done; });
			}
			if (job->inflater.joinable())
				job->inflater.join();
			if (job->inflate_ok && base == 0 && job->blocks.size() == n_blocks
					&& memcmp(job->blocks.data(), blocks, n_blocks * sizeof(conga_bgzf_block)) == 0) {
				std::swap(ctx->d_bz_out, ctx->d_bz_out2);
				std::swap(ctx->d_bz_blocks, ctx->d_bz_blocks2);
				std::swap(ctx->d_bz_off, ctx->d_bz_off2);
				std::swap(ctx->d_bz_status, ctx->d_bz_status2);
				for (int k = 0; k < 2; k++)
					HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_bz_ahead[k], 0));
				inflated_ahead = true;
			} else if (ctx->bz_ahead[0]) { // (whatever it launched writes the spare set: let it finish before that set is used again)
				for (int k = 0; k < 2; k++)
					(void) hipStreamSynchronize(ctx->bz_ahead[k]);
			}
			bz_spare_free(ctx, job);
		}
	}
	enqueue_later();
	if (inflated_ahead) {
		{
			std::unique_lock<std::mutex> lk(job->mu);
			job->cv.wait(lk, [&] { return job->done; });
		}
		if (timing)
			fprintf(stderr, "\n[timing] overlapped upload: named ahead with its block table %d ms before this call: %zu pieces by %d threads enqueued after "
					"%.1f ms (threads: %.1f ms copying, %.1f ms waiting for a free slot, each), %d inflate launches made ahead (their thread: %.1f ms)\n",
					(int) ms_head_start, job->n_pieces, job->n_threads, job->ms_enqueued, job->ms_copy, job->ms_wait, job->launches_ahead,
					job->ms_inflate_ahead);
		ctx->bz_in_now = job->d_bytes;
		ctx->bz_job_kept = job;
		return CONGA_OK;
	}
	// Launch size: a launch lasts at least one block's 4.4 ms and the launches of a stream follow one another, so with two
	// launch streams (the first call of a context, make_bz_ring) 128 MB per launch -- 3 440 blocks, 42 % of the waves the
	// machine holds -- left it half empty: 97 ms for the stage against 86-93 with 256 MB (32 MB: 248 ms, 64: 143, 384: 96);
	// with three streams 128 MB fill it.  (CONGA_BGZF_LAUNCH_MB: measurement switch, in batches of 128 MB.)
	size_t batches_per_launch = job->piece < kBzPiece ? 1 : (ctx->n_bz_streams >= 3 ? 1 : 2);
	if (const char *e = getenv("CONGA_BGZF_LAUNCH_MB"))
		batches_per_launch = std::max<size_t>(1, (size_t) std::max(1, atoi(e)) / 128);
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
			const hipError_t e = hipStreamWaitEvent(ks, job->ev_batch[last_batch], 0);
			if (e != hipSuccess)
				rc = fail(ctx, CONGA_ERR_HIP, std::string("conga_reads_bgzf: ") + hipGetErrorString(e));
			else if (!getenv("CONGA_BGZF_UPLOAD_ONLY")) // (measurement switch: the copy up alone; the call then fails its checks)
				rc = launch_inflate(ctx, b1 - b_done, 0, ks, b_done, job->d_bytes);
			launches++;
			b_done = b1;
		}
		batch = last_batch + 1;
	}
	if (rc != CONGA_OK)
		bz_abandon(ctx, job);
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
		(void) hipStreamWaitEvent(ctx->stream, job->ev_batch[job->n_batches - 1], 0);
	ctx->bz_in_now = job->d_bytes;
	ctx->bz_job_kept = job; // (its events are waited for by work still in flight)
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

// =============================================================================================
extern "C" {

int conga_abi_version(void)
{
	return CONGA_ABI_VERSION;
}

int conga_release_staging(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	// (the thread the first conga_reads_bgzf* left behind to make a third launch stream: a caller that is done reading must not
	// leave the process while it is inside the runtime)
	if (ctx->bz_third_maker.joinable())
		ctx->bz_third_maker.join();
	bz_upload_quiesce(ctx, false); // (an upload named ahead and never asked for is given up: its pieces go through the ring)
	if (!ctx->h_bz_ring)
		return CONGA_OK;
	if (hipSetDevice(ctx->device) != hipSuccess)
		return CONGA_ERR_HIP;
	if (ctx->bz_copy)
		(void) hipStreamSynchronize(ctx->bz_copy); // (every piece has long gone up: conga_reads_bgzf* returns behind its checks)
	uint8_t *ring = ctx->h_bz_ring;
	ctx->h_bz_ring = nullptr;
	return hipHostFree(ring) == hipSuccess ? CONGA_OK : CONGA_ERR_HIP;
}

int conga_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

const char *conga_strerror(int status)
{
	switch (status) {
	case CONGA_OK: return "ok";
	case CONGA_ERR_INVALID: return "invalid argument or call order";
	case CONGA_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
	case CONGA_ERR_HIP: return "HIP runtime error";
	case CONGA_ERR_NOMEM: return "out of memory";
	case CONGA_ERR_UNSORTED: return "reads are not sorted by position";
	case CONGA_ERR_RANGE: return "coordinate out of range";
	case CONGA_ERR_DATA: return "BGZF / BAM data does not check out (decode on the host)";
	default: return "unknown status";
	}
}

const char *conga_last_error(const conga_ctx *ctx)
{
	return ctx ? ctx->err.c_str() : "";
}

float conga_host_repeat_add_f32(float s, float c, uint32_t k)
{
	return conga_repeat_add_f32(s, c, k);
}

float conga_host_window_add_f32(float s, float c, uint32_t k)
{
	return conga_window_add_f32(s, c, k);
}

conga_ctx *conga_create(int device, const conga_opts *opts, int *status)
{
	int st_dummy;
	if (!status)
		status = &st_dummy;
	int n = 0;
	// CONGA_TIMING: where the creation's time goes (the first call of a process brings the HIP runtime up)
	const bool say = getenv("CONGA_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	double t_part[5] = {0, 0, 0, 0, 0};
	auto lap = [&](int k) {
		const auto now = std::chrono::steady_clock::now();
		t_part[k] += std::chrono::duration<double, std::milli>(now - t_last).count();
		t_last = now;
	};
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
		*status = CONGA_ERR_NO_DEVICE;
		return nullptr;
	}
	lap(0);
	conga_ctx *ctx = new (std::nothrow) conga_ctx();
	if (!ctx) {
		*status = CONGA_ERR_NOMEM;
		return nullptr;
	}
	ctx->device = device;
	ctx->opts.struct_size = sizeof(conga_opts);
	ctx->opts.mq_threshold = -1;
	ctx->opts.gc_step = 100;
	ctx->opts.flags = 0;
	if (opts) {
		if (opts->struct_size < 16) {
			*status = CONGA_ERR_INVALID;
			delete ctx;
			return nullptr;
		}
		ctx->opts.mq_threshold = opts->mq_threshold;
		ctx->opts.gc_step = opts->gc_step > 0 ? opts->gc_step : 100;
		ctx->opts.flags = opts->flags;
		if (opts->struct_size >= 20)
			ctx->opts.min_read_length = opts->min_read_length;
	}
	if (ctx->opts.min_read_length <= 0)
		ctx->opts.min_read_length = 60;
	if (ctx->opts.gc_step > 1024) {
		*status = CONGA_ERR_INVALID;
		delete ctx;
		return nullptr;
	}
	ctx->step = ctx->opts.gc_step;
	// depth tile: 2048 positions (4 KiB of int16, so every tile is one aligned 4 KiB store run) unless the GC step is
	// so small that a tile would touch more than kDepthMaxWin windows
	{
		int32_t tl = kDepthMaxTile;
		if ((int64_t) (kDepthMaxWin - 2) * ctx->step < tl)
			tl = (int32_t) (((int64_t) (kDepthMaxWin - 2) * ctx->step) & ~7);
		if (tl < 8) {
			*status = CONGA_ERR_INVALID;
			delete ctx;
			return nullptr;
		}
		ctx->tile_len = tl;
	}

	auto bail = [&](int st) -> conga_ctx * {
		*status = st;
		conga_destroy(ctx);
		return nullptr;
	};
	if (hipSetDevice(device) != hipSuccess)
		return bail(CONGA_ERR_NO_DEVICE);
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
		ctx->n_cu = prop.multiProcessorCount;
	lap(1);
	{
		// the depth kernel keeps a histogram per workgroup, so its grid is exactly one resident wave of workgroups
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, depth_tile_kernel, kDepthBlock, 0) == hipSuccess && nb > 0)
			ctx->depth_blocks_per_cu = std::min(nb, 8);
		// the tuple pass hands each workgroup a contiguous run of chunks and sizes the grid to one resident wave of
		// workgroups: a workgroup that had to wait for a free slot would double the launch time
		int occ = 8;
		for (const void *k : {(const void *) ingest_tuples_kernel<false>, (const void *) ingest_tuples_kernel<true>,
				(const void *) tuple_pass_kernel<false, false>, (const void *) tuple_pass_kernel<true, false>,
				(const void *) tuple_pass_kernel<false, true>, (const void *) tuple_pass_kernel<true, true>}) {
			int n = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, kTupleBlock, 0) == hipSuccess && n > 0)
				occ = std::min(occ, n);
		}
		ctx->tuple_blocks_per_cu = occ;
		int ns = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&ns, split_map_kernel, 256, 0) == hipSuccess && ns > 0)
			ctx->split_blocks_per_cu = std::min(ns, 8);
	}
	lap(2);
	// (lowest priority: these two streams also take the inflate launches of conga_reads_bgzf*, which must rank below its
	// copy stream -- make_bz_ring; among themselves and against other contexts' streams nothing changes)
	ctx->bz_shared = getenv("CONGA_STREAMS_NORMAL") == nullptr;
	int prio_low = 0, prio_high = 0;
	(void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
	if (!ctx->bz_shared)
		prio_low = 0;
	if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_low) != hipSuccess)
		return bail(CONGA_ERR_HIP);
	if (hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_head, hipEventDisableTiming) != hipSuccess)
		return bail(CONGA_ERR_HIP);
	if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, prio_low) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_reads, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_pair[0], hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_pair[1], hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_fork2, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_counted, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess)
		return bail(CONGA_ERR_HIP);
	for (int k = 0; k < CONGA_K_COUNT; k++)
		if (hipEventCreate(&ctx->ev_k0[k]) != hipSuccess || hipEventCreate(&ctx->ev_k1[k]) != hipSuccess)
			return bail(CONGA_ERR_HIP);
	lap(3);
	if (ctx->opts.flags & CONGA_FLAG_EXPECT_BGZF)
		make_bz_ring(ctx); // (a failure shows when the ring is asked for)
	lap(4);
	if (say)
		fprintf(stderr, "[timing] conga_create: device count %.1f ms, device + properties %.1f ms, occupancy queries (code object load) %.1f ms, "
				"streams + events %.1f ms, pinned ring + its streams %.1f ms\n", t_part[0], t_part[1], t_part[2], t_part[3], t_part[4]);
	*status = CONGA_OK;
	return ctx;
}

void conga_destroy(conga_ctx *ctx)
{
	if (!ctx)
		return;
	(void) hipSetDevice(ctx->device);
	if (ctx->stream)
		(void) hipStreamSynchronize(ctx->stream);
	drop_graph(ctx);
	if (ctx->stream2) {
		(void) hipStreamSynchronize(ctx->stream2);
		(void) hipStreamDestroy(ctx->stream2);
	}
	for (hipEvent_t e : {ctx->ev_reads, ctx->ev_pair[0], ctx->ev_pair[1]})
		if (e)
			(void) hipEventDestroy(e);
	if (ctx->ev_fork)
		(void) hipEventDestroy(ctx->ev_fork);
	if (ctx->ev_fork2)
		(void) hipEventDestroy(ctx->ev_fork2);
	if (ctx->ev_counted)
		(void) hipEventDestroy(ctx->ev_counted);
	if (ctx->ev_join)
		(void) hipEventDestroy(ctx->ev_join);
	// (d_slots and d_block_home are views into d_head)
	DevBuf *bufs[] = {&ctx->d_pos, &ctx->d_mapq, &ctx->d_pos_alt, &ctx->d_mapq_alt, &ctx->d_delta[0], &ctx->d_delta[1], &ctx->d_delta_esc[0],
			&ctx->d_delta_esc[1], &ctx->d_delta_agg, &ctx->d_tile_start, &ctx->d_small_scratch, &ctx->d_item_slot,
			&ctx->d_head, &ctx->d_item_row0, &ctx->d_item_row1, &ctx->d_item_rt_off, &ctx->d_item_lo, &ctx->d_rd, &ctx->d_gc_hist, &ctx->d_gc_like,
			&ctx->d_small, &ctx->d_map, &ctx->d_winner, &ctx->d_map_start, &ctx->d_map_end,
			&ctx->d_map_val, &ctx->d_iv_start, &ctx->d_iv_end, &ctx->d_iv_type, &ctx->d_iv_slot, &ctx->d_iv_has_map,
			&ctx->d_order, &ctx->d_bz_in, &ctx->d_bz_blocks, &ctx->d_bz_off, &ctx->d_bz_out, &ctx->d_bz_status, &ctx->d_bz_out2, &ctx->d_bz_blocks2, &ctx->d_bz_off2, &ctx->d_bz_status2, &ctx->d_bz_scratch,
			&ctx->d_bz_crc, &ctx->d_bz_x2n, &ctx->d_bz_seg, &ctx->d_bz_cnt, &ctx->d_bz_first, &ctx->d_bz_stop, &ctx->d_bz_bad, &ctx->d_bz_at, &ctx->d_bz_flag, &ctx->d_expected, &ctx->d_item_off, &ctx->d_item_len, &ctx->d_item_iv,
			&ctx->d_item_has_map, &ctx->d_item_first, &ctx->d_map_part, &ctx->d_support, &ctx->d_results,
			&ctx->d_bases, &ctx->d_row_tile, &ctx->d_depth_blocks, &ctx->d_support_base, &ctx->d_ref, &ctx->d_sat_start, &ctx->d_sat_end,
			&ctx->d_sr_pos, &ctx->d_sr_mapq, &ctx->d_sr_flag, &ctx->d_sr_lq, &ctx->d_sr_off, &ctx->d_sr_data,
			&ctx->d_sr_recoff, &ctx->d_refn, &ctx->d_kmer_keys, &ctx->d_kmer_sorted, &ctx->d_kmer_tmp, &ctx->d_kmer_offset, &ctx->d_kmer_pos};
	for (DevBuf *b : bufs)
		free_buf(*b);
	for (auto &s : ctx->staging) {
		if (s.pos)
			(void) hipHostFree(s.pos);
		if (s.mapq)
			(void) hipHostFree(s.mapq);
		if (s.copied)
			(void) hipEventDestroy(s.copied);
	}
	{
		void *pinned[] = {ctx->sr_stage.pos, ctx->sr_stage.mapq, ctx->sr_stage.flag, ctx->sr_stage.l_qseq,
				ctx->sr_stage.data_off, ctx->sr_stage.data};
		for (void *q : pinned)
			if (q)
				(void) hipHostFree(q);
	}
	if (ctx->h_small)
		(void) hipHostFree(ctx->h_small);
	if (ctx->h_head)
		(void) hipHostFree(ctx->h_head);
	bz_upload_quiesce(ctx, true);
	ctx->bz_job_kept.reset();
	for (int k = 0; k < 2; k++) {
		if (ctx->bz_ahead[k])
			(void) hipStreamDestroy(ctx->bz_ahead[k]);
		if (ctx->ev_bz_ahead[k])
			(void) hipEventDestroy(ctx->ev_bz_ahead[k]);
	}
	for (uint8_t *q : ctx->bz_up_buf)
		if (q)
			(void) hipFree(q);
	if (ctx->h_bz_ring)
		(void) hipHostFree(ctx->h_bz_ring);
	for (hipEvent_t e : ctx->ev_bz_slot)
		if (e)
			(void) hipEventDestroy(e);
	for (hipEvent_t e : ctx->ev_bz_kernel)
		if (e)
			(void) hipEventDestroy(e);
	if (ctx->bz_copy)
		(void) hipStreamDestroy(ctx->bz_copy);
	if (ctx->bz_third_maker.joinable())
		ctx->bz_third_maker.join();
	if (ctx->bz_shared) { // (two of them are `stream2` and `stream`, destroyed as such; the event of the third is in ev_bz_kernel when it was taken up)
		if (ctx->bz_third)
			(void) hipStreamDestroy(ctx->bz_third);
		if (ctx->ev_bz_third && ctx->ev_bz_kernel[2] != ctx->ev_bz_third)
			(void) hipEventDestroy(ctx->ev_bz_third);
	} else
		for (hipStream_t q : ctx->bz_kernel)
			if (q)
				(void) hipStreamDestroy(q);
	if (ctx->ev_head)
		(void) hipEventDestroy(ctx->ev_head);
	if (ctx->h_results)
		(void) hipHostFree(ctx->h_results);
	if (ctx->ev_done)
		(void) hipEventDestroy(ctx->ev_done);
	for (int k = 0; k < CONGA_K_COUNT; k++) {
		if (ctx->ev_k0[k])
			(void) hipEventDestroy(ctx->ev_k0[k]);
		if (ctx->ev_k1[k])
			(void) hipEventDestroy(ctx->ev_k1[k]);
	}
	if (ctx->stream)
		(void) hipStreamDestroy(ctx->stream);
	delete ctx;
}

int conga_reset(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	reset_slots(ctx);
	return CONGA_OK;
}

int conga_chrom_begin(conga_ctx *ctx, int64_t chrom_len, const uint8_t *gc_hist_w, const uint8_t *gc_like_w,
		int64_t n_win)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (chrom_len <= 0 || chrom_len > (int64_t) INT32_MAX - 2 * kDepthMaxTile || !gc_hist_w || !gc_like_w)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_begin: bad length or null GC array");
	const int32_t step = ctx->step;
	if (n_win != (chrom_len + step - 1) / step)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_begin: n_win must be ceil(chrom_len / gc_step)");
	if (ctx->staging_cur >= 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_begin: a staging buffer is handed out and not committed");
	{
		// the reference indexes its 101-entry tables with these (read_distribution.c:71-72, likelihood.c:118): a value
		// above 100 would read and write past them there
		uint8_t top = 0;
		for (int64_t w = 0; w < n_win; w++)
			top = std::max(top, std::max(gc_hist_w[w], gc_like_w[w]));
		if (top >= kGcBins)
			return fail(ctx, CONGA_ERR_RANGE, "conga_chrom_begin: GC value above 100");
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (!batch_mode(ctx)) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		reset_slots(ctx);
	}
	HostSlot h;
	h.L = chrom_len;
	h.n_win = n_win;
	const int64_t T = ctx->tile_len;
	h.n_tiles = (chrom_len + T - 1) / T;
	h.read_off = ctx->n_reads_total;
	h.gc_hist.assign(gc_hist_w, gc_hist_w + n_win);
	if (gc_like_w != gc_hist_w && memcmp(gc_like_w, gc_hist_w, (size_t) n_win) != 0)
		h.gc_like.assign(gc_like_w, gc_like_w + n_win);
	ctx->slots.push_back(std::move(h));
	ctx->cur = (int) ctx->slots.size() - 1;
	ctx->read_target = -1; // reads stream into the chromosome begun last
	ctx->layout_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_chrom_count(const conga_ctx *ctx)
{
	return ctx ? (int) ctx->slots.size() : 0;
}

int conga_chrom_select(conga_ctx *ctx, int index)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (index < 0 || index >= (int) ctx->slots.size())
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_select: no such chromosome");
	ctx->cur = index;
	return CONGA_OK;
}

int conga_reads_staging(conga_ctx *ctx, conga_read_staging *out)
{
	if (!ctx || !out)
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty())
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_staging: no chromosome open");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	Staging &s = ctx->staging[ctx->staging_next];
	if (!s.pos) {
		HIP_TRY(ctx, hipHostMalloc((void **) &s.pos, kStagingTuples * sizeof(int32_t), hipHostMallocDefault));
		HIP_TRY(ctx, hipHostMalloc((void **) &s.mapq, kStagingTuples, hipHostMallocDefault));
		HIP_TRY(ctx, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
	}
	if (s.in_flight) {
		HIP_TRY(ctx, hipEventSynchronize(s.copied));
		s.in_flight = false;
	}
	ctx->staging_cur = ctx->staging_next;
	out->pos = s.pos;
	out->mapq = s.mapq;
	out->capacity = kStagingTuples;
	return CONGA_OK;
}

int conga_reads_commit(conga_ctx *ctx, size_t n)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty() || ctx->staging_cur < 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_commit: call conga_reads_staging first");
	if (n > kStagingTuples)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_commit: n exceeds the staging capacity");
	// reads stream into the chromosome begun last (BAM order), or the one conga_sample_chrom() named
	HostSlot &h = ctx->read_target >= 0 ? ctx->slots[(size_t) ctx->read_target] : ctx->slots.back();
	if (n && h.device_fed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_commit: this chromosome's reads came from conga_reads_bgzf");
	if ((uint64_t) ctx->n_reads_total + n >= 0xFFFFFFF0ull)
		return fail(ctx, CONGA_ERR_RANGE, "conga_reads_commit: more than 2^32 reads in one context");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	Staging &s = ctx->staging[ctx->staging_cur];
	ctx->staging_cur = -1;
	if (n == 0)
		return CONGA_OK;
	const size_t total = (size_t) ctx->n_reads_total + n;
	if (total * 4 > ctx->d_pos.cap || total > ctx->d_mapq.cap) {
		const size_t want = std::max(total, (size_t) 1 << 22);
		TRY(ensure(ctx, ctx->d_pos, want * 4, true));
		TRY(ensure(ctx, ctx->d_mapq, want, true));
	}
	note_equal_runs(ctx, h, s.pos, n);
	HIP_TRY(ctx, hipMemcpyAsync(ptr<int32_t>(ctx->d_pos) + ctx->n_reads_total, s.pos, n * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ptr<uint8_t>(ctx->d_mapq) + ctx->n_reads_total, s.mapq, n, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipEventRecord(s.copied, ctx->stream));
	s.in_flight = true;
	h.n_reads += (int64_t) n;
	ctx->n_reads_total += (int64_t) n;
	ctx->staging_next = (ctx->staging_next + 1) % kStagingRing;
	ctx->sample_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

// ---- cohort mode: another sample's reads behind the layout the context already holds ------------------------------

void *conga_host_alloc(conga_ctx *ctx, size_t bytes)
{
	if (!ctx || bytes == 0 || hipSetDevice(ctx->device) != hipSuccess)
		return nullptr;
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		(void) hipGetLastError();
		ctx->err = "conga_host_alloc: hipHostMalloc failed";
		return nullptr;
	}
	return p;
}

void conga_host_free(conga_ctx *ctx, void *p)
{
	if (!ctx || !p)
		return;
	(void) hipSetDevice(ctx->device);
	(void) hipHostFree(p);
}

namespace {

// Forget the reads of every chromosome; chromosomes, GC arrays, intervals, tracks and the device layout stay.
int drop_reads(conga_ctx *ctx, const char *who, bool keep_computed = false)
{
	if (ctx->slots.empty())
		return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": no chromosome open");
	if (ctx->staging_cur >= 0)
		return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": a staging buffer is handed out and not committed");
	for (HostSlot &h : ctx->slots) {
		h.read_off = 0;
		h.n_reads = 0;
		h.device_fed = false;
		h.tail_val = 0;
		h.tail_len = 0;
		// the split-read records are the sample's too; reference sequences, satellites and the 10-mer indexes are the layout's
		h.sr_off = 0;
		h.n_sr = 0;
		h.sr_inplace = false;
	}
	ctx->n_reads_total = 0;
	ctx->n_sr_total = 0;
	ctx->sr_bytes_total = 0;
	ctx->sr_staged = false;
	ctx->bz_keep_bytes = 0;
	ctx->expand_pending = false; // (differences that no compute has taken up go with the reads they stood for)
	ctx->wrap_risk = false;
	ctx->sample_dirty = true;
	if (keep_computed && ctx->computed)
		ctx->reads_ahead = true; // (the records of the last compute stay fetchable: conga_sample_reads has left its inputs alone)
	else {
		ctx->depth_resident = false;
		ctx->computed = false;
		ctx->reads_ahead = false;
	}
	return CONGA_OK;
}

} // namespace

int conga_sample_begin(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	TRY(drop_reads(ctx, "conga_sample_begin"));
	ctx->read_target = 0;
	return CONGA_OK;
}

int conga_sample_chrom(conga_ctx *ctx, int index)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (index < 0 || index >= (int) ctx->slots.size())
		return fail(ctx, CONGA_ERR_INVALID, "conga_sample_chrom: no such chromosome");
	if (ctx->staging_cur >= 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_sample_chrom: a staging buffer is handed out and not committed");
	// the tuples of a context lie in chromosome order: no later chromosome may have reads yet
	for (size_t c = (size_t) index + 1; c < ctx->slots.size(); c++)
		if (ctx->slots[c].n_reads != 0)
			return fail(ctx, CONGA_ERR_INVALID, "conga_sample_chrom: a later chromosome already has reads (ascending order only)");
	ctx->read_target = index;
	return CONGA_OK;
}

} // extern "C"

namespace {

// conga_sample_reads / conga_sample_reads_d16: `pos` (32-bit positions) or `delta` + exceptions (16-bit differences)
int sample_reads_impl(conga_ctx *ctx, const char *who, const int32_t *pos, const uint8_t *delta, int width, const uint32_t *esc_index,
		const int32_t *esc_pos, size_t n_esc, const uint8_t *mapq, const uint64_t *chrom_off, int n_chrom)
{
	if (!ctx || !chrom_off)
		return CONGA_ERR_INVALID;
	if (n_chrom != (int) ctx->slots.size())
		return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": n_chrom differs from the chromosomes the context holds");
	const uint64_t total = chrom_off[n_chrom];
	// With the default threshold (-1: cmdline.c:188-194) every read passes `qual > mq_threshold` (bam_data.c:205) whatever its
	// MAPQ: the bytes are never looked at, so they are not sent either (4 bytes per read over PCIe instead of 5) and may be NULL.
	const bool need_mapq = ctx->opts.mq_threshold >= 0;
	const bool packed = delta != nullptr;
	if (chrom_off[0] != 0 || (total && ((!pos && !delta) || (need_mapq && !mapq))))
		return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": chrom_off must start at 0 and the arrays must be given");
	for (int c = 0; c < n_chrom; c++)
		if (chrom_off[c + 1] < chrom_off[c])
			return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": chrom_off must not decrease");
	if (total >= 0xFFFFFFF0ull)
		return fail(ctx, CONGA_ERR_RANGE, std::string(who) + ": more than 2^32 reads in one context");
	if (packed && (width < 4 || width > 16))
		return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": differences are 4 to 16 bits wide");
	// exceptions behind the differences in ONE buffer (esc_index == NULL): [differences | up to the next multiple of 16 bytes |
	// esc_index[n_esc] | esc_pos[n_esc]] -- one DMA per sample instead of three (each carries tens of microseconds of its own)
	const size_t d_bytes_host = packed ? ((size_t) total + 7) / 8 * (size_t) width : 0;
	const size_t esc_at = (d_bytes_host + 15) & ~(size_t) 15;
	const bool inline_esc = packed && n_esc > 0 && !esc_index && !esc_pos;
	if (inline_esc) {
		esc_index = reinterpret_cast<const uint32_t *>(delta + esc_at);
		esc_pos = reinterpret_cast<const int32_t *>(delta + esc_at) + n_esc;
	}
	if (packed) {
		// the exceptions: sorted by index, one for the first read of every chromosome that has reads (nothing can be carried
		// over a chromosome's border)
		if (n_esc > 0xFFFFFFF0ull || (n_esc && (!esc_index || !esc_pos)))
			return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": the exception list is missing");
		// (no early way out of the loop: it is on the hand-over's critical path -- the copy is enqueued behind it -- and a narrow
		// width has hundreds of thousands of exceptions; without a branch the compiler makes it a vector loop)
		uint32_t bad = n_esc && esc_index[0] >= total ? 1u : 0u;
		for (size_t k = 1; k < n_esc; k++)
			bad |= (uint32_t) (esc_index[k] >= total) | (uint32_t) (esc_index[k] <= esc_index[k - 1]);
		if (bad)
			return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": exceptions must be sorted by index and lie inside the reads");
		for (int c = 0; c < n_chrom; c++)
			if (chrom_off[c + 1] > chrom_off[c]
					&& !std::binary_search(esc_index, esc_index + n_esc, (uint32_t) chrom_off[c]))
				return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": the first read of every chromosome must be an exception (an absolute position)");
	}
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (ctx->reads_ahead) // (two samples handed over without a compute in between: the first one's copy must not be overtaken)
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2));
	const bool ahead = ctx->computed && !ctx->reads_ahead && ctx->n_sr_total == 0 && ctx->bz_keep_bytes == 0;
	TRY(drop_reads(ctx, who, ahead));
	ctx->read_target = -1;
	if (ahead) {
		// The last compute (and the fetch that may still follow it) reads d_pos / d_mapq: this sample's tuples go into the
		// other pair, on stream2, behind the last compute that read THAT pair.  A caller that hands over sample k + 1, fetches
		// sample k and then computes sample k + 1 has the copy running beside the kernels and the fetch with one context.
		std::swap(ctx->d_pos, ctx->d_pos_alt);
		std::swap(ctx->d_mapq, ctx->d_mapq_alt);
		ctx->pos_buf ^= 1;
	}
	if (total * 4 > ctx->d_pos.cap || total > ctx->d_mapq.cap) {
		const size_t want = std::max((size_t) total + (size_t) total / 8, (size_t) 1 << 22); // (samples of a cohort differ a little)
		TRY(ensure(ctx, ctx->d_pos, want * 4));
		TRY(ensure(ctx, ctx->d_mapq, want));
	}
	if (total) {
		// Straight from the caller's arrays: from pinned memory (conga_host_alloc) this is one DMA each at the link's rate.
		// The packed form always travels on stream2 (one staging buffer for the differences: the stream keeps its users in order).
		const bool on2 = ahead || packed;
		hipStream_t cs = on2 ? ctx->stream2 : ctx->stream;
		if (on2 && ctx->used_recorded[ctx->pos_buf])
			HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->ev_pair[ctx->pos_buf], 0));
		if (!packed)
			HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pos.p, pos, (size_t) total * 4, hipMemcpyHostToDevice, cs));
		else {
			// only the copies here: the differences are turned into positions by the compute that takes them up, on ITS stream
			// (expand_positions), so that the next sample's copy follows this one's without a kernel in between
			DevBuf &dd = ctx->d_delta[ctx->pos_buf], &de = ctx->d_delta_esc[ctx->pos_buf];
			const size_t d_bytes = ((size_t) total + 7) / 8 * (size_t) width; // (eight differences are `width` whole bytes)
			TRY(ensure(ctx, dd, esc_at + n_esc * 8 + 64));
			TRY(ensure(ctx, de, std::max<size_t>(n_esc, 1) * 8));
			TRY(ensure(ctx, ctx->d_delta_agg, (size_t) ((total + kDeltaChunk - 1) / kDeltaChunk) * 16 + 16)); // (aggregates, carries, exception ranks)
			uint32_t *d_ei = ptr<uint32_t>(de);
			if (inline_esc) // differences and exceptions in one go; the expansion finds the exceptions behind the differences
				HIP_TRY(ctx, hipMemcpyAsync(dd.p, delta, esc_at + n_esc * 8, hipMemcpyHostToDevice, cs));
			else {
				HIP_TRY(ctx, hipMemcpyAsync(dd.p, delta, std::min(d_bytes, ((size_t) total * (size_t) width + 7) / 8), hipMemcpyHostToDevice, cs));
				if (n_esc) {
					HIP_TRY(ctx, hipMemcpyAsync(d_ei, esc_index, n_esc * 4, hipMemcpyHostToDevice, cs));
					HIP_TRY(ctx, hipMemcpyAsync(d_ei + n_esc, esc_pos, n_esc * 4, hipMemcpyHostToDevice, cs));
				}
			}
			ctx->expand_esc_at = inline_esc ? esc_at : (size_t) -1;
			ctx->expand_pending = true;
			ctx->expand_total = total;
			ctx->expand_n_esc = n_esc;
			ctx->expand_width = width;
		}
		if (need_mapq)
			HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mapq.p, mapq, (size_t) total, hipMemcpyHostToDevice, cs));
		if (on2) {
			HIP_TRY(ctx, hipEventRecord(ctx->ev_reads, cs));
			ctx->reads_on_stream2 = true;
		}
	}
	for (int c = 0; c < n_chrom; c++)
		ctx->slots[(size_t) c].n_reads = (int64_t) (chrom_off[c + 1] - chrom_off[c]);
	ctx->n_reads_total = (int64_t) total;
	return CONGA_OK;
}

} // namespace

extern "C" {

int conga_sample_reads(conga_ctx *ctx, const int32_t *pos, const uint8_t *mapq, const uint64_t *chrom_off, int n_chrom)
{
	return sample_reads_impl(ctx, "conga_sample_reads", pos, nullptr, 0, nullptr, nullptr, 0, mapq, chrom_off, n_chrom);
}

int conga_sample_reads_packed(conga_ctx *ctx, const uint8_t *bits, int width, const uint32_t *esc_index, const int32_t *esc_pos, size_t n_esc,
		const uint8_t *mapq, const uint64_t *chrom_off, int n_chrom)
{
	if (!bits && chrom_off && n_chrom >= 0 && chrom_off[n_chrom] != 0)
		return CONGA_ERR_INVALID;
	static const uint8_t none[16] = {0};
	return sample_reads_impl(ctx, "conga_sample_reads_packed", nullptr, bits ? bits : none, width, esc_index, esc_pos, n_esc, mapq, chrom_off, n_chrom);
}

// ---- the producer of the packed form, on the host (pack_host.h)
struct conga_packer {
	conga_pack::Packer impl;
	explicit conga_packer(int n) : impl(n) {}
};

conga_packer *conga_packer_create(int n_threads)
{
	if (n_threads <= 0)
		n_threads = (int) std::max(1u, cpus_allowed() / 2);
	return new (std::nothrow) conga_packer(std::min(n_threads, 64));
}

void conga_packer_destroy(conga_packer *p)
{
	delete p;
}

int conga_packer_threads(const conga_packer *p)
{
	return p ? p->impl.threads() : 0;
}

size_t conga_pack_bound(uint64_t n_reads, size_t max_esc)
{
	return conga_pack::bound(n_reads, max_esc);
}

int conga_packer_start(conga_packer *p, const int32_t *pos, const uint64_t *chrom_off, int n_chrom, int width, uint8_t *out, size_t out_cap)
{
	if (!p)
		return CONGA_ERR_INVALID;
	const int rc = p->impl.start(pos, chrom_off, n_chrom, width, out, out_cap);
	return rc == 0 ? CONGA_OK : rc == -4 ? CONGA_ERR_NOMEM : CONGA_ERR_INVALID;
}

int conga_packer_finish(conga_packer *p, int *width, size_t *n_esc, size_t *out_bytes)
{
	if (!p)
		return CONGA_ERR_INVALID;
	const int rc = p->impl.finish(width, n_esc, out_bytes);
	return rc == 0 ? CONGA_OK : rc == -4 ? CONGA_ERR_NOMEM : CONGA_ERR_INVALID;
}

int conga_sample_reads_d16(conga_ctx *ctx, const uint16_t *delta, const uint32_t *esc_index, const int32_t *esc_pos, size_t n_esc,
		const uint8_t *mapq, const uint64_t *chrom_off, int n_chrom)
{
	return conga_sample_reads_packed(ctx, reinterpret_cast<const uint8_t *>(delta), 16, esc_index, esc_pos, n_esc, mapq, chrom_off, n_chrom);
}

} // extern "C"

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
	const char *ov = getenv("CONGA_BGZF_OVERLAP");
	if (lane_kernel_asked() || !(ov ? atoi(ov) != 0 : n_bytes >= ((size_t) 96 << 20)) || n_bytes == 0 || getenv("CONGA_BGZF_NO_AHEAD"))
		return CONGA_OK;
	std::lock_guard<std::mutex> g(ctx->bz_up_mu);
	if (!ctx->h_bz_ring || ctx->bz_ring_failed || !ctx->bz_copy || ctx->bz_named.size() >= 3 || ctx->bz_up_quit)
		return CONGA_OK;
	ByteSource src;
	src.fd = fd;
	src.file_off = file_off;
	// (inside a call: behind that call's bytes, at once.  Between calls -- or before the call in front of these bytes' own has
	// begun, which a cohort's planning thread cannot know --: when the next call begins, behind its bytes or as its bytes)
	// (only behind a job that is itself on its way: what was named first goes up first)
	const bool now = ctx->bz_in_call && (ctx->bz_named.empty() || ctx->bz_named.back()->queued);
	std::shared_ptr<BzJob> job = bz_queue_job(ctx, src, n_bytes, false);
	if (n_known && known_starts[0] == 0 && !getenv("CONGA_BGZF_NO_TABLE")) {
		job->build_table = true;
		job->known.assign(known_starts, known_starts + n_known);
		job->stop_at = stop_at;
		job->piece_tables.resize(job->n_pieces);
		job->cap_blocks = n_bytes / 4096 + 65536; // (BAM writers fill a block with ~64 KB of records: 10-30 KB deflated)
		job->blocks.reserve(job->cap_blocks);     // (the inflate-ahead thread reads what is published while the upload thread appends)
		job->out_off.reserve(job->cap_blocks);
	}
	if (now)
		bz_enqueue(ctx, job);
	ctx->bz_named.push_back(job);
	job->ticket = *ticket = ++ctx->bz_up_tickets;
	return CONGA_OK;
}

int conga_reads_bgzf_next_blocks(conga_ctx *ctx, uint64_t ticket, const conga_bgzf_block *blocks, size_t n_blocks)
{
	if (!ctx || !blocks || n_blocks == 0 || n_blocks > (size_t) 1 << 28)
		return CONGA_ERR_INVALID;
	if (ticket == 0 || getenv("CONGA_BGZF_NO_INFLATE_AHEAD") || (getenv("CONGA_BGZF_KERNEL") && strcmp(getenv("CONGA_BGZF_KERNEL"), "wave") != 0))
		return CONGA_OK;
	// (one spare output set: one named job at a time is inflated ahead -- the set is free again when the call that takes those
	// bytes up has swapped it in)
	std::lock_guard<std::mutex> g(ctx->bz_up_mu);
	std::shared_ptr<BzJob> job;
	for (const std::shared_ptr<BzJob> &j : ctx->bz_named) {
		std::lock_guard<std::mutex> gj(j->mu);
		if (j->ticket == ticket)
			job = j;
		else if (j->inflate_asked)
			return CONGA_OK;
	}
	if (!job || !ctx->d_bz_x2n.p || !ctx->d_bz_crc.p) // (taken up already, or no call of this context has inflated anything yet)
		return CONGA_OK;
	// the table as conga_reads_bgzf* checks it: in file order, inside the bytes, 1..64 KiB each
	std::vector<uint64_t> out_off(n_blocks);
	uint64_t total = 0;
	for (size_t b = 0; b < n_blocks; b++) {
		const conga_bgzf_block &bl = blocks[b];
		if (bl.data_off > job->n_bytes || (uint64_t) bl.data_len > job->n_bytes - bl.data_off || bl.inflated_len == 0 || bl.inflated_len > 65536u
				|| (b && bl.data_off < blocks[b - 1].data_off + blocks[b - 1].data_len))
			return CONGA_OK; // (the call itself will say what is wrong with it)
		out_off[b] = total;
		total += bl.inflated_len;
	}
	std::lock_guard<std::mutex> gj(job->mu);
	if (job->inflate_asked || job->cancel.load())
		return CONGA_OK;
	if (job->build_table)
		return CONGA_OK; // (the engine reads the table off the bytes itself)
	job->blocks.assign(blocks, blocks + n_blocks);
	job->out_off.swap(out_off);
	job->total_out = total;
	job->table_n = n_blocks;
	job->table_final = job->table_ok = true;
	job->inflate_asked = true;
	// (the spare output set goes to the oldest ticket that WAITS for it, bz_inflate_ahead: a job whose table the caller brought
	// never passes bz_run_job's `ahead` branch, so it is entered here -- without this its thread slept for ever and the call that
	// adopted the job with it)
	ctx->bz_spare_waiting.insert(job->ticket);
	job->inflater = std::thread(bz_inflate_ahead, ctx, job);
	return CONGA_OK;
}

int conga_reads_bgzf_next_table(conga_ctx *ctx, uint64_t ticket, const conga_bgzf_block **blocks, size_t *n_blocks)
{
	if (!ctx || !blocks || !n_blocks)
		return CONGA_ERR_INVALID;
	*blocks = nullptr;
	*n_blocks = 0;
	std::shared_ptr<BzJob> job;
	{
		std::unique_lock<std::mutex> lk(ctx->bz_up_mu);
		for (const std::shared_ptr<BzJob> &j : ctx->bz_named)
			if (ticket != 0 && j->ticket == ticket)
				job = j;
		if (!job || !job->build_table)
			return CONGA_OK;
		// Bytes named before the call in front of theirs has begun go up when it does -- a cohort's planning thread names sample
		// k + 1 a few milliseconds before the call for sample k begins.  Not for ever: with no such call to come (the caller
		// decodes sample k on the host after all) there will be no table, and the caller must not wait for one.
		ctx->bz_up_cv.wait_for(lk, std::chrono::milliseconds(400), [&] { return job->queued; });
		if (!job->queued)
			return CONGA_OK;
	}
	std::unique_lock<std::mutex> lk(job->mu);
	job->cv.wait(lk, [&] { return job->table_final || job->done; });
	if (job->table_final && job->table_ok) {
		*blocks = job->blocks.data();
		*n_blocks = job->blocks.size();
	}
	return CONGA_OK;
}

int conga_reads_bgzf_forget(conga_ctx *ctx, uint64_t ticket)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	std::shared_ptr<BzJob> job;
	{
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		for (size_t k = 0; ticket != 0 && k < ctx->bz_named.size(); k++)
			if (ctx->bz_named[k]->ticket == ticket) {
				job = ctx->bz_named[k];
				ctx->bz_named.erase(ctx->bz_named.begin() + (long) k);
				break;
			}
	}
	bz_abandon(ctx, job); // (returns when the upload thread no longer reads from the descriptor)
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
	{
		std::lock_guard<std::mutex> g(ctx->bz_up_mu);
		ctx->bz_ratio = std::max(ctx->bz_ratio, (double) total / (double) std::max<size_t>(n_bytes, 1));
	}
	const bool timing = getenv("CONGA_TIMING") != nullptr;
	const auto t_begin = std::chrono::steady_clock::now();
	auto ms_since = [](std::chrono::steady_clock::time_point t) {
		return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
	};
	// one decoder scratch (17 KB) per lane; a lane takes several blocks only beyond 131 072 of them (2.2 GB of scratch)
	uint32_t lanes = (uint32_t) std::min<size_t>((n_blocks + 63) & ~(size_t) 63, 131072);
	if (const char *e = getenv("CONGA_BGZF_LANES")) // (tests: few lanes, several blocks each)
		lanes = std::min(lanes, (uint32_t) std::max(64, atoi(e) & ~63));
	// the bytes and their inflate: overlapped (pinned pieces, several launches) for a piece of the file worth it and blocks in
	// file order; otherwise one copy, one launch
	bool in_order = true;
	for (size_t b = 1; b < n_blocks && in_order; b++)
		in_order = blocks[b].data_off >= blocks[b - 1].data_off + blocks[b - 1].data_len;
	const char *ov = getenv("CONGA_BGZF_OVERLAP"); // (0 / 1 forces; tests run both forms on small files)
	const bool overlapped = !lane_kernel_asked() && in_order && (ov ? atoi(ov) != 0 : n_bytes >= ((size_t) 96 << 20));
	if (!overlapped) // (the overlapped form has device buffers of its own for the compressed bytes: the upload jobs')
		TRY(ensure(ctx, ctx->d_bz_in, n_bytes + 512)); // (the decoders read ahead of their position: up to 64 dwords)
	TRY(ensure(ctx, ctx->d_bz_blocks, n_blocks * sizeof(conga_bgzf_block)));
	TRY(ensure(ctx, ctx->d_bz_off, n_blocks * 8));
	TRY(ensure(ctx, ctx->d_bz_out, (size_t) (base + total) + 64, base > 0));
	TRY(ensure(ctx, ctx->d_bz_status, n_blocks));
	TRY(ensure(ctx, ctx->d_bz_seg, n_segments * sizeof(conga_bam_segment)));
	TRY(ensure(ctx, ctx->d_bz_cnt, n_segments * 4));
	TRY(ensure(ctx, ctx->d_bz_first, n_segments * 8));
	TRY(ensure(ctx, ctx->d_bz_stop, n_segments * 8));
	TRY(ensure(ctx, ctx->d_bz_bad, n_segments));
	TRY(ensure(ctx, ctx->d_bz_at, n_segments * 8));
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
		~InCall()
		{
			std::lock_guard<std::mutex> g(c->bz_up_mu);
			c->bz_in_call = false;
		}
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
	std::vector<uint8_t> status(n_blocks), bad(n_segments);
	std::vector<uint32_t> count(n_segments);
	std::vector<uint64_t> v_first(n_segments), v_stop(n_segments);
	HIP_TRY(ctx, hipMemcpyAsync(status.data(), ctx->d_bz_status.p, n_blocks, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(bad.data(), ctx->d_bz_bad.p, n_segments, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(count.data(), ctx->d_bz_cnt.p, n_segments * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(v_first.data(), ctx->d_bz_first.p, n_segments * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipMemcpyAsync(v_stop.data(), ctx->d_bz_stop.p, n_segments * 8, hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipStreamSynchronize(st));
	for (size_t b = 0; b < n_blocks; b++)
		if (status[b] != kBgzfOk)
			return fail(ctx, CONGA_ERR_DATA, status[b] == kBgzfCrc ? "conga_reads_bgzf: a block fails its CRC32"
					: "conga_reads_bgzf: a block does not inflate to its recorded size");
	std::vector<uint64_t> write_at(n_segments);
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
			const size_t want = std::max(total_reads, (size_t) 1 << 22);
			TRY(ensure(ctx, ctx->d_pos, want * 4, true));
			TRY(ensure(ctx, ctx->d_mapq, want, true));
		}
		if (want_rec) {
			TRY(ensure(ctx, ctx->d_sr_recoff, std::max(total_reads, (size_t) 1 << 22) * 8, true));
			w.rec_off = ptr<uint64_t>(ctx->d_sr_recoff);
		}
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_bz_at.p, write_at.data(), n_segments * 8, hipMemcpyHostToDevice, st));
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

int conga_mappability(conga_ctx *ctx, const int32_t *start, const int32_t *end, const float *val, size_t m)
{
	if (!ctx || (m && (!start || !end || !val)))
		return CONGA_ERR_INVALID;
	HostSlot *h = current(ctx);
	if (!h)
		return fail(ctx, CONGA_ERR_INVALID, "conga_mappability: no chromosome open");
	if (m > (size_t) INT32_MAX)
		return fail(ctx, CONGA_ERR_RANGE, "conga_mappability: too many rows");
	// sorted, abutting-at-most rows can be painted in one pass (kernels.hip.h: paint_sorted_kernel)
	bool sorted = true;
	for (size_t k = 0; k < m && sorted; k++) {
		if (end[k] < start[k])
			sorted = false;
		if (k + 1 < m && (start[k + 1] < end[k] || start[k + 1] < start[k]))
			sorted = false;
	}
	h->map_sorted = sorted;
	h->has_map = true;
	h->map_start.assign(start, start + m);
	h->map_end.assign(end, end + m);
	h->map_val.assign(val, val + m);
	ctx->layout_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_intervals(conga_ctx *ctx, char type, const int32_t *start, const int32_t *end, size_t n)
{
	if (!ctx || (n && (!start || !end)))
		return CONGA_ERR_INVALID;
	const int t = type_index(type);
	if (t < 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_intervals: type must be 'D' or 'E'");
	HostSlot *h = current(ctx);
	if (!h)
		return fail(ctx, CONGA_ERR_INVALID, "conga_intervals: no chromosome open");
	if (n > (size_t) 1 << 28)
		return fail(ctx, CONGA_ERR_RANGE, "conga_intervals: too many intervals");
	for (size_t i = 0; i < n; i++) {
		// the reference would read before/after its arrays for such rows
		if (start[i] < 0 || end[i] < start[i])
			return fail(ctx, CONGA_ERR_RANGE, "conga_intervals: interval with start < 0 or end < start");
	}
	h->iv_start[t].assign(start, start + n);
	h->iv_end[t].assign(end, end + n);
	h->iv_support[t].clear();
	ctx->layout_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_reference(conga_ctx *ctx, const char *seq, int64_t len)
{
	if (!ctx || !seq)
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty())
		return fail(ctx, CONGA_ERR_INVALID, "conga_reference: no chromosome open");
	HostSlot &h = ctx->slots.back();
	if (len != h.L)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reference: length differs from the chromosome length");
	// (readReferenceSeq upper-cases every base, common.c:449: ref_pack_kernel does that on the device while it packs the text)
	h.ref.assign(reinterpret_cast<const uint8_t *>(seq), reinterpret_cast<const uint8_t *>(seq) + len);
	ctx->sr_layout.store(true);
	h.ref_version = ++ctx->ref_stamp;
	ctx->layout_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_satellites(conga_ctx *ctx, const int32_t *start, const int32_t *end, size_t n)
{
	if (!ctx || (n && (!start || !end)))
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty())
		return fail(ctx, CONGA_ERR_INVALID, "conga_satellites: no chromosome open");
	HostSlot &h = ctx->slots.back();
	// sort and merge, so that "any interval overlaps [a, b)" is one binary search on the device
	std::vector<std::pair<int32_t, int32_t>> iv;
	for (size_t i = 0; i < n; i++)
		if (end[i] > start[i])
			iv.emplace_back(start[i], end[i]);
	std::sort(iv.begin(), iv.end());
	h.sat_start.clear();
	h.sat_end.clear();
	for (const auto &x : iv) {
		if (!h.sat_end.empty() && x.first <= h.sat_end.back())
			h.sat_end.back() = std::max(h.sat_end.back(), x.second);
		else {
			h.sat_start.push_back(x.first);
			h.sat_end.push_back(x.second);
		}
	}
	ctx->layout_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

namespace {
constexpr size_t kSrStageReads = (size_t) 1 << 20;
constexpr size_t kSrStageBytes = (size_t) 192 << 20;
}

int conga_split_reads_staging(conga_ctx *ctx, conga_split_staging *out)
{
	if (!ctx || !out)
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty())
		return fail(ctx, CONGA_ERR_INVALID, "conga_split_reads_staging: no chromosome open");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	conga_split_staging &s = ctx->sr_stage;
	if (!s.pos) {
		HIP_TRY(ctx, hipHostMalloc((void **) &s.pos, kSrStageReads * 4, hipHostMallocDefault));
		HIP_TRY(ctx, hipHostMalloc((void **) &s.mapq, kSrStageReads, hipHostMallocDefault));
		HIP_TRY(ctx, hipHostMalloc((void **) &s.flag, kSrStageReads * 2, hipHostMallocDefault));
		HIP_TRY(ctx, hipHostMalloc((void **) &s.l_qseq, kSrStageReads * 4, hipHostMallocDefault));
		HIP_TRY(ctx, hipHostMalloc((void **) &s.data_off, kSrStageReads * 8, hipHostMallocDefault));
		HIP_TRY(ctx, hipHostMalloc((void **) &s.data, kSrStageBytes, hipHostMallocDefault));
		s.capacity_reads = kSrStageReads;
		s.capacity_bytes = kSrStageBytes;
	}
	ctx->sr_staged = true;
	*out = s;
	return CONGA_OK;
}

int conga_split_reads_commit(conga_ctx *ctx, size_t n_reads, size_t n_bytes)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty() || !ctx->sr_staged)
		return fail(ctx, CONGA_ERR_INVALID, "conga_split_reads_commit: call conga_split_reads_staging first");
	if (n_reads > kSrStageReads || n_bytes > kSrStageBytes)
		return fail(ctx, CONGA_ERR_INVALID, "conga_split_reads_commit: exceeds the staging capacity");
	ctx->sr_staged = false;
	if (n_reads == 0)
		return CONGA_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	conga_split_staging &st = ctx->sr_stage;
	// records stream into the chromosome begun last (BAM order), or the one conga_sample_chrom() named
	HostSlot &h = ctx->read_target >= 0 ? ctx->slots[(size_t) ctx->read_target] : ctx->slots.back();
	if (h.sr_inplace)
		return fail(ctx, CONGA_ERR_INVALID, "conga_split_reads_commit: this chromosome's records came from conga_reads_bgzf");
	if (h.n_sr > 0 && h.sr_off + h.n_sr != ctx->n_sr_total)
		return fail(ctx, CONGA_ERR_INVALID, "conga_split_reads_commit: a chromosome's records must be committed without another's in between");
	for (size_t i = 0; i < n_reads; i++) {
		const int32_t l = st.l_qseq[i];
		const uint64_t need = (uint64_t) (l < 0 ? 0 : l + 1) / 2 + (uint64_t) (l < 0 ? 0 : l);
		if (l < 0 || st.data_off[i] > n_bytes || need > n_bytes - st.data_off[i]) // (no sum that could wrap)
			return fail(ctx, CONGA_ERR_RANGE, "conga_split_reads_commit: record block outside the committed bytes");
		st.data_off[i] += (uint64_t) ctx->sr_bytes_total; // rebase into the device arena
		// (a negative position is no record of this chromosome -- the reference's iterator never returns one --: the kernel's
		// gate drops every position <= 0; find_split_reads returns at pos == 0, split_read.c:216)
	}
	const size_t nr = (size_t) ctx->n_sr_total + n_reads, nb = (size_t) ctx->sr_bytes_total + n_bytes;
	TRY(ensure(ctx, ctx->d_sr_pos, std::max(nr, (size_t) 1 << 20) * 4, true));
	TRY(ensure(ctx, ctx->d_sr_mapq, std::max(nr, (size_t) 1 << 20), true));
	TRY(ensure(ctx, ctx->d_sr_flag, std::max(nr, (size_t) 1 << 20) * 2, true));
	TRY(ensure(ctx, ctx->d_sr_lq, std::max(nr, (size_t) 1 << 20) * 4, true));
	TRY(ensure(ctx, ctx->d_sr_off, std::max(nr, (size_t) 1 << 20) * 8, true));
	TRY(ensure(ctx, ctx->d_sr_data, std::max(nb + 64, (size_t) 64 << 20), true)); // (the kernel reads a few bytes past a sequence)
	hipStream_t s = ctx->stream;
	const int64_t o = ctx->n_sr_total;
	HIP_TRY(ctx, hipMemcpyAsync(ptr<int32_t>(ctx->d_sr_pos) + o, st.pos, n_reads * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(ctx, hipMemcpyAsync(ptr<uint8_t>(ctx->d_sr_mapq) + o, st.mapq, n_reads, hipMemcpyHostToDevice, s));
	HIP_TRY(ctx, hipMemcpyAsync(ptr<uint16_t>(ctx->d_sr_flag) + o, st.flag, n_reads * 2, hipMemcpyHostToDevice, s));
	HIP_TRY(ctx, hipMemcpyAsync(ptr<int32_t>(ctx->d_sr_lq) + o, st.l_qseq, n_reads * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(ctx, hipMemcpyAsync(ptr<uint64_t>(ctx->d_sr_off) + o, st.data_off, n_reads * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(ctx, hipMemcpyAsync(ptr<uint8_t>(ctx->d_sr_data) + ctx->sr_bytes_total, st.data, n_bytes, hipMemcpyHostToDevice, s));
	HIP_TRY(ctx, hipStreamSynchronize(s)); // single staging set: it is the caller's again on return
	if (h.n_sr == 0)
		h.sr_off = ctx->n_sr_total;
	h.n_sr += (int64_t) n_reads;
	ctx->n_sr_total += (int64_t) n_reads;
	ctx->sr_bytes_total += (int64_t) n_bytes;
	ctx->sample_dirty = true; // (the records are the sample's: the layout -- references, indexes -- is untouched)
	ctx->computed = false;
	return CONGA_OK;
}

int conga_split_support(conga_ctx *ctx, char type, const int32_t *support, size_t n)
{
	if (!ctx || (n && !support))
		return CONGA_ERR_INVALID;
	const int t = type_index(type);
	HostSlot *h = current(ctx);
	if (t < 0 || !h || n != h->iv_start[t].size())
		return fail(ctx, CONGA_ERR_INVALID, "conga_split_support: type / count does not match conga_intervals");
	h->iv_support[t].assign(support, support + n);
	ctx->layout_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_chrom_compute(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (ctx->slots.empty())
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_compute: no chromosome open");
	if (ctx->staging_cur >= 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_compute: a staging buffer is handed out and not committed");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const auto t_prepare = std::chrono::steady_clock::now();
	const bool whole_layout = ctx->layout_dirty || ctx->layout_dense != dense_formulation(ctx);
	if (whole_layout)
		TRY(prepare_layout(ctx));
	else if (ctx->sample_dirty)
		TRY(prepare_sample(ctx));
	if (whole_layout && getenv("CONGA_TIMING"))
		fprintf(stderr, "[timing] conga_chrom_compute: layout prepared in %.1f ms (host tables, uploads, GC bases per bin; once per layout)\n",
				std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prepare).count());
	if (ctx->n_reads_total == 0) {
		TRY(ensure(ctx, ctx->d_pos, 256));
		TRY(ensure(ctx, ctx->d_mapq, 256));
	}
	// Formulation: tuple-space unless read_depth[] was asked for, the reads may be unsorted, or a `short` may wrap.
	const bool unsorted_mode = (ctx->opts.flags & CONGA_FLAG_READS_UNSORTED) != 0;
	const bool dense = unsorted_mode || (ctx->opts.flags & CONGA_FLAG_MATERIALIZE_DEPTH) != 0 || ctx->wrap_risk;
	if (dense) { // nothing may be allocated while the launches are being captured
		TRY(ensure(ctx, ctx->d_rd, std::max<size_t>((size_t) ctx->total_L, 8) * 2));
		if (!unsorted_mode)
			TRY(ensure(ctx, ctx->d_tile_start, ((size_t) ctx->total_tiles + 2) * 4));
	}
	for (int k = 0; k < CONGA_K_COUNT; k++)
		ctx->ev_used[k] = false;

	// The step is a fixed sequence of small launches on the same resident buffers.  With CONGA_GRAPH=1 in the
	// environment it is captured into a hipGraph on the third compute of an unchanged layout and replayed from then
	// on.  Off by default: on ROCm 7.2 / MI355X the replay was measured no faster than the two-stream launch sequence
	// (0.275 vs 0.268 ms per genome), and instantiation costs tens of milliseconds.
	hipStream_t st = ctx->stream;
	if (ctx->reads_on_stream2) { // the tuples came up on stream2 (conga_sample_reads beside the previous compute)
		HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_reads, 0));
		ctx->reads_on_stream2 = false;
	}
	if (ctx->expand_pending) { // ... as 16-bit differences (conga_sample_reads_d16): positions first -- delta16.hip.h
		const uint64_t total = ctx->expand_total;
		const uint32_t n_chunks = (uint32_t) ((total + kDeltaChunk - 1) / kDeltaChunk), n_esc = (uint32_t) ctx->expand_n_esc;
		const uint8_t *dd = ptr<uint8_t>(ctx->d_delta[ctx->pos_buf]);
		const uint32_t *d_ei = ctx->expand_esc_at == (size_t) -1 ? ptr<uint32_t>(ctx->d_delta_esc[ctx->pos_buf])
				: reinterpret_cast<const uint32_t *>(dd + ctx->expand_esc_at);
		const int32_t *d_ep = reinterpret_cast<const int32_t *>(d_ei + n_esc);
		int2 *d_agg = ptr<int2>(ctx->d_delta_agg);
		int32_t *d_carry = reinterpret_cast<int32_t *>(d_agg + n_chunks);
		uint32_t *d_rank = reinterpret_cast<uint32_t *>(d_carry + n_chunks);
		int32_t *d_pos = ptr<int32_t>(ctx->d_pos);
		KernelTimer t_expand(ctx, CONGA_K_EXPAND);
		auto launch = [&](auto width_tag) {
			constexpr int W = decltype(width_tag)::value;
			hipLaunchKernelGGL(delta_esc_rank_kernel, dim3((n_chunks + 256) / 256), dim3(256), 0, st, d_ei, n_esc, n_chunks, d_rank);
			hipLaunchKernelGGL(delta_aggregate_kernel<W>, dim3(n_chunks), dim3(256), 0, st, dd, total, d_ei, d_ep, n_esc, d_rank, d_agg);
			hipLaunchKernelGGL(delta_carry_kernel, dim3(1), dim3(1024), 0, st, d_agg, n_chunks, d_carry);
			hipLaunchKernelGGL(delta_expand_kernel<W>, dim3(n_chunks), dim3(256), 0, st, dd, total, d_ei, d_ep, n_esc, d_rank, d_carry, d_pos);
		};
		switch (ctx->expand_width) { // (any width from 4 to 16: eight differences are `width` whole bytes)
		case 4: launch(std::integral_constant<int, 4>()); break;
		case 5: launch(std::integral_constant<int, 5>()); break;
		case 6: launch(std::integral_constant<int, 6>()); break;
		case 7: launch(std::integral_constant<int, 7>()); break;
		case 8: launch(std::integral_constant<int, 8>()); break;
		case 9: launch(std::integral_constant<int, 9>()); break;
		case 10: launch(std::integral_constant<int, 10>()); break;
		case 11: launch(std::integral_constant<int, 11>()); break;
		case 12: launch(std::integral_constant<int, 12>()); break;
		case 13: launch(std::integral_constant<int, 13>()); break;
		case 14: launch(std::integral_constant<int, 14>()); break;
		case 15: launch(std::integral_constant<int, 15>()); break;
		default: launch(std::integral_constant<int, 16>()); break;
		}
		HIP_TRY(ctx, hipGetLastError());
		ctx->expand_pending = false;
	}
	ctx->computes_on_layout++;
	const bool use_graph = (ctx->opts.flags & CONGA_FLAG_PROFILE) == 0 && getenv("CONGA_GRAPH")
			&& (ctx->graph_exec || ctx->computes_on_layout >= 3);
	if (!use_graph)
		TRY(enqueue_compute(ctx, dense));
	else {
		if (!ctx->graph_exec || ctx->graph_dense != dense) {
			drop_graph(ctx);
			hipGraph_t graph = nullptr;
			HIP_TRY(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
			const int rc = enqueue_compute(ctx, dense);
			const hipError_t e = hipStreamEndCapture(st, &graph);
			if (rc != CONGA_OK || e != hipSuccess || !graph) {
				if (graph)
					(void) hipGraphDestroy(graph);
				(void) hipGetLastError();
				return rc != CONGA_OK ? rc : fail(ctx, CONGA_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
			}
			const hipError_t ei = hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0);
			(void) hipGraphDestroy(graph);
			if (ei != hipSuccess) {
				ctx->graph_exec = nullptr;
				return fail(ctx, CONGA_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
			}
			ctx->graph_dense = dense;
		}
		HIP_TRY(ctx, hipGraphLaunch(ctx->graph_exec, st));
	}
	ctx->depth_resident = dense;
	ctx->small_cur = ctx->small_cur_next; // the arena the chain launch has just cleared, if it did
	HIP_TRY(ctx, hipEventRecord(ctx->ev_done, st));
	HIP_TRY(ctx, hipEventRecord(ctx->ev_pair[ctx->pos_buf], st));
	ctx->used_recorded[ctx->pos_buf] = true;
	ctx->reads_ahead = false;
	ctx->computed_reads.resize(ctx->slots.size());
	for (size_t c = 0; c < ctx->slots.size(); c++)
		ctx->computed_reads[c] = std::make_pair(ctx->slots[c].read_off, ctx->slots[c].n_reads);
	ctx->computed_total = ctx->n_reads_total;
	HIP_TRY(ctx, hipGetLastError());
	ctx->computed = true;
	return CONGA_OK;
}

} // extern "C"

namespace {

// Every launch of one compute, in order, on ctx->stream (and ctx->stream2 for the forked interval count / reduce).
// Allocates nothing, waits for nothing: it can run under stream capture.
int enqueue_compute(conga_ctx *ctx, bool dense)
{
	hipStream_t st = ctx->stream;
	const int n_slots = (int) ctx->slots.size();
	Small *small = reinterpret_cast<Small *>(arena_of(ctx, ctx->small_cur));
	const Slot *dslots = ptr<Slot>(ctx->d_slots);
	const uint8_t *gc_like = ctx->gc_like_distinct ? ptr<uint8_t>(ctx->d_gc_like) : ptr<uint8_t>(ctx->d_gc_hist);
	const bool unsorted_mode = (ctx->opts.flags & CONGA_FLAG_READS_UNSORTED) != 0;

	if (!ctx->arena_zeroed[ctx->small_cur])
		HIP_TRY(ctx, hipMemsetAsync(small, 0, ctx->arena_bytes, st)); // Small blocks + observed[]
	ctx->arena_zeroed[ctx->small_cur] = false; // dirty from here on
	// The Small blocks are final after expected_table unless split-read kernels add their counters later: that
	// kernel then writes the pinned host copy itself (pinned host memory is device-visible) and the copy at the end goes away.
	const bool small_by_kernel = !ctx->any_sr;
	// Scoring inside the chain kernel: possible when nothing the score needs is produced beside the chain.
	const bool fused_score = !dense && !ctx->any_map_painted && ctx->n_iv > 0;

	// Second stream: work that does not depend on the main chain of kernels.  With per-kernel timing on
	// (CONGA_FLAG_PROFILE) everything stays on one stream so the event pairs bracket one kernel each.
	// (measured: a cross-stream event dependency costs more than a 15 us kernel, so stream2 is only used where it
	// hides a long one: interval_reduce beside the chain)
	const bool two_streams = (ctx->opts.flags & CONGA_FLAG_PROFILE) == 0 && ctx->n_items > 0 && (dense || ctx->any_map_painted);
	hipStream_t s2 = two_streams ? ctx->stream2 : st;
	bool s2_busy = false;      // something was put on stream2 that the main stream has not waited for yet
	bool count_pending = false; // ev_counted marks the end of interval_count on stream2
	// stream2 continues from this point of the main stream
	auto fork_to_s2 = [&](hipEvent_t ev) -> int {
		if (two_streams) {
			HIP_TRY(ctx, hipEventRecord(ev, st));
			HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ev, 0));
			s2_busy = true;
		}
		return CONGA_OK;
	};
	// the main stream waits for everything stream2 holds
	auto join_s2 = [&]() -> int {
		if (s2_busy) {
			HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
			HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
			s2_busy = false;
			count_pending = false;
		}
		return CONGA_OK;
	};
	const bool profile = (ctx->opts.flags & CONGA_FLAG_PROFILE) != 0;
	int next_arena = ctx->small_cur;
	if (!dense) {
		// tuple space: the per-interval read counts need nothing but the tuples.  Outside profiling they share ONE launch
		// with the pass over the tuples (tuple_pass_kernel); with a second stream in use they go there.
		const bool want_count = ctx->n_iv > 0 && ctx->n_items > 0 && ctx->n_reads_total > 0;
		CountArgs c;
		c.pos = ptr<int32_t>(ctx->d_pos);
		c.mapq = ptr<uint8_t>(ctx->d_mapq);
		c.item_slot = ptr<int32_t>(ctx->d_item_slot);
		c.slots = dslots;
		c.item_lo = ptr<int32_t>(ctx->d_item_lo);
		c.item_len = ptr<int32_t>(ctx->d_item_len);
		c.item_iv = ptr<int32_t>(ctx->d_item_iv);
		c.n_items = ctx->n_items;
		c.mq_threshold = ctx->opts.mq_threshold;
		c.observed = observed_of(ctx);
		const int count_grid = (int) ((ctx->n_items + 255) / 256);
		// ... and so do the mappability sums of the chromosomes whose track is summed in row space
		const bool want_rows = ctx->any_map_rows && ctx->n_iv > 0 && ctx->n_items > 0;
		MapRowsArgs mr;
		mr.row_start = ptr<int32_t>(ctx->d_map_start);
		mr.row_end = ptr<int32_t>(ctx->d_map_end);
		mr.row_val = ptr<float>(ctx->d_map_val);
		mr.item_row0 = ptr<uint32_t>(ctx->d_item_row0);
		mr.item_row1 = ptr<uint32_t>(ctx->d_item_row1);
		mr.row_tile = ptr<uint32_t>(ctx->d_row_tile);
		mr.item_rt_off = ptr<uint32_t>(ctx->d_item_rt_off);
		mr.item_lo = ptr<int32_t>(ctx->d_item_lo);
		mr.item_len = ptr<int32_t>(ctx->d_item_len);
		mr.item_has_map = ptr<uint8_t>(ctx->d_item_has_map);
		mr.n_items = ctx->n_items;
		mr.map_part = ptr<double>(ctx->d_map_part);
		const int map_grid = (int) ((ctx->n_items + kMapRowsItemsPerBlock - 1) / kMapRowsItemsPerBlock);
		const bool fuse = !profile && !two_streams && ctx->n_reads_total > 0;
		if (want_count && !fuse) {
			TRY(fork_to_s2(ctx->ev_fork));
			KernelTimer t(ctx, CONGA_K_COUNT_READS);
			hipLaunchKernelGGL(interval_count_kernel, dim3(count_grid), dim3(256), 0, s2, c);
			if (two_streams) {
				HIP_TRY(ctx, hipEventRecord(ctx->ev_counted, ctx->stream2));
				count_pending = true;
			}
		}
		if (want_rows && !fuse) {
			KernelTimer t(ctx, CONGA_K_REDUCE);
			hipLaunchKernelGGL(interval_map_rows_kernel, dim3(map_grid), dim3(256), 0, st, mr);
		}
		KernelTimer t(ctx, CONGA_K_INGEST);
		if (ctx->n_reads_total > 0) {
			TupleArgs a;
			a.pos = ptr<int32_t>(ctx->d_pos);
			a.mapq = ptr<uint8_t>(ctx->d_mapq);
			a.n_total = (uint32_t) ctx->n_reads_total;
			a.slots = dslots;
			a.n_slots = n_slots;
			a.gc_hist = ptr<uint8_t>(ctx->d_gc_hist);
			a.step = ctx->step;
			a.mq_threshold = ctx->opts.mq_threshold;
			a.small = small;
			a.n_chunks = ctx->tuple_chunks;
			a.chunks_per_block = ctx->tuple_chunks_per_block;
			a.block_home = ptr<TupleBlockHome>(ctx->d_block_home);
			const int grid = (int) ((a.n_chunks + a.chunks_per_block - 1) / a.chunks_per_block);
			if (fuse && (want_count || want_rows)) {
				const int cb = want_count ? count_grid : 0, mb = want_rows ? map_grid : 0;
				const bool all = a.mq_threshold < 0; // every read counts: the variant that never looks at the MAPQ bytes
				if (mb && all)
					hipLaunchKernelGGL((tuple_pass_kernel<true, true>), dim3(grid + cb + mb), dim3(kTupleBlock), 0, st, a, c, cb, mr, mb, grid);
				else if (mb)
					hipLaunchKernelGGL((tuple_pass_kernel<true, false>), dim3(grid + cb + mb), dim3(kTupleBlock), 0, st, a, c, cb, mr, mb, grid);
				else if (all)
					hipLaunchKernelGGL((tuple_pass_kernel<false, true>), dim3(grid + cb), dim3(kTupleBlock), 0, st, a, c, cb, mr, 0, grid);
				else
					hipLaunchKernelGGL((tuple_pass_kernel<false, false>), dim3(grid + cb), dim3(kTupleBlock), 0, st, a, c, cb, mr, 0, grid);
			} else if (a.mq_threshold < 0)
				hipLaunchKernelGGL(ingest_tuples_kernel<true>, dim3(grid), dim3(kTupleBlock), 0, st, a);
			else
				hipLaunchKernelGGL(ingest_tuples_kernel<false>, dim3(grid), dim3(kTupleBlock), 0, st, a);
		}
	} else if (!unsorted_mode) {
		TRY(launch_dense_depth(ctx, small, true));
	} else {
		KernelTimer t(ctx, CONGA_K_DEPTH);
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_rd.p, 0, (size_t) ctx->total_L * 2, st));
		for (int s = 0; s < n_slots; s++) {
			const HostSlot &h = ctx->slots[s];
			if (h.n_reads > 0) {
				const int grid = (int) std::min<int64_t>((h.n_reads + 255) / 256, (int64_t) ctx->n_cu * 8);
				hipLaunchKernelGGL(depth_atomic_kernel, dim3(grid), dim3(256), 0, st,
						ptr<int32_t>(ctx->d_pos) + h.read_off, ptr<uint8_t>(ctx->d_mapq) + h.read_off, h.n_reads, h.L,
						ctx->opts.mq_threshold, ptr<int16_t>(ctx->d_rd) + h.rd_off, small[s].counters);
			}
			const int64_t n_w = (h.L + ctx->step - 1) / ctx->step;
			const int grid = (int) std::min<int64_t>((n_w + 255) / 256, (int64_t) ctx->n_cu * 8);
			hipLaunchKernelGGL(gc_hist_kernel, dim3(grid), dim3(256), 0, st, ptr<int16_t>(ctx->d_rd) + h.rd_off, h.L,
					ptr<uint8_t>(ctx->d_gc_hist) + h.gc_off, h.n_win, ctx->step, small[s].hist_sum, small[s].hist_bases);
		}
	}

	// expected_read_depth[101] per chromosome.  The chain kernel derives its tables from the two histograms itself,
	// so outside profiling this job rides along as a few extra workgroups of the chain launch.
	const bool fuse_tables = !profile && ctx->n_iv > 0;
	if (!fuse_tables) {
		KernelTimer t(ctx, CONGA_K_EXPECTED);
		hipLaunchKernelGGL(expected_table_kernel, dim3(n_slots), dim3(128), 0, st, small,
				ptr<unsigned long long>(ctx->d_bases), small_by_kernel ? ctx->h_small : (Small *) nullptr);
	}

	// the reference paints the track only when the chromosome has at least one kept SV
	// (likelihood.c:332-336 returns before :352-356)
	if (ctx->any_map_painted && ctx->n_iv > 0) {
		KernelTimer t(ctx, CONGA_K_PAINT);
		for (int s = 0; s < n_slots; s++) {
			const HostSlot &h = ctx->slots[s];
			if (!track_painted(ctx, h) || h.iv_start[0].size() + h.iv_start[1].size() == 0)
				continue;
			const int32_t *ms = ptr<int32_t>(ctx->d_map_start) + h.map_row_off;
			const int32_t *me = ptr<int32_t>(ctx->d_map_end) + h.map_row_off;
			const float *mv = ptr<float>(ctx->d_map_val) + h.map_row_off;
			float *map = ptr<float>(ctx->d_map) + h.rd_off;
			const int64_t m = (int64_t) h.map_start.size();
			if (h.map_sorted) {
				const int64_t n_pt = (h.L + kPaintTile - 1) / kPaintTile;
				const int grid = (int) ((n_pt + kPaintTilesPerBlock - 1) / kPaintTilesPerBlock);
				hipLaunchKernelGGL(paint_sorted_kernel, dim3(grid), dim3(256), 0, st, ms, me, mv, m,
						ptr<uint32_t>(ctx->d_row_tile) + h.row_tile_off, map, h.L);
			} else {
				HIP_TRY(ctx, hipMemsetAsync(ctx->d_winner.p, 0xFF, (size_t) h.L * 4, st));
				if (m > 0) {
					const int grid = (int) std::min<int64_t>((m + 3) / 4, (int64_t) ctx->n_cu * 8);
					hipLaunchKernelGGL(paint_winner_kernel, dim3(grid), dim3(256), 0, st, ms, me, m,
							ptr<int32_t>(ctx->d_winner), h.L);
				}
				const int grid = (int) std::min<int64_t>((h.L + 255) / 256, (int64_t) ctx->n_cu * 16);
				hipLaunchKernelGGL(paint_resolve_kernel, dim3(grid), dim3(256), 0, st, ptr<int32_t>(ctx->d_winner), mv,
						map, h.L);
			}
		}
	}

	if (ctx->n_iv > 0 && (ctx->support_given || ctx->any_ref)) {
		if (ctx->support_given)
			HIP_TRY(ctx, hipMemcpyAsync(ctx->d_support.p, ctx->d_support_base.p, (size_t) ctx->n_iv * 4, hipMemcpyDeviceToDevice, st));
		else
			HIP_TRY(ctx, hipMemsetAsync(ctx->d_support.p, 0, (size_t) ctx->n_iv * 4, st));
	}
	// split-read evidence: half-read mapping against the resident 10-mer indexes -> pairing -> support, every chromosome's
	// records in one launch (count_ReadPairs runs only for chromosomes with SVs, likelihood.c:332-348: the others have no
	// interval to add to)
	if (ctx->any_sr && ctx->sr_units > 0) {
		KernelTimer t(ctx, CONGA_K_SPLIT);
		SplitMapArgs g;
		memset(&g, 0, sizeof g);
		g.pos = ptr<int32_t>(ctx->d_sr_pos);
		g.mapq = ptr<uint8_t>(ctx->d_sr_mapq);
		g.flag = ptr<uint16_t>(ctx->d_sr_flag);
		g.l_qseq = ptr<int32_t>(ctx->d_sr_lq);
		g.data_off = ptr<uint64_t>(ctx->d_sr_off);
		g.data = ptr<uint8_t>(ctx->d_sr_data);
		g.rec_off = ptr<uint64_t>(ctx->d_sr_recoff);
		g.stream = ptr<uint8_t>(ctx->d_bz_out);
		g.refn = ptr<uint32_t>(ctx->d_refn);
		g.sat_start = ptr<int32_t>(ctx->d_sat_start);
		g.sat_end = ptr<int32_t>(ctx->d_sat_end);
		g.offset = ptr<uint32_t>(ctx->d_kmer_offset);
		g.positions = ptr<int32_t>(ctx->d_kmer_pos);
		g.iv_start = ptr<int32_t>(ctx->d_iv_start);
		g.iv_end = ptr<int32_t>(ctx->d_iv_end);
		g.support = ptr<int32_t>(ctx->d_support);
		g.slots = ptr<SplitSlot>(ctx->d_sr_slots);
		g.n_slots = ctx->n_sr_slots;
		g.n_units = ctx->sr_units;
		g.small = small;
		g.mq_threshold = ctx->opts.mq_threshold;
		g.min_read_length = ctx->opts.min_read_length;
		const int sgrid = (int) std::min<int64_t>((int64_t) ctx->sr_units, (int64_t) ctx->n_cu * ctx->split_blocks_per_cu);
		hipLaunchKernelGGL(split_map_kernel, dim3(sgrid), dim3(256), 0, st, g);
	}

	if (ctx->n_iv > 0) {
		// interval_reduce needs read_depth and / or the painted track; the float chain needs only the depth table:
		// they run side by side on two streams and meet again in front of interval_score.
		hipStream_t st_reduce = st;
		const bool reduce = ctx->n_items > 0 && (dense || ctx->any_map_painted);
		if (reduce) {
			TRY(fork_to_s2(ctx->ev_fork2));
			st_reduce = s2;
		}
		if (reduce) {
			KernelTimer t(ctx, CONGA_K_REDUCE);
			ReduceArgs a;
			a.rd = dense ? ptr<int16_t>(ctx->d_rd) : nullptr;
			a.map = ptr<float>(ctx->d_map);
			a.item_off = ptr<int64_t>(ctx->d_item_off);
			a.item_len = ptr<int32_t>(ctx->d_item_len);
			a.item_iv = ptr<int32_t>(ctx->d_item_iv);
			a.item_has_map = ptr<uint8_t>(ctx->d_item_has_map);
			a.n_items = ctx->n_items;
			a.observed = observed_of(ctx);
			a.map_part = ptr<double>(ctx->d_map_part);
			const int waves_per_block = 256 / kWave;
			const int grid = (int) ((ctx->n_items + waves_per_block - 1) / waves_per_block);
			hipLaunchKernelGGL(interval_reduce_kernel, dim3(grid), dim3(256), 0, st_reduce, a);
		}
		if (fused_score && count_pending)
			HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_counted, 0)); // observed[] must be final before the first chain ends
		ScoreArgs sa;
		sa.start = ptr<int32_t>(ctx->d_iv_start);
		sa.end = ptr<int32_t>(ctx->d_iv_end);
		sa.type = ptr<uint8_t>(ctx->d_iv_type);
		sa.n_iv = ctx->n_iv;
		sa.observed = observed_of(ctx);
		sa.expected = ptr<float>(ctx->d_expected);
		sa.map_part = ptr<double>(ctx->d_map_part);
		sa.item_first = ptr<int32_t>(ctx->d_item_first);
		sa.iv_has_map = ptr<uint8_t>(ctx->d_iv_has_map);
		sa.support = (ctx->support_given || ctx->any_ref) ? ptr<int32_t>(ctx->d_support) : nullptr;
		sa.out = ptr<conga_result>(ctx->d_results);
		{
			KernelTimer t(ctx, CONGA_K_CHAIN);
			ChainArgs c;
			c.fused_score = fused_score ? 1 : 0;
			c.score = sa;
			const bool to_host = (ctx->opts.flags & CONGA_FLAG_RESULTS_ON_DEVICE) == 0;
			c.out_host = to_host ? ctx->h_results : nullptr;
			ctx->host_results_by_order = fused_score && to_host;
			ctx->host_results_valid = to_host;
			c.start = ptr<int32_t>(ctx->d_iv_start);
			c.end = ptr<int32_t>(ctx->d_iv_end);
			c.iv_slot = ptr<int32_t>(ctx->d_iv_slot);
			c.order = ptr<int32_t>(ctx->d_order);
			c.gc_like = gc_like;
			c.slots = dslots;
			c.small = small;
			c.bases = ptr<unsigned long long>(ctx->d_bases);
			c.step = ctx->step;
			c.expected = ptr<float>(ctx->d_expected);
			c.n_x = ctx->n_chain_x;
			c.n_a = ctx->n_chain_a;
			c.n_b = ctx->n_chain_b;
			c.n_iv = ctx->n_iv;
			c.n_slots = n_slots;
			// classes A and B share workgroups: one class-A wave in each of the first n_a, class-B waves (four 16-lane
			// groups each) in every other slot
			const int64_t waves_b = (c.n_b + 3) / 4;
			c.blocks_ab = (int32_t) (waves_b <= 3 * c.n_a ? c.n_a : c.n_a + (waves_b - 3 * c.n_a + 3) / 4);
			const int blocks_c = (int) ((c.n_iv - c.n_x - c.n_a - c.n_b + 255) / 256); // one lane per interval
			c.table_blocks = fuse_tables ? n_slots : 0;
			c.host_small = small_by_kernel ? ctx->h_small : nullptr;
			// clear the other arena on the side (not when the step is being captured into a graph: pointers are baked in)
			const int other = ctx->small_cur ^ 1;
			const bool zero_other = fuse_tables && !getenv("CONGA_GRAPH");
			c.zero_blocks = zero_other ? 8 : 0;
			c.zero_ptr = reinterpret_cast<uint4 *>(arena_of(ctx, other));
			c.zero_n16 = (int64_t) (ctx->arena_bytes / 16);
			next_arena = zero_other ? other : ctx->small_cur;
			if (zero_other)
				ctx->arena_zeroed[other] = true;
			hipLaunchKernelGGL(interval_chain_kernel, dim3((int) c.n_x + c.blocks_ab + blocks_c + c.zero_blocks + c.table_blocks), dim3(256), 0,
					st, c);
		}
		if (!fused_score) {
			TRY(join_s2());
			{
				KernelTimer t(ctx, CONGA_K_SCORE);
				const int grid = (int) ((ctx->n_iv + 63) / 64);
				hipLaunchKernelGGL(interval_score_kernel, dim3(grid), dim3(64), 0, st, sa);
			}
			if ((ctx->opts.flags & CONGA_FLAG_RESULTS_ON_DEVICE) == 0)
				HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results.p, (size_t) ctx->n_iv * sizeof(conga_result),
						hipMemcpyDeviceToHost, st));
		}
	}
	TRY(join_s2());
	ctx->small_cur_next = next_arena;
	if (!small_by_kernel)
	HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small, small, (size_t) n_slots * sizeof(Small), hipMemcpyDeviceToHost, st));
	return CONGA_OK;
}

// The tuple pass looks for runs of equal positions long enough to wrap the reference's `short` depth counter
// (kStatusWrapRisk, kernels.hip.h).  Reads that came through conga_sample_reads() have no other guard, and the finding
// arrives with the results: the records just computed then count reads where the reference counts modulo 2^16, so the
// step is computed once more in the dense formulation, which reproduces the wrap.  Called at every point where the host
// waits for a compute (fetch, conga_sync).
int settle_wrap_risk(conga_ctx *ctx)
{
	if (!ctx->computed || ctx->depth_resident || ctx->wrap_risk || !ctx->h_small)
		return CONGA_OK;
	bool risk = false;
	for (size_t s = 0; s < ctx->slots.size(); s++)
		risk = risk || (ctx->h_small[s].status & kStatusWrapRisk) != 0;
	if (!risk)
		return CONGA_OK;
	if (!ctx->reads_ahead) {
		ctx->wrap_risk = true;
		TRY(conga_chrom_compute(ctx));
		HIP_TRY(ctx, hipEventSynchronize(ctx->ev_done));
		return CONGA_OK;
	}
	// The next sample's tuples are already on their way (conga_sample_reads beside this compute): the sample that has to be
	// computed again lies in the other pair of buffers, described by computed_reads.  Swap it in, compute, swap back; the
	// copy under way writes the pair that is not touched here.
	std::vector<std::pair<int64_t, int64_t>> next_reads(ctx->slots.size());
	for (size_t c = 0; c < ctx->slots.size(); c++) {
		next_reads[c] = std::make_pair(ctx->slots[c].read_off, ctx->slots[c].n_reads);
		ctx->slots[c].read_off = ctx->computed_reads[c].first;
		ctx->slots[c].n_reads = ctx->computed_reads[c].second;
	}
	const int64_t next_total = ctx->n_reads_total;
	const bool pending_copy = ctx->reads_on_stream2, pending_expand = ctx->expand_pending;
	ctx->expand_pending = false; // (the NEXT sample's differences: not this compute's to expand)
	ctx->n_reads_total = ctx->computed_total;
	std::swap(ctx->d_pos, ctx->d_pos_alt);
	std::swap(ctx->d_mapq, ctx->d_mapq_alt);
	ctx->pos_buf ^= 1;
	ctx->reads_on_stream2 = false; // (this compute reads the OLD pair: nothing to wait for)
	ctx->wrap_risk = true;
	ctx->sample_dirty = true;
	int rc = conga_chrom_compute(ctx);
	if (rc == CONGA_OK && hipEventSynchronize(ctx->ev_done) != hipSuccess)
		rc = fail(ctx, CONGA_ERR_HIP, "settle_wrap_risk: waiting for the second compute failed");
	for (size_t c = 0; c < ctx->slots.size(); c++) {
		ctx->slots[c].read_off = next_reads[c].first;
		ctx->slots[c].n_reads = next_reads[c].second;
	}
	ctx->n_reads_total = next_total;
	std::swap(ctx->d_pos, ctx->d_pos_alt);
	std::swap(ctx->d_mapq, ctx->d_mapq_alt);
	ctx->pos_buf ^= 1;
	ctx->reads_on_stream2 = pending_copy;
	ctx->expand_pending = pending_expand;
	ctx->wrap_risk = false; // (the next sample's own guard runs with its compute)
	ctx->sample_dirty = true;
	ctx->reads_ahead = true;
	return rc;
}

} // namespace

extern "C" {

int conga_chrom_fetch(conga_ctx *ctx, conga_result *dels, conga_result *dups, float expected_rd[101],
		conga_chrom_stats *stats)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (!ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_fetch: nothing computed");
	HostSlot *h = current(ctx);
	if (!h)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_fetch: no chromosome selected");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipEventSynchronize(ctx->ev_done));
	TRY(settle_wrap_risk(ctx));
	const Small &sb = ctx->h_small[ctx->cur];
	if (sb.status & kStatusUnsorted)
		return fail(ctx, CONGA_ERR_UNSORTED,
				"reads were committed out of position order; pass CONGA_FLAG_READS_UNSORTED to accept that");
	const size_t nd = h->iv_start[0].size(), nu = h->iv_start[1].size();
	if ((nd && !dels) || (nu && !dups))
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_fetch: result array missing");
	if (!ctx->host_results_valid && ctx->n_iv > 0) { // CONGA_FLAG_RESULTS_ON_DEVICE: bring the records over now
		HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results.p, (size_t) ctx->n_iv * sizeof(conga_result),
				hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		ctx->host_results_by_order = false;
		ctx->host_results_valid = true;
	}
	if (!ctx->host_results_by_order) {
		if (nd)
			memcpy(dels, ctx->h_results + h->iv0, nd * sizeof(conga_result));
		if (nu)
			memcpy(dups, ctx->h_results + h->iv0 + nd, nu * sizeof(conga_result));
	} else { // the chain kernel wrote the host copy in its own processing order
		for (size_t i = 0; i < nd; i++)
			dels[i] = ctx->h_results[ctx->order_pos[(size_t) h->iv0 + i]];
		for (size_t i = 0; i < nu; i++)
			dups[i] = ctx->h_results[ctx->order_pos[(size_t) h->iv0 + nd + i]];
	}
	if (expected_rd)
		memcpy(expected_rd, sb.E, kGcBins * sizeof(float));
	if (stats) {
		memset(stats, 0, sizeof *stats);
		stats->reads_committed = (size_t) ctx->cur < ctx->computed_reads.size() ? ctx->computed_reads[(size_t) ctx->cur].second : h->n_reads;
		stats->reads_counted = (int64_t) sb.counters[CNT_COUNTED];
		stats->reads_out_of_range = (int64_t) sb.counters[CNT_OUT_OF_RANGE];
		long long total = 0;
		for (int g = 0; g < kGcBins; g++) {
			stats->rd_per_gc[g] = (int64_t) sb.hist_sum[g];
			stats->window_per_gc[g] = (int64_t) sb.hist_bases[g];
			total += (long long) sb.hist_sum[g];
		}
		stats->rd_sum = total;
		stats->mean = (float) ((double) total / (double) h->L); // read_distribution.c:39
		stats->n_kernels = CONGA_K_COUNT;
		stats->split_elements = (int64_t) sb.counters[CNT_SR_ELEMENTS];
		stats->split_mappings = (int64_t) sb.counters[CNT_SR_MAPPINGS];
		stats->split_del_rows = (int64_t) sb.counters[CNT_SR_DEL_ROWS];
		stats->split_dup_rows = (int64_t) sb.counters[CNT_SR_DUP_ROWS];
		stats->depth_materialized = ctx->depth_resident ? 1 : 0;
		if (ctx->opts.flags & CONGA_FLAG_PROFILE) {
			for (int k = 0; k < CONGA_K_COUNT; k++) {
				float ms = 0.0f;
				if (ctx->ev_used[k] && hipEventElapsedTime(&ms, ctx->ev_k0[k], ctx->ev_k1[k]) == hipSuccess)
					stats->kernel_ms[k] = ms;
			}
		}
	}
	return CONGA_OK;
}

int conga_chrom_finish(conga_ctx *ctx, conga_result *dels, conga_result *dups, float expected_rd[101],
		conga_chrom_stats *stats)
{
	int rc = conga_chrom_compute(ctx);
	if (rc != CONGA_OK)
		return rc;
	return conga_chrom_fetch(ctx, dels, dups, expected_rd, stats);
}

int conga_sample_fetch(conga_ctx *ctx, conga_result *records, size_t n_records, float *expected_rd, conga_chrom_stats *stats)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (!ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_sample_fetch: nothing computed");
	if (n_records != (size_t) ctx->n_iv || (n_records && !records))
		return fail(ctx, CONGA_ERR_INVALID, "conga_sample_fetch: n_records differs from the intervals the context holds");
	const int keep = ctx->cur;
	// per chromosome through the one-chromosome fetch: same checks, same un-permutation, records laid out one
	// chromosome behind the other (deletions, then duplications) -- the order of conga_results_device()
	for (int c = 0; c < (int) ctx->slots.size(); c++) {
		const HostSlot &h = ctx->slots[(size_t) c];
		ctx->cur = c;
		const size_t nd = h.iv_start[0].size();
		const int rc = conga_chrom_fetch(ctx, records + h.iv0, records + h.iv0 + nd, expected_rd ? expected_rd + (size_t) c * kGcBins : nullptr,
				stats ? stats + c : nullptr);
		if (rc != CONGA_OK) {
			ctx->cur = keep;
			return rc;
		}
	}
	ctx->cur = keep;
	return CONGA_OK;
}

int conga_results_device(conga_ctx *ctx, void **dev_ptr, size_t *n_records)
{
	if (!ctx || !dev_ptr)
		return CONGA_ERR_INVALID;
	if (!ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_results_device: nothing computed");
	*dev_ptr = ctx->n_iv ? ctx->d_results.p : nullptr;
	if (n_records)
		*n_records = (size_t) ctx->n_iv;
	return CONGA_OK;
}

int conga_results_copy(conga_ctx *ctx, void *dst_device, size_t dst_bytes)
{
	if (!ctx || (!dst_device && ctx->n_iv))
		return CONGA_ERR_INVALID;
	if (!ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_results_copy: nothing computed");
	const size_t bytes = (size_t) ctx->n_iv * sizeof(conga_result);
	if (dst_bytes < bytes)
		return fail(ctx, CONGA_ERR_INVALID, "conga_results_copy: destination too small");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (bytes)
		HIP_TRY(ctx, hipMemcpyAsync(dst_device, ctx->d_results.p, bytes, hipMemcpyDeviceToDevice, ctx->stream));
	return CONGA_OK;
}

int conga_set_profile(conga_ctx *ctx, int on)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (on)
		ctx->opts.flags |= CONGA_FLAG_PROFILE;
	else
		ctx->opts.flags &= ~CONGA_FLAG_PROFILE;
	return CONGA_OK;
}

void *conga_stream(conga_ctx *ctx)
{
	return ctx ? (void *) ctx->stream : nullptr;
}

int conga_sync(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->reads_on_stream2)
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream2)); // (the copies of a conga_sample_reads that no compute has taken up yet)
	TRY(settle_wrap_risk(ctx));
	return CONGA_OK;
}

int conga_copy_read_depth(conga_ctx *ctx, int16_t *out, int64_t n)
{
	HostSlot *h = ctx ? current(ctx) : nullptr;
	if (!ctx || !out || !ctx->computed || !h || n > h->L || n < 0)
		return CONGA_ERR_INVALID;
	if (ctx->reads_ahead)
		return fail(ctx, CONGA_ERR_INVALID, "conga_copy_read_depth: the reads have been replaced since the compute");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (!ctx->depth_resident) {
		// tuple-space compute: build read_depth[] now; its by-products go to a scratch block, not into the results
		const size_t bytes = ctx->slots.size() * sizeof(Small);
		TRY(ensure(ctx, ctx->d_small_scratch, bytes));
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_small_scratch.p, 0, bytes, ctx->stream));
		TRY(launch_dense_depth(ctx, ptr<Small>(ctx->d_small_scratch), false));
		HIP_TRY(ctx, hipGetLastError());
		ctx->depth_resident = true;
	}
	HIP_TRY(ctx, hipMemcpyAsync(out, ptr<int16_t>(ctx->d_rd) + h->rd_off, (size_t) n * 2, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CONGA_OK;
}

int conga_copy_mappability(conga_ctx *ctx, float *out, int64_t n)
{
	HostSlot *h = ctx ? current(ctx) : nullptr;
	if (!ctx || !out || !ctx->computed || !h || !h->has_map || h->iv_start[0].size() + h->iv_start[1].size() == 0
			|| n > h->L || n < 0)
		return CONGA_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (!track_painted(ctx, *h)) {
		// row-space compute: paint this chromosome's (sorted) track now
		TRY(ensure(ctx, ctx->d_map, std::max<size_t>((size_t) ctx->total_L, 8) * 4));
		const int64_t n_pt = (h->L + kPaintTile - 1) / kPaintTile;
		const int grid = (int) ((n_pt + kPaintTilesPerBlock - 1) / kPaintTilesPerBlock);
		hipLaunchKernelGGL(paint_sorted_kernel, dim3(grid), dim3(256), 0, ctx->stream,
				ptr<int32_t>(ctx->d_map_start) + h->map_row_off, ptr<int32_t>(ctx->d_map_end) + h->map_row_off,
				ptr<float>(ctx->d_map_val) + h->map_row_off, (int64_t) h->map_start.size(),
				ptr<uint32_t>(ctx->d_row_tile) + h->row_tile_off, ptr<float>(ctx->d_map) + h->rd_off, h->L);
		HIP_TRY(ctx, hipGetLastError());
	}
	HIP_TRY(ctx, hipMemcpyAsync(out, ptr<float>(ctx->d_map) + h->rd_off, (size_t) n * 4, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CONGA_OK;
}

} // extern "C"
