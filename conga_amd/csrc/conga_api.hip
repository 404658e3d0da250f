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
//
// One translation unit, in parts (round 4; it was one file of 4 195 lines):
//   engine_ctx.hip.h      device buffers, the host side of a chromosome, conga_ctx, small helpers
//   engine_knobs.h        every switch taken from the environment, read once per context (no other getenv in the library)
//   engine_layout.hip.h   prepare_sample / prepare_layout: the tables the kernels work from
//   bz_sched.h            the upload pipeline's scheduler -- host code, tested without a GPU under the thread sanitizer
//   engine_bgzf.hip.h     conga_reads_bgzf*: inflate launches, pinned ring, the scheduler's HIP machine, record walks
//   pack_host.h           conga_packer_*: the host producer of the packed hand-over
//   engine_compute.hip.h  conga_chrom_compute / fetch: the launches of a step, the records back
//   this file             lifetime, chromosomes, read staging, cohort hand-over, tracks, intervals, split-read staging
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
#include "bz_sched.h"
#include "engine_knobs.h"

using namespace conga;

#include "engine_ctx.hip.h"
#include "engine_layout.hip.h"
#include "engine_bgzf.hip.h"

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
	hand_spare_on(ctx);
	ctx->sched.quiesce(false); // (an upload named ahead and never asked for is given up: its pieces go through the ring)
	if (!ctx->h_bz_ring)
		return CONGA_OK;
	if (hipSetDevice(ctx->device) != hipSuccess)
		return CONGA_ERR_HIP;
	if (ctx->bz_copy2)
		(void) hipStreamSynchronize(ctx->bz_copy2);
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
	const conga::Knobs knobs = conga::read_knobs(); // (the one place the library looks at the environment: engine_knobs.h)
	const bool say = knobs.timing;
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
	ctx->knobs = knobs;
	ctx->machine.reset(new HipMachine(ctx));
	ctx->sched.m = ctx->machine.get();
	ctx->sched.cfg.slot_bytes = knobs.bgzf_slot_bytes;
	ctx->sched.cfg.n_slots = knobs.bgzf_slots;
	ctx->sched.cfg.piece = knobs.bgzf_piece_kb > 0 ? (size_t) knobs.bgzf_piece_kb << 10 : 0;
	ctx->sched.cfg.pieces_per_launch_small = kBzPiecesPerLaunch;
	ctx->sched.cfg.copy_threads = knobs.bgzf_copy_threads;
	ctx->sched.cfg.cpus = cpus_allowed();
	ctx->sched.cfg.plain_pread = knobs.bgzf_plain_pread;
	ctx->sched.cfg.no_inflate_ahead = knobs.bgzf_no_inflate_ahead || knobs.bgzf_other_kernel;
	ctx->sched.cfg.no_table = knobs.bgzf_no_table;
	ctx->sched.cfg.no_ahead = knobs.bgzf_no_ahead;
	if (knobs.bgzf_trace)
		bz::trace_on().store(true);
	if (knobs.bgzf_ahead_follow)
		ctx->sched.cfg.ahead_wait_factor = 0;
	ctx->sched.cfg.timing = knobs.timing;
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
	ctx->bz_shared = !ctx->knobs.streams_normal;
	int prio_low = 0, prio_high = 0;
	(void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
	if (!ctx->bz_shared)
		prio_low = 0;
	if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_low) != hipSuccess)
		return bail(CONGA_ERR_HIP);
	if (hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_set, hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->ev_set_prev, hipEventDisableTiming) != hipSuccess
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
			&ctx->d_bz_crc, &ctx->d_bz_x2n, &ctx->d_bz_ticket, &ctx->d_bz_seg, &ctx->d_bz_cnt, &ctx->d_bz_first, &ctx->d_bz_stop, &ctx->d_bz_bad, &ctx->d_bz_at, &ctx->d_bz_flag, &ctx->d_expected, &ctx->d_item_off, &ctx->d_item_len, &ctx->d_item_iv,
			&ctx->d_item_has_map, &ctx->d_item_first, &ctx->d_map_part, &ctx->d_support, &ctx->d_results, &ctx->d_results_prev,
			&ctx->d_bases, &ctx->d_row_tile, &ctx->d_depth_blocks, &ctx->d_support_base, &ctx->d_ref, &ctx->d_sat_start, &ctx->d_sat_end,
			&ctx->d_sr_pos, &ctx->d_sr_mapq, &ctx->d_sr_flag, &ctx->d_sr_lq, &ctx->d_sr_off, &ctx->d_sr_data,
			&ctx->d_sr_recoff, &ctx->d_refn, &ctx->d_kmer_keys, &ctx->d_kmer_sorted, &ctx->d_kmer_tmp, &ctx->d_kmer_offset, &ctx->d_kmer_pos, &ctx->d_kmer_pres};
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
	if (ctx->h_small_prev)
		(void) hipHostFree(ctx->h_small_prev);
	if (ctx->h_walk)
		(void) hipHostFree(ctx->h_walk);
	if (ctx->h_results_prev)
		(void) hipHostFree(ctx->h_results_prev);
	if (ctx->h_head)
		(void) hipHostFree(ctx->h_head);
	hand_spare_on(ctx);
	ctx->sched.quiesce(true);
	ctx->sched.job_kept.reset();
	for (auto &o : ctx->sched.buf_owner)
		o.reset(); // (the jobs' events go while the device is still there)
	ctx->sched.spare_owner.reset();
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
	if (ctx->bz_copy2)
		(void) hipStreamDestroy(ctx->bz_copy2);
	if (ctx->ev_bz_copy2)
		(void) hipEventDestroy(ctx->ev_bz_copy2);
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
	if (ctx->ev_set)
		(void) hipEventDestroy(ctx->ev_set);
	if (ctx->ev_set_prev)
		(void) hipEventDestroy(ctx->ev_set_prev);
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
	// (two computes in flight, conga_chrom_compute_ahead: the older one's tuples are about to be given up -- its guard first)
	TRY(settle_previous(ctx));
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
	bz::trace("conga_sample_begin");
	TRY(drop_reads(ctx, "conga_sample_begin"));
	bz::trace("conga_sample_begin: the reads are dropped");
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

int conga_packer_start_v(conga_packer *p, const int32_t *const *chrom_pos, const uint64_t *chrom_off, int n_chrom, int width, uint8_t *out,
		size_t out_cap)
{
	if (!p)
		return CONGA_ERR_INVALID;
	const int rc = p->impl.start_v(chrom_pos, chrom_off, n_chrom, width, out, out_cap);
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

extern "C" {

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

} // extern "C"

#include "engine_compute.hip.h"
