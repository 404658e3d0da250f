// engine_compute.hip.h -- calc_mean_per_chr + find_depths on the device (read_distribution.c:49-84, likelihood.c:96-169,290-308):
// conga_chrom_compute enqueues every launch of a step, conga_chrom_fetch / conga_sample_fetch bring the records back, the wrap
// guard's second compute, and the device-side access used by the multi-GPU gather.  Part of conga_api.hip's one translation unit.
#pragma once

namespace {

int settle_previous(conga_ctx *ctx);

// the set of the latest compute <-> the set of the one before (engine_ctx.hip.h)
void swap_result_sets(conga_ctx *ctx)
{
	std::swap(ctx->ev_set, ctx->ev_set_prev);
	std::swap(ctx->h_small, ctx->h_small_prev);
	std::swap(ctx->h_small_cap, ctx->h_small_prev_cap);
	std::swap(ctx->h_results, ctx->h_results_prev);
	std::swap(ctx->h_results_cap, ctx->h_results_prev_cap);
	std::swap(ctx->d_results, ctx->d_results_prev);
	std::swap(ctx->host_results_by_order, ctx->prev_by_order);
	std::swap(ctx->host_results_valid, ctx->prev_valid);
	std::swap(ctx->depth_resident, ctx->prev_depth_resident);
	ctx->computed_reads.swap(ctx->prev_computed_reads);
	std::swap(ctx->computed_total, ctx->prev_computed_total);
}

// every launch of a step into the queues (conga_chrom_compute, conga_chrom_compute_ahead, the wrap guard's second compute)
int compute_step(conga_ctx *ctx)
{
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
	if (whole_layout && ctx->knobs.timing)
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
	const bool use_graph = (ctx->opts.flags & CONGA_FLAG_PROFILE) == 0 && ctx->knobs.graph
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
	ctx->rd_clobbered = false;
	ctx->small_cur = ctx->small_cur_next; // the arena the chain launch has just cleared, if it did
	HIP_TRY(ctx, hipEventRecord(ctx->ev_done, st));
	HIP_TRY(ctx, hipEventRecord(ctx->ev_set, st));
	ctx->computed_once.store(true, std::memory_order_release);
	HIP_TRY(ctx, hipEventRecord(ctx->ev_pair[ctx->pos_buf], st));
	ctx->used_recorded[ctx->pos_buf] = true;
	ctx->reads_ahead = false;
	ctx->computed_reads.resize(ctx->slots.size());
	for (size_t c = 0; c < ctx->slots.size(); c++)
		ctx->computed_reads[c] = std::make_pair(ctx->slots[c].read_off, ctx->slots[c].n_reads);
	ctx->computed_total = ctx->n_reads_total;
	HIP_TRY(ctx, hipGetLastError());
	ctx->computed = true;
	bz::trace("compute: enqueued");
	hand_spare_on(ctx); // (this sample's launches are in the queues: the next sample's inflates may follow them -- engine_bgzf.hip.h)
	return CONGA_OK;
}

} // namespace

extern "C" {

int conga_chrom_compute(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	ctx->have_previous = false; // (what the compute before the last one left is out of reach from here on)
	return compute_step(ctx);
}

// The step's launches go into the queues BEHIND the last compute's, whose results stay where they are until
// conga_sample_fetch_previous / conga_sync_previous / conga_results_copy_previous have had them: hand over sample k + 1, compute it
// ahead, fetch sample k -- the GPU goes from one sample's last launch to the next one's first while the host is still waking up.
// Without a compute whose results could be kept (none yet, or the layout has changed since) this is conga_chrom_compute.
int conga_chrom_compute_ahead(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	// (a layout that only has to be laid out for the other formulation -- the sample before needed the dense kernels -- keeps its
	// chromosomes, intervals and their order: the older records stay what they were)
	if (!ctx->computed || ctx->layout_dirty || (ctx->opts.flags & CONGA_FLAG_PROFILE) != 0 || ctx->n_sr_total != 0 || ctx->bz_keep_bytes != 0)
		return conga_chrom_compute(ctx);
	if (ctx->staging_cur >= 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_compute_ahead: a staging buffer is handed out and not committed");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	// (the set before the last one -- if there is one -- is given up: its guard is looked at first, an error of its own is the caller's
	// who never fetched it)
	TRY(settle_previous(ctx));
	// the spare set as large as the one in use (first time: two pinned blocks and the records' buffer; nothing is in flight on them)
	if (ctx->h_small_prev_cap < ctx->h_small_cap) {
		if (ctx->h_small_prev)
			(void) hipHostFree(ctx->h_small_prev);
		ctx->h_small_prev = nullptr;
		ctx->h_small_prev_cap = 0;
		HIP_TRY(ctx, hipHostMalloc((void **) &ctx->h_small_prev, ctx->h_small_cap * sizeof(Small), hipHostMallocDefault));
		ctx->h_small_prev_cap = ctx->h_small_cap;
	}
	if (ctx->h_results_prev_cap < ctx->h_results_cap) {
		if (ctx->h_results_prev)
			(void) hipHostFree(ctx->h_results_prev);
		ctx->h_results_prev = nullptr;
		ctx->h_results_prev_cap = 0;
		HIP_TRY(ctx, hipHostMalloc((void **) &ctx->h_results_prev, ctx->h_results_cap * sizeof(conga_result), hipHostMallocDefault));
		ctx->h_results_prev_cap = ctx->h_results_cap;
	}
	TRY(ensure(ctx, ctx->d_results_prev, ctx->d_results.cap));
	drop_graph(ctx); // (a captured step has the other set's addresses in it)
	swap_result_sets(ctx);
	const int rc = compute_step(ctx);
	if (rc != CONGA_OK) {
		swap_result_sets(ctx);
		ctx->have_previous = false;
		return rc;
	}
	ctx->have_previous = true;
	ctx->previous_settled = false;
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
		g.pres = ptr<uint2>(ctx->d_kmer_pres);
		g.flags = (uint32_t) ctx->knobs.split_flags;
		// The solo / echo bits (split_map.hip.h) cost a genome's index ~0.15 s more and save a sample's launch a quarter of its time:
		// a single sample is better off without them, a cohort's second sample pays for them and every later one gains.  Made
		// here, once per index, in front of the second launch that uses it (everything they are made from is resident).
		// (a context that was told a cohort is coming -- CONGA_FLAG_EXPECT_COHORT -- makes them with the first launch)
		if (!ctx->pres_built && (ctx->sr_launches_on_index >= 1 || (ctx->opts.flags & CONGA_FLAG_EXPECT_COHORT) != 0) && (g.flags & 1u) == 0) {
			for (int s = 0; s < n_slots; s++) {
				const HostSlot &h = ctx->slots[(size_t) s];
				if (h.kidx < 0)
					continue;
				const uint32_t *off_c = ptr<uint32_t>(ctx->d_kmer_offset) + (size_t) h.kidx * ((size_t) kKmerBuckets + 2);
				const int32_t *pos_c = ptr<int32_t>(ctx->d_kmer_pos) + h.kpos_off;
				uint2 *bits_c = ptr<uint2>(ctx->d_kmer_pres) + h.pres_off;
				hipLaunchKernelGGL(kmer_solo_kernel, dim3(ctx->n_cu * 8), dim3(256), 0, st, pos_c, off_c, bits_c);
				const int ge = (int) std::min<int64_t>((h.L + 255) / 256, (int64_t) ctx->n_cu * 16);
				hipLaunchKernelGGL(kmer_echo_kernel, dim3(ge), dim3(256), 0, st, ptr<uint32_t>(ctx->d_refn) + h.refn_off, h.L, off_c, pos_c, bits_c);
			}
			ctx->pres_built = true;
		}
		if (!ctx->pres_built)
			g.flags |= 1u; // (no bits yet: every seed asks its bucket, as in round 3)
		ctx->sr_launches_on_index++;
		g.iv_start = ptr<int32_t>(ctx->d_iv_start);
		g.iv_end = ptr<int32_t>(ctx->d_iv_end);
		g.support = ptr<int32_t>(ctx->d_support);
		g.slots = ptr<SplitSlot>(ctx->d_sr_slots);
		g.n_slots = ctx->n_sr_slots;
		g.n_units = ctx->sr_units;
		g.small = small;
		g.mq_threshold = ctx->opts.mq_threshold;
		g.min_read_length = ctx->opts.min_read_length;
		// (a multiple of eight workgroups: workgroup w runs on XCD w % 8, and the kernel gives every XCD a run of consecutive units)
		const int sgrid = (int) std::max<int64_t>(8, std::min<int64_t>(((int64_t) ctx->sr_units + 7) & ~(int64_t) 7, (int64_t) ctx->n_cu * ctx->split_blocks_per_cu) & ~(int64_t) 7);
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
			ctx->host_results_by_order = false; // (the chain kernel writes the host copy in interval order too: the fetch is a plain copy)
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
			const bool zero_other = fuse_tables && !ctx->knobs.graph;
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
		TRY(compute_step(ctx));
		HIP_TRY(ctx, hipEventSynchronize(ctx->ev_set));
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
	int rc = compute_step(ctx);
	if (rc == CONGA_OK && hipEventSynchronize(ctx->ev_set) != hipSuccess)
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

// The same for the compute BEFORE the latest one (conga_chrom_compute_ahead): its tuples lie in the other pair of buffers as long
// as no further sample has been handed over -- every path that hands one over comes through here first.  With the sets swapped the
// case is settle_wrap_risk's own "the next sample is already there": the HostSlots describe a newer sample (here: one that has even
// been computed) than the set being looked at.
int settle_previous(conga_ctx *ctx)
{
	if (!ctx->have_previous || ctx->previous_settled)
		return CONGA_OK;
	if (!ctx->computed) { // (the reads were dropped or the layout touched since: the older results went with the latest ones)
		ctx->have_previous = false;
		return CONGA_OK;
	}
	ctx->previous_settled = true;
	swap_result_sets(ctx);
	int rc = hipEventSynchronize(ctx->ev_set) == hipSuccess ? CONGA_OK : fail(ctx, CONGA_ERR_HIP, "settle_previous: waiting for the compute failed");
	if (rc == CONGA_OK) {
		// (wrap_risk is what the host knows about the LATEST sample -- set, for one, when that sample's own guard has just sent it to
		// the dense kernels; it says nothing about the older one)
		const bool keep = ctx->reads_ahead, keep_risk = ctx->wrap_risk, was_dense = ctx->depth_resident;
		ctx->reads_ahead = true;
		ctx->wrap_risk = false;
		rc = settle_wrap_risk(ctx);
		ctx->reads_ahead = keep;
		ctx->wrap_risk = keep_risk;
		if (!was_dense && ctx->depth_resident && ctx->prev_depth_resident)
			ctx->rd_clobbered = true; // read_depth[] in HBM is the older sample's now: conga_copy_read_depth builds the latest one's again
	}
	swap_result_sets(ctx);
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
	const auto tf0 = std::chrono::steady_clock::now();
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipEventSynchronize(ctx->ev_set));
	const auto tf1 = std::chrono::steady_clock::now();
	TRY(settle_wrap_risk(ctx));
	const auto tf2 = std::chrono::steady_clock::now();
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
		const conga_result *src = ctx->h_results;
		for (size_t i = 0; i < nd; i++)
			dels[i] = src[ctx->order_pos[(size_t) h->iv0 + i]];
		for (size_t i = 0; i < nu; i++)
			dups[i] = src[ctx->order_pos[(size_t) h->iv0 + nd + i]];
	}
	const auto tf3 = std::chrono::steady_clock::now();
	if (expected_rd)
		memcpy(expected_rd, sb.E, kGcBins * sizeof(float));
	if (ctx->knobs.timing) { // (measurement switch CONGA_DEBUG=1 CONGA_TIMING=1: where a fetch's time goes, every 64th call)
		static thread_local double acc[4] = {0, 0, 0, 0};
		static thread_local int calls = 0;
		const auto tf4 = std::chrono::steady_clock::now();
		acc[0] += std::chrono::duration<double, std::micro>(tf1 - tf0).count();
		acc[1] += std::chrono::duration<double, std::micro>(tf2 - tf1).count();
		acc[2] += std::chrono::duration<double, std::micro>(tf3 - tf2).count();
		acc[3] += std::chrono::duration<double, std::micro>(tf4 - tf3).count();
		if (++calls % 64 == 0) {
			fprintf(stderr, "[timing] conga_chrom_fetch x64: the wait %.1f us, the guard %.1f, the records %.1f, the table %.1f (per call)\n", acc[0] / 64, acc[1] / 64,
					acc[2] / 64, acc[3] / 64);
			acc[0] = acc[1] = acc[2] = acc[3] = 0;
		}
	}
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

// ---- the results of the compute BEFORE the latest one (conga_chrom_compute_ahead).  Each call changes the two sets over, does
// what its namesake does, and changes them back: nothing else of the context notices.
namespace {
int previous_or_fail(conga_ctx *ctx, const char *who)
{
	if (!ctx->have_previous || !ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, std::string(who) + ": no compute before the latest one whose results are kept (conga_chrom_compute_ahead)");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	return settle_previous(ctx);
}
} // namespace

int conga_sample_fetch_previous(conga_ctx *ctx, conga_result *records, size_t n_records, float *expected_rd, conga_chrom_stats *stats)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	TRY(previous_or_fail(ctx, "conga_sample_fetch_previous"));
	swap_result_sets(ctx);
	const bool keep = ctx->reads_ahead;
	ctx->reads_ahead = true; // (the HostSlots describe a newer sample; the guard has been settled: nothing is computed again)
	const int rc = conga_sample_fetch(ctx, records, n_records, expected_rd, stats);
	ctx->reads_ahead = keep;
	swap_result_sets(ctx);
	return rc;
}

int conga_sync_previous(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	return previous_or_fail(ctx, "conga_sync_previous"); // (settle_previous waits for that compute's last launch)
}

int conga_results_copy_previous(conga_ctx *ctx, void *dst_device, size_t dst_bytes)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (!ctx->have_previous || !ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_results_copy_previous: no compute before the latest one whose results are kept");
	swap_result_sets(ctx);
	const int rc = conga_results_copy(ctx, dst_device, dst_bytes); // (on the context's stream: behind the launches of BOTH computes)
	swap_result_sets(ctx);
	return rc;
}

int conga_copy_read_depth(conga_ctx *ctx, int16_t *out, int64_t n)
{
	HostSlot *h = ctx ? current(ctx) : nullptr;
	if (!ctx || !out || !ctx->computed || !h || n > h->L || n < 0)
		return CONGA_ERR_INVALID;
	if (ctx->reads_ahead)
		return fail(ctx, CONGA_ERR_INVALID, "conga_copy_read_depth: the reads have been replaced since the compute");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (!ctx->depth_resident || ctx->rd_clobbered) {
		// tuple-space compute: build read_depth[] now; its by-products go to a scratch block, not into the results
		const size_t bytes = ctx->slots.size() * sizeof(Small);
		TRY(ensure(ctx, ctx->d_small_scratch, bytes));
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_small_scratch.p, 0, bytes, ctx->stream));
		TRY(launch_dense_depth(ctx, ptr<Small>(ctx->d_small_scratch), false));
		HIP_TRY(ctx, hipGetLastError());
		ctx->depth_resident = true;
		ctx->rd_clobbered = false;
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
