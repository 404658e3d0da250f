// engine_layout.hip.h -- what stands between the caller's arrays and the kernels: the per-sample table (prepare_sample: where
// each chromosome's tuples lie, the tuple pass's geometry) and the per-layout one (prepare_layout: interval ordering, work items,
// track rows, GC bases per bin, the split-read indexes) -- the host half of calc_mean_per_chr / find_depths' set-up
// (read_distribution.c:49-84, likelihood.c:290-371).  Part of conga_api.hip's one translation unit.
#pragma once

namespace {

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
	if (ctx->knobs.tuple_blocks_per_cu > 0) // measurement switch
		blocks = ctx->n_cu * ctx->knobs.tuple_blocks_per_cu;
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
			sl.pres_off = h.pres_off;
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
	int64_t refn_words = 0, kpos_total = 0, sat_total = 0, pres_words = 0;
	int n_ref = 0;
	for (int s = 0; s < n_slots; s++) {
		HostSlot &h = ctx->slots[s];
		h.kidx = -1;
		if (!h.ref.empty()) {
			h.kidx = n_ref++;
			h.refn_off = refn_words;
			h.kpos_off = kpos_total;
			h.pres_off = pres_words;
			pres_words += ((h.L + 31) / 32 + 63) & ~(int64_t) 63;
			h.sat_off = sat_total;
			refn_words += ((h.L + kRefPadBases + 7) / 8 + 63) & ~(int64_t) 63;
			kpos_total += (h.L + 63) & ~(int64_t) 63;
			sat_total += (int64_t) h.sat_start.size();
		}
	}
	ctx->any_ref = n_ref > 0;
	ctx->refn_words = refn_words;
	ctx->kpos_total = kpos_total;
	ctx->pres_words = pres_words;
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
		if (ctx->knobs.depth_tiles_per_block > 0) // measurement switch
			tiles_per_block = ctx->knobs.depth_tiles_per_block;
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
			// two bits per position: what its own 10-mer's bucket would answer (split_map.hip.h: solo / echo)
			TRY(ensure(ctx, ctx->d_kmer_pres, (size_t) std::max<int64_t>(ctx->pres_words, 1) * sizeof(uint2) + 256));
			HIP_TRY(ctx, hipMemsetAsync(ctx->d_kmer_pres.p, 0, (size_t) std::max<int64_t>(ctx->pres_words, 1) * sizeof(uint2), ctx->stream));
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
			ctx->pres_built = false; // (the solo / echo bits of the new text: made before the second split-read launch on it)
			ctx->sr_launches_on_index = 0;
			if (ctx->knobs.timing)
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
		if (ctx->knobs.chain_long_windows >= 0) // test knobs: the class borders (tests/soak.py --batch draws them)
			long_min = std::max(1, ctx->knobs.chain_long_windows);
		if (ctx->knobs.chain_serial_windows >= 0)
			serial_max = ctx->knobs.chain_serial_windows;
		int32_t block_min = kChainBlockWindows;
		if (ctx->knobs.chain_block_windows >= 0)
			block_min = std::max(1, ctx->knobs.chain_block_windows);
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

} // namespace
