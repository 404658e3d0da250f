// conga_api.hip -- C-ABI of include/conga_hip.h over the gfx950 kernels in kernels.hip.h.
//
// One context = one GPU, one HIP stream, one chromosome in flight.  Everything the kernels need
// stays resident in HBM between conga_chrom_begin() and the next one, so conga_chrom_compute()
// can be replayed on the same inputs (bench.py times exactly that).
//
// There is deliberately no CPU fallback anywhere in this file: without a HIP device
// conga_create() returns NULL / CONGA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/conga_hip.h"
#include "kernels.hip.h"

using namespace conga;

namespace {

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
};

constexpr size_t kStagingTuples = (size_t) 1 << 22; // per pinned buffer
constexpr int kStagingRing = 2;

// small device block read back after every compute
struct SmallBlock {
	uint32_t status;
	uint32_t pad;
	unsigned long long counters[CNT_N];
	unsigned long long hist_sum[kGcBins];
	unsigned long long hist_bases[kGcBins];
	float E[kGcBins];
	float pad2;
};

struct Staging {
	int32_t *pos = nullptr;
	uint8_t *mapq = nullptr;
	hipEvent_t copied = nullptr; // H2D of the last commit from this buffer
	bool in_flight = false;
};

} // namespace

struct conga_ctx {
	int device = 0;
	int n_cu = 256;
	hipStream_t stream = nullptr;
	conga_opts opts{};
	std::string err;

	// chromosome geometry
	bool chrom_open = false;
	int64_t L = 0, n_win = 0, n_tiles = 0;
	int32_t step = 100, tile_win = 0;
	bool gc_aliased = false;

	// reads
	int64_t n_reads = 0;
	Staging staging[kStagingRing];
	int staging_next = 0;  // buffer the next conga_reads_staging() hands out
	int staging_cur = -1;  // buffer handed out and not yet committed

	// intervals (host copies; [0] = dels, [1] = dups)
	std::vector<int32_t> iv_start[2], iv_end[2], iv_support[2];
	bool iv_given[2] = {false, false};
	bool iv_dirty = true;
	int64_t n_iv = 0, n_items = 0, n_long = 0;

	// mappability rows
	bool has_map = false, map_sorted = false;
	int64_t n_map_rows = 0;

	// device buffers
	DevBuf d_pos, d_mapq, d_tile_start, d_rd, d_gc_hist, d_gc_like, d_small, d_map, d_winner, d_map_start,
			d_map_end, d_map_val, d_iv_start, d_iv_end, d_iv_type, d_order, d_observed, d_item_iv, d_item_start,
			d_item_end, d_item_first, d_map_part, d_support, d_results, d_expected;

	// pinned read-back
	SmallBlock *h_small = nullptr;
	conga_result *h_results = nullptr;
	size_t h_results_cap = 0;

	bool computed = false;
	bool support_given = false;
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

#define HIP_TRY(ctx, call)                                                                              \
	do {                                                                                                \
		hipError_t e_ = (call);                                                                         \
		if (e_ != hipSuccess)                                                                           \
			return fail((ctx), (e_ == hipErrorOutOfMemory) ? CONGA_ERR_NOMEM : CONGA_ERR_HIP,          \
					std::string(#call) + ": " + hipGetErrorString(e_));                                \
	} while (0)

int ensure(conga_ctx *ctx, DevBuf &b, size_t bytes, bool keep = false)
{
	if (bytes <= b.cap)
		return CONGA_OK;
	size_t want = std::max(bytes, b.cap + b.cap / 2);
	want = (want + 255) & ~(size_t) 255;
	void *np = nullptr;
	HIP_TRY(ctx, hipMalloc(&np, want));
	if (keep && b.p && b.cap) {
		hipError_t e = hipMemcpyAsync(np, b.p, b.cap, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess)
			e = hipStreamSynchronize(ctx->stream);
		if (e != hipSuccess) {
			(void) hipFree(np);
			return fail(ctx, CONGA_ERR_HIP, std::string("grow copy: ") + hipGetErrorString(e));
		}
	} else if (b.p) {
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // nothing in flight may still use the old block
	}
	if (b.p)
		(void) hipFree(b.p);
	b.p = np;
	b.cap = want;
	return CONGA_OK;
}

#define TRY(expr)                \
	do {                         \
		int rc_ = (expr);        \
		if (rc_ != CONGA_OK)     \
			return rc_;          \
	} while (0)

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

// Build the combined interval arrays, the processing order and the reduce work items.
int prepare_intervals(conga_ctx *ctx)
{
	const size_t nd = ctx->iv_start[0].size(), nu = ctx->iv_start[1].size();
	const size_t n = nd + nu;
	ctx->n_iv = (int64_t) n;
	ctx->n_items = 0;
	ctx->iv_dirty = false;
	if (n == 0)
		return CONGA_OK;

	std::vector<int32_t> start(n), end(n), order(n), item_first(n + 1), support;
	std::vector<uint8_t> type(n);
	for (size_t i = 0; i < nd; i++) {
		start[i] = ctx->iv_start[0][i];
		end[i] = ctx->iv_end[0][i];
		type[i] = CONGA_DELETION;
	}
	for (size_t i = 0; i < nu; i++) {
		start[nd + i] = ctx->iv_start[1][i];
		end[nd + i] = ctx->iv_end[1][i];
		type[nd + i] = CONGA_DUPLICATION;
	}

	// longest chains first, so the lanes of a wave in interval_score_kernel retire together
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
		return (int64_t) end[x] - start[x] > (int64_t) end[y] - start[y];
	});
	// intervals spanning more than kLongWindows GC windows take the wave-cooperative chain
	ctx->n_long = 0;
	for (size_t i = 0; i < n; i++) {
		const int32_t iv = order[i];
		if (end[iv] <= start[iv])
			break;
		const int64_t nw = ((int64_t) end[iv] - 1) / ctx->step - (int64_t) start[iv] / ctx->step + 1;
		if (nw <= kLongWindows)
			break;
		ctx->n_long = (int64_t) i + 1;
	}

	// reduce work items: [max(start,0), min(end,L)) cut into kItemLen pieces
	std::vector<int32_t> item_iv, item_start, item_end;
	item_iv.reserve(n + n / 2);
	item_start.reserve(n + n / 2);
	item_end.reserve(n + n / 2);
	for (size_t i = 0; i < n; i++) {
		item_first[i] = (int32_t) item_iv.size();
		int64_t s = std::max<int64_t>(start[i], 0), e = std::min<int64_t>(end[i], ctx->L);
		for (int64_t a = s; a < e; a += kItemLen) {
			item_iv.push_back((int32_t) i);
			item_start.push_back((int32_t) a);
			item_end.push_back((int32_t) std::min<int64_t>(a + kItemLen, e));
		}
	}
	item_first[n] = (int32_t) item_iv.size();
	ctx->n_items = (int64_t) item_iv.size();

	TRY(upload(ctx, ctx->d_iv_start, start.data(), n * 4));
	TRY(upload(ctx, ctx->d_iv_end, end.data(), n * 4));
	TRY(upload(ctx, ctx->d_iv_type, type.data(), n));
	TRY(upload(ctx, ctx->d_order, order.data(), n * 4));
	TRY(upload(ctx, ctx->d_item_first, item_first.data(), (n + 1) * 4));
	TRY(upload(ctx, ctx->d_item_iv, item_iv.data(), item_iv.size() * 4));
	TRY(upload(ctx, ctx->d_item_start, item_start.data(), item_start.size() * 4));
	TRY(upload(ctx, ctx->d_item_end, item_end.data(), item_end.size() * 4));
	TRY(ensure(ctx, ctx->d_observed, n * 4));
	TRY(ensure(ctx, ctx->d_map_part, std::max<size_t>(item_iv.size(), 1) * 8));
	TRY(ensure(ctx, ctx->d_results, n * sizeof(conga_result)));
	TRY(ensure(ctx, ctx->d_expected, n * 4));

	ctx->support_given = !ctx->iv_support[0].empty() || !ctx->iv_support[1].empty();
	if (ctx->support_given) {
		support.assign(n, 0);
		for (size_t i = 0; i < ctx->iv_support[0].size() && i < nd; i++)
			support[i] = ctx->iv_support[0][i];
		for (size_t i = 0; i < ctx->iv_support[1].size() && i < nu; i++)
			support[nd + i] = ctx->iv_support[1][i];
		TRY(upload(ctx, ctx->d_support, support.data(), n * 4));
	}

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

} // namespace

// =============================================================================================
extern "C" {

int conga_abi_version(void)
{
	return CONGA_ABI_VERSION;
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

conga_ctx *conga_create(int device, const conga_opts *opts, int *status)
{
	int st_dummy;
	if (!status)
		status = &st_dummy;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
		*status = CONGA_ERR_NO_DEVICE;
		return nullptr;
	}
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
	}
	if (ctx->opts.gc_step > 1024) {
		*status = CONGA_ERR_INVALID;
		delete ctx;
		return nullptr;
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
	if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
		return bail(CONGA_ERR_HIP);
	if (hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming) != hipSuccess)
		return bail(CONGA_ERR_HIP);
	for (int k = 0; k < CONGA_K_COUNT; k++)
		if (hipEventCreate(&ctx->ev_k0[k]) != hipSuccess || hipEventCreate(&ctx->ev_k1[k]) != hipSuccess)
			return bail(CONGA_ERR_HIP);
	if (hipHostMalloc((void **) &ctx->h_small, sizeof(SmallBlock), hipHostMallocDefault) != hipSuccess)
		return bail(CONGA_ERR_NOMEM);
	void *small = nullptr;
	if (hipMalloc(&small, sizeof(SmallBlock)) != hipSuccess)
		return bail(CONGA_ERR_NOMEM);
	ctx->d_small.p = small;
	ctx->d_small.cap = sizeof(SmallBlock);
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
	DevBuf *bufs[] = {&ctx->d_pos, &ctx->d_mapq, &ctx->d_tile_start, &ctx->d_rd, &ctx->d_gc_hist, &ctx->d_gc_like,
			&ctx->d_small, &ctx->d_map, &ctx->d_winner, &ctx->d_map_start, &ctx->d_map_end, &ctx->d_map_val,
			&ctx->d_iv_start, &ctx->d_iv_end, &ctx->d_iv_type, &ctx->d_order, &ctx->d_observed, &ctx->d_item_iv,
			&ctx->d_item_start, &ctx->d_item_end, &ctx->d_item_first, &ctx->d_map_part, &ctx->d_support,
			&ctx->d_results, &ctx->d_expected};
	if (ctx->gc_aliased)
		ctx->d_gc_like = DevBuf();
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
	if (ctx->h_small)
		(void) hipHostFree(ctx->h_small);
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

int conga_chrom_begin(conga_ctx *ctx, int64_t chrom_len, const uint8_t *gc_hist_w, const uint8_t *gc_like_w,
		int64_t n_win)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (chrom_len <= 0 || chrom_len > (int64_t) INT32_MAX - 2 * kDepthMaxTile || !gc_hist_w || !gc_like_w)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_begin: bad length or null GC array");
	const int32_t step = ctx->opts.gc_step;
	if (n_win != (chrom_len + step - 1) / step)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_begin: n_win must be ceil(chrom_len / gc_step)");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));

	ctx->chrom_open = false;
	ctx->computed = false;
	ctx->L = chrom_len;
	ctx->n_win = n_win;
	ctx->step = step;
	// tile = tile_win windows; tile_win * step must be a multiple of 8 (16-byte stores) and fit the LDS tile
	int32_t tw = std::min<int32_t>(kDepthMaxTile / step, 1024);
	tw &= ~7;
	if (tw < 8)
		tw = 8;
	ctx->tile_win = tw;
	const int64_t T = (int64_t) tw * step;
	ctx->n_tiles = (chrom_len + T - 1) / T;

	ctx->n_reads = 0;
	ctx->staging_cur = -1;
	for (int t = 0; t < 2; t++) {
		ctx->iv_start[t].clear();
		ctx->iv_end[t].clear();
		ctx->iv_support[t].clear();
		ctx->iv_given[t] = false;
	}
	ctx->iv_dirty = true;
	ctx->n_iv = ctx->n_items = 0;
	ctx->has_map = false;
	ctx->n_map_rows = 0;

	// GC bytes, padded to a multiple of 4 (interval_score_kernel reads them as words)
	const size_t gc_bytes = ((size_t) n_win + 3) & ~(size_t) 3;
	if (ctx->gc_aliased)
		ctx->d_gc_like = DevBuf();
	ctx->gc_aliased = false;
	TRY(ensure(ctx, ctx->d_gc_hist, gc_bytes));
	HIP_TRY(ctx, hipMemsetAsync(ctx->d_gc_hist.p, 0, gc_bytes, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_gc_hist.p, gc_hist_w, (size_t) n_win, hipMemcpyHostToDevice, ctx->stream));
	if (gc_like_w == gc_hist_w) {
		free_buf(ctx->d_gc_like);
		ctx->d_gc_like = ctx->d_gc_hist;
		ctx->gc_aliased = true;
	} else {
		TRY(ensure(ctx, ctx->d_gc_like, gc_bytes));
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_gc_like.p, 0, gc_bytes, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->d_gc_like.p, gc_like_w, (size_t) n_win, hipMemcpyHostToDevice, ctx->stream));
	}
	TRY(ensure(ctx, ctx->d_rd, ((size_t) chrom_len * 2 + 15) & ~(size_t) 15));
	TRY(ensure(ctx, ctx->d_tile_start, ((size_t) ctx->n_tiles + 2) * 4));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // caller's GC arrays are free again
	ctx->chrom_open = true;
	return CONGA_OK;
}

int conga_reads_staging(conga_ctx *ctx, conga_read_staging *out)
{
	if (!ctx || !out)
		return CONGA_ERR_INVALID;
	if (!ctx->chrom_open)
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
	if (!ctx->chrom_open || ctx->staging_cur < 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_commit: call conga_reads_staging first");
	if (n > kStagingTuples)
		return fail(ctx, CONGA_ERR_INVALID, "conga_reads_commit: n exceeds the staging capacity");
	if ((uint64_t) ctx->n_reads + n >= 0xFFFFFFF0ull)
		return fail(ctx, CONGA_ERR_RANGE, "conga_reads_commit: more than 2^32 reads on one chromosome");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	Staging &s = ctx->staging[ctx->staging_cur];
	ctx->staging_cur = -1;
	if (n == 0)
		return CONGA_OK;
	const size_t total = (size_t) ctx->n_reads + n;
	if (total * 4 > ctx->d_pos.cap || total > ctx->d_mapq.cap) {
		const size_t want = std::max(total, (size_t) 1 << 22);
		TRY(ensure(ctx, ctx->d_pos, want * 4, true));
		TRY(ensure(ctx, ctx->d_mapq, want, true));
	}
	HIP_TRY(ctx, hipMemcpyAsync(ptr<int32_t>(ctx->d_pos) + ctx->n_reads, s.pos, n * 4, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(ptr<uint8_t>(ctx->d_mapq) + ctx->n_reads, s.mapq, n, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipEventRecord(s.copied, ctx->stream));
	s.in_flight = true;
	ctx->n_reads += (int64_t) n;
	ctx->staging_next = (ctx->staging_next + 1) % kStagingRing;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_mappability(conga_ctx *ctx, const int32_t *start, const int32_t *end, const float *val, size_t m)
{
	if (!ctx || (m && (!start || !end || !val)))
		return CONGA_ERR_INVALID;
	if (!ctx->chrom_open)
		return fail(ctx, CONGA_ERR_INVALID, "conga_mappability: no chromosome open");
	if (m > (size_t) INT32_MAX)
		return fail(ctx, CONGA_ERR_RANGE, "conga_mappability: too many rows");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	// sorted, abutting-at-most rows can be painted in one pass (kernels.hip.h: paint_sorted_kernel)
	bool sorted = true;
	for (size_t k = 0; k < m && sorted; k++) {
		if (end[k] < start[k])
			sorted = false;
		if (k + 1 < m && (start[k + 1] < end[k] || start[k + 1] < start[k]))
			sorted = false;
	}
	ctx->map_sorted = sorted;
	ctx->n_map_rows = (int64_t) m;
	ctx->has_map = true;
	ctx->computed = false;
	TRY(upload(ctx, ctx->d_map_start, start, m * 4));
	TRY(upload(ctx, ctx->d_map_end, end, m * 4));
	TRY(upload(ctx, ctx->d_map_val, val, m * 4));
	TRY(ensure(ctx, ctx->d_map, ((size_t) ctx->L * 4 + 15) & ~(size_t) 15));
	if (!sorted)
		TRY(ensure(ctx, ctx->d_winner, (size_t) ctx->L * 4));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // caller's arrays are free again
	return CONGA_OK;
}

int conga_intervals(conga_ctx *ctx, char type, const int32_t *start, const int32_t *end, size_t n)
{
	if (!ctx || (n && (!start || !end)))
		return CONGA_ERR_INVALID;
	const int t = type_index(type);
	if (t < 0)
		return fail(ctx, CONGA_ERR_INVALID, "conga_intervals: type must be 'D' or 'E'");
	if (!ctx->chrom_open)
		return fail(ctx, CONGA_ERR_INVALID, "conga_intervals: no chromosome open");
	if (n > (size_t) 1 << 28)
		return fail(ctx, CONGA_ERR_RANGE, "conga_intervals: too many intervals");
	for (size_t i = 0; i < n; i++) {
		// the reference would read before/after its arrays for such rows
		if (start[i] < 0 || end[i] < start[i])
			return fail(ctx, CONGA_ERR_RANGE, "conga_intervals: interval with start < 0 or end < start");
	}
	ctx->iv_start[t].assign(start, start + n);
	ctx->iv_end[t].assign(end, end + n);
	ctx->iv_support[t].clear();
	ctx->iv_given[t] = true;
	ctx->iv_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_split_support(conga_ctx *ctx, char type, const int32_t *support, size_t n)
{
	if (!ctx || (n && !support))
		return CONGA_ERR_INVALID;
	const int t = type_index(type);
	if (t < 0 || n != ctx->iv_start[t].size())
		return fail(ctx, CONGA_ERR_INVALID, "conga_split_support: type / count does not match conga_intervals");
	ctx->iv_support[t].assign(support, support + n);
	ctx->iv_dirty = true;
	ctx->computed = false;
	return CONGA_OK;
}

int conga_chrom_compute(conga_ctx *ctx)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (!ctx->chrom_open)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_compute: no chromosome open");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (ctx->iv_dirty)
		TRY(prepare_intervals(ctx));
	if (ctx->n_reads == 0) {
		TRY(ensure(ctx, ctx->d_pos, 256));
		TRY(ensure(ctx, ctx->d_mapq, 256));
	}

	hipStream_t st = ctx->stream;
	SmallBlock *small = ptr<SmallBlock>(ctx->d_small);
	const bool unsorted_mode = (ctx->opts.flags & CONGA_FLAG_READS_UNSORTED) != 0;
	for (int k = 0; k < CONGA_K_COUNT; k++)
		ctx->ev_used[k] = false;

	HIP_TRY(ctx, hipMemsetAsync(small, 0, sizeof(SmallBlock), st));

	if (!unsorted_mode) {
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_tile_start.p, 0xFF, ((size_t) ctx->n_tiles + 2) * 4, st));
		{
			KernelTimer t(ctx, CONGA_K_INGEST);
			if (ctx->n_reads > 0) {
				const int grid = (int) std::min<int64_t>((ctx->n_reads + 255) / 256, (int64_t) ctx->n_cu * 8);
				hipLaunchKernelGGL(ingest_kernel, dim3(grid), dim3(256), 0, st, ptr<int32_t>(ctx->d_pos), ctx->n_reads,
						ctx->L, ctx->tile_win * ctx->step, ctx->n_tiles, ptr<uint32_t>(ctx->d_tile_start),
						&small->status, small->counters);
			}
		}
		{
			KernelTimer t(ctx, CONGA_K_DEPTH);
			DepthArgs a;
			a.pos = ptr<int32_t>(ctx->d_pos);
			a.mapq = ptr<uint8_t>(ctx->d_mapq);
			a.n = ctx->n_reads;
			a.tile_start = ptr<uint32_t>(ctx->d_tile_start);
			a.rd = ptr<int16_t>(ctx->d_rd);
			a.L = ctx->L;
			a.gc_hist = ptr<uint8_t>(ctx->d_gc_hist);
			a.n_win = ctx->n_win;
			a.step = ctx->step;
			a.tile_win = ctx->tile_win;
			a.mq_threshold = ctx->opts.mq_threshold;
			a.n_tiles = ctx->n_tiles;
			a.hist_sum = small->hist_sum;
			a.hist_bases = small->hist_bases;
			a.counters = small->counters;
			a.status = &small->status;
			const int grid = (int) std::min<int64_t>(ctx->n_tiles, (int64_t) ctx->n_cu * 4);
			hipLaunchKernelGGL(depth_tile_kernel, dim3(grid), dim3(kDepthBlock), 0, st, a);
		}
	} else {
		KernelTimer t(ctx, CONGA_K_DEPTH);
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_rd.p, 0, ((size_t) ctx->L * 2 + 15) & ~(size_t) 15, st));
		if (ctx->n_reads > 0) {
			const int grid = (int) std::min<int64_t>((ctx->n_reads + 255) / 256, (int64_t) ctx->n_cu * 8);
			hipLaunchKernelGGL(depth_atomic_kernel, dim3(grid), dim3(256), 0, st, ptr<int32_t>(ctx->d_pos),
					ptr<uint8_t>(ctx->d_mapq), ctx->n_reads, ctx->L, ctx->opts.mq_threshold, ptr<int16_t>(ctx->d_rd),
					small->counters);
		}
		const int64_t n_w = (ctx->L + ctx->step - 1) / ctx->step;
		const int grid = (int) std::min<int64_t>((n_w + 255) / 256, (int64_t) ctx->n_cu * 8);
		hipLaunchKernelGGL(gc_hist_kernel, dim3(grid), dim3(256), 0, st, ptr<int16_t>(ctx->d_rd), ctx->L,
				ptr<uint8_t>(ctx->d_gc_hist), ctx->n_win, ctx->step, small->hist_sum, small->hist_bases);
	}

	{
		KernelTimer t(ctx, CONGA_K_EXPECTED);
		hipLaunchKernelGGL(expected_table_kernel, dim3(1), dim3(128), 0, st, small->hist_sum, small->hist_bases, small->E);
	}

	// the reference paints the track only when the chromosome has at least one kept SV
	// (likelihood.c:332-336 returns before :352-356)
	if (ctx->has_map && ctx->n_iv > 0) {
		KernelTimer t(ctx, CONGA_K_PAINT);
		if (ctx->map_sorted) {
			const int64_t tile = 256 * 4;
			const int grid = (int) std::min<int64_t>((ctx->L + tile - 1) / tile, (int64_t) ctx->n_cu * 16);
			hipLaunchKernelGGL(paint_sorted_kernel, dim3(grid), dim3(256), 0, st, ptr<int32_t>(ctx->d_map_start),
					ptr<int32_t>(ctx->d_map_end), ptr<float>(ctx->d_map_val), ctx->n_map_rows, ptr<float>(ctx->d_map),
					ctx->L);
		} else {
			HIP_TRY(ctx, hipMemsetAsync(ctx->d_winner.p, 0xFF, (size_t) ctx->L * 4, st));
			if (ctx->n_map_rows > 0) {
				const int grid = (int) std::min<int64_t>((ctx->n_map_rows + 3) / 4, (int64_t) ctx->n_cu * 8);
				hipLaunchKernelGGL(paint_winner_kernel, dim3(grid), dim3(256), 0, st, ptr<int32_t>(ctx->d_map_start),
						ptr<int32_t>(ctx->d_map_end), ctx->n_map_rows, ptr<int32_t>(ctx->d_winner), ctx->L);
			}
			const int grid = (int) std::min<int64_t>((ctx->L + 255) / 256, (int64_t) ctx->n_cu * 16);
			hipLaunchKernelGGL(paint_resolve_kernel, dim3(grid), dim3(256), 0, st, ptr<int32_t>(ctx->d_winner),
					ptr<float>(ctx->d_map_val), ptr<float>(ctx->d_map), ctx->L);
		}
	}

	if (ctx->n_iv > 0) {
		HIP_TRY(ctx, hipMemsetAsync(ctx->d_observed.p, 0, (size_t) ctx->n_iv * 4, st));
		if (ctx->n_items > 0) {
			KernelTimer t(ctx, CONGA_K_REDUCE);
			ReduceArgs a;
			a.rd = ptr<int16_t>(ctx->d_rd);
			a.map = ctx->has_map ? ptr<float>(ctx->d_map) : nullptr;
			a.item_iv = ptr<int32_t>(ctx->d_item_iv);
			a.item_start = ptr<int32_t>(ctx->d_item_start);
			a.item_end = ptr<int32_t>(ctx->d_item_end);
			a.n_items = ctx->n_items;
			a.observed = ptr<int32_t>(ctx->d_observed);
			a.map_part = ptr<double>(ctx->d_map_part);
			const int waves_per_block = 256 / kWave;
			const int grid = (int) ((ctx->n_items + waves_per_block - 1) / waves_per_block);
			hipLaunchKernelGGL(interval_reduce_kernel, dim3(grid), dim3(256), 0, st, a);
		}
		if (ctx->n_long > 0) {
			KernelTimer t(ctx, CONGA_K_CHAIN);
			ChainArgs c;
			c.start = ptr<int32_t>(ctx->d_iv_start);
			c.end = ptr<int32_t>(ctx->d_iv_end);
			c.order = ptr<int32_t>(ctx->d_order);
			c.n_long = ctx->n_long;
			c.gc_like = ptr<uint8_t>(ctx->d_gc_like);
			c.n_win = ctx->n_win;
			c.step = ctx->step;
			c.E = small->E;
			c.expected = ptr<float>(ctx->d_expected);
			const int grid = (int) ((ctx->n_long + 3) / 4);
			hipLaunchKernelGGL(chain_long_kernel, dim3(grid), dim3(256), 0, st, c);
		}
		{
			KernelTimer t(ctx, CONGA_K_SCORE);
			ScoreArgs a;
			a.start = ptr<int32_t>(ctx->d_iv_start);
			a.end = ptr<int32_t>(ctx->d_iv_end);
			a.type = ptr<uint8_t>(ctx->d_iv_type);
			a.order = ptr<int32_t>(ctx->d_order);
			a.n_iv = ctx->n_iv;
			a.gc_like = ptr<uint8_t>(ctx->d_gc_like);
			a.n_win = ctx->n_win;
			a.step = ctx->step;
			a.E = small->E;
			a.observed = ptr<int32_t>(ctx->d_observed);
			a.map_part = ctx->has_map ? ptr<double>(ctx->d_map_part) : nullptr;
			a.item_first = ptr<int32_t>(ctx->d_item_first);
			a.support = ctx->support_given ? ptr<int32_t>(ctx->d_support) : nullptr;
			a.has_map = ctx->has_map ? 1 : 0;
			a.n_long = ctx->n_long;
			a.expected_long = ptr<float>(ctx->d_expected);
			a.out = ptr<conga_result>(ctx->d_results);
			const int grid = (int) ((ctx->n_iv + 63) / 64);
			hipLaunchKernelGGL(interval_score_kernel, dim3(grid), dim3(64), 0, st, a);
		}
		HIP_TRY(ctx, hipMemcpyAsync(ctx->h_results, ctx->d_results.p, (size_t) ctx->n_iv * sizeof(conga_result),
				hipMemcpyDeviceToHost, st));
	}
	HIP_TRY(ctx, hipMemcpyAsync(ctx->h_small, small, sizeof(SmallBlock), hipMemcpyDeviceToHost, st));
	HIP_TRY(ctx, hipEventRecord(ctx->ev_done, st));
	HIP_TRY(ctx, hipGetLastError());
	ctx->computed = true;
	return CONGA_OK;
}

int conga_chrom_fetch(conga_ctx *ctx, conga_result *dels, conga_result *dups, float expected_rd[101],
		conga_chrom_stats *stats)
{
	if (!ctx)
		return CONGA_ERR_INVALID;
	if (!ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_fetch: nothing computed");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipEventSynchronize(ctx->ev_done));
	const SmallBlock &sb = *ctx->h_small;
	if (sb.status & kStatusUnsorted)
		return fail(ctx, CONGA_ERR_UNSORTED,
				"reads were committed out of position order; pass CONGA_FLAG_READS_UNSORTED to accept that");
	const size_t nd = ctx->iv_start[0].size(), nu = ctx->iv_start[1].size();
	if ((nd && !dels) || (nu && !dups))
		return fail(ctx, CONGA_ERR_INVALID, "conga_chrom_fetch: result array missing");
	if (nd)
		memcpy(dels, ctx->h_results, nd * sizeof(conga_result));
	if (nu)
		memcpy(dups, ctx->h_results + nd, nu * sizeof(conga_result));
	if (expected_rd)
		memcpy(expected_rd, sb.E, kGcBins * sizeof(float));
	if (stats) {
		memset(stats, 0, sizeof *stats);
		stats->reads_committed = ctx->n_reads;
		stats->reads_counted = (int64_t) sb.counters[CNT_COUNTED];
		stats->reads_out_of_range = (int64_t) sb.counters[CNT_OUT_OF_RANGE];
		long long total = 0;
		for (int g = 0; g < kGcBins; g++) {
			stats->rd_per_gc[g] = (int64_t) sb.hist_sum[g];
			stats->window_per_gc[g] = (int64_t) sb.hist_bases[g];
			total += (long long) sb.hist_sum[g];
		}
		stats->rd_sum = total;
		stats->mean = (float) ((double) total / (double) ctx->L); // read_distribution.c:39
		stats->n_kernels = CONGA_K_COUNT;
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

int conga_results_device(conga_ctx *ctx, void **dev_ptr, size_t *n_dels, size_t *n_dups)
{
	if (!ctx || !dev_ptr)
		return CONGA_ERR_INVALID;
	if (!ctx->computed)
		return fail(ctx, CONGA_ERR_INVALID, "conga_results_device: nothing computed");
	*dev_ptr = ctx->n_iv ? ctx->d_results.p : nullptr;
	if (n_dels)
		*n_dels = ctx->iv_start[0].size();
	if (n_dups)
		*n_dups = ctx->iv_start[1].size();
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
	return CONGA_OK;
}

int conga_copy_read_depth(conga_ctx *ctx, int16_t *out, int64_t n)
{
	if (!ctx || !out || !ctx->computed || n > ctx->L || n < 0)
		return CONGA_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_rd.p, (size_t) n * 2, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CONGA_OK;
}

int conga_copy_mappability(conga_ctx *ctx, float *out, int64_t n)
{
	if (!ctx || !out || !ctx->computed || !ctx->has_map || ctx->n_iv == 0 || n > ctx->L || n < 0)
		return CONGA_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_map.p, (size_t) n * 4, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return CONGA_OK;
}

} // extern "C"
