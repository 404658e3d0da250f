// bam_data.cpp -- read_bam / count_reads_bam / find_SVs re-hosted on the C-ABI of include/conga_hip.h.
//
// Order of work differs from the reference on purpose: all chromosomes of a context are handed to it as ONE batch
// (CONGA_FLAG_BATCH) and computed by a single conga_chrom_compute(), then written out in annotation order.
// Outputs are byte-identical to doing begin / finish per chromosome; the GPU just gets launches that fill it.
// `--gpus N` runs N such contexts, one host thread and one HIP device each, over a longest-first partition of the
// chromosomes; the files are still written by the calling thread in annotation order.
#include "bam_data.h"
#include "knobs.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/conga_hip.h"
#include "likelihood.h"
#include "reads.h"
#include "svs.h"

namespace conga_host {

FILE *logFile = nullptr;

namespace {

// print_error (common.c:102-108)
[[noreturn]] void print_error(const std::string &msg)
{
	fprintf(stderr, "\n%s\n", msg.c_str());
	fprintf(stderr, "Invoke parameter -h for help.\n");
	exit(CONGA_EXIT_COMMON);
}

// safe_fopen (common.c:111-125)
FILE *safe_fopen(const std::string &path, const char *mode)
{
	FILE *f = fopen(path.c_str(), mode);
	if (!f)
		print_error("[CONGA INPUT ERROR] Unable to open file " + path + " in " + (mode[0] == 'w' ? "write" : "read") + " mode.");
	return f;
}

void engine_check(conga_ctx *ctx, int rc, const char *what)
{
	if (rc != CONGA_OK) {
		fprintf(stderr, "\n[CONGA ENGINE ERROR] %s: %s (%s)\n", what, conga_strerror(rc), conga_last_error(ctx));
		exit(CONGA_EXIT_COMMON);
	}
}

// From how much of the file on the decode goes to the GPU when the environment does not say: one lane per block costs
// ~0.25 s per call whatever the size (launch floor + upload), a host core inflates and walks a 40 KB block in ~0.37 ms,
// so the break-even is ~680 blocks, 27 MB of file, per core this worker may use.
uint64_t gpu_bam_min_piece()
{
	return (uint64_t) 27000000 * (uint64_t) std::max(1, usable_cpus() / reader_share());
}

// (measurement switch: the sample loop's events on the clock of the engine's own trace lines, conga_amd/csrc/bz_sched.h)
void host_trace(const char *what, size_t k = 0)
{
	if (!knobs().trace)
		return;
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	fprintf(stderr, "[bz %9.2f] cli: %s (sample %zu)\n", ms - 100000.0 * (double) (long long) (ms / 100000.0), what, k);
}

// count_reads_bam (bam_data.c:192-221), producer side: records go straight into the pinned staging ring.
// With split reads enabled (--rp and --dups) every record is also handed over whole (split_read.c:206-354).
// `chrom`: the chromosome of the engine context the records go to (the one begun last in a fresh context; any of them when
// a kept context takes the next sample of a cohort chromosome by chromosome).
int64_t count_reads_bam(conga_ctx *ctx, read_source *src, int chr_index_bam, int64_t chrom_len, bool split_reads, int chrom)
{
	std::string err;
	if (!src->begin(chr_index_bam, chrom_len, &err)) {
		fprintf(stderr, "Error: Iterator cannot be loaded (bam_itr_queryi)\n");
		exit(1);
	}
	int64_t cnt = 0;
	const int gpu_bam = knobs().gpu_bam; // CONGA_GPU_BAM: 0 never, 1 always, -1 (unset): when it pays
	if (gpu_bam != 0) {
		// a BAM with an index: its compressed blocks go to the GPU as they are and are inflated and walked there
		// (conga_reads_bgzf); with split reads the engine then reads the records where they lie in HBM.  Anything that does
		// not check out falls through to the host decoders below.  A launch lasts at least one block's few milliseconds, so it
		// pays from some size of the piece on (a whole low-coverage genome, a chromosome of a deep sample: the default asks
		// for 27 MB of file per host core).
		file_piece bytes;
		std::vector<conga_bgzf_block> blocks;
		std::vector<conga_bam_segment> segments;
		const uint64_t min_piece = gpu_bam >= 0 ? 0 : gpu_bam_min_piece();
		if (src->device_plan({device_target{chr_index_bam, chrom_len, chrom}}, min_piece, &bytes, &blocks, &segments, &err)) {
			std::vector<uint64_t> per_chrom((size_t) conga_chrom_count(ctx), 0);
			const int rc = bytes.data ? conga_reads_bgzf(ctx, bytes.data, bytes.size, blocks.data(), blocks.size(), segments.data(), segments.size(),
					per_chrom.data())
					: conga_reads_bgzf_fd(ctx, bytes.fd, bytes.file_off, bytes.size, blocks.data(), blocks.size(), segments.data(), segments.size(),
					per_chrom.data());
			if (rc == CONGA_OK)
				return (int64_t) per_chrom[(size_t) chrom];
			if (rc != CONGA_ERR_DATA)
				engine_check(ctx, rc, "conga_reads_bgzf");
			fprintf(stderr, "\n[CONGA] decoding on the host: %s\n", conga_last_error(ctx));
		}
		err.clear();
	}
	if (split_reads) {
		for (;;) {
			conga_split_staging ss;
			engine_check(ctx, conga_split_reads_staging(ctx, &ss), "conga_split_reads_staging");
			full_batch fb = {ss.pos, ss.mapq, ss.flag, ss.l_qseq, ss.data_off, ss.data, ss.capacity_reads, ss.capacity_bytes, 0, 0};
			if (!src->next_full(&fb, &err))
				print_error("[CONGA INPUT ERROR] " + err);
			// the depth side gets the same records
			for (size_t done = 0; done < fb.n_reads || (done == 0 && fb.n_reads == 0);) {
				conga_read_staging stg;
				engine_check(ctx, conga_reads_staging(ctx, &stg), "conga_reads_staging");
				const size_t k = std::min(stg.capacity, fb.n_reads - done);
				if (k) {
					memcpy(stg.pos, fb.pos + done, k * sizeof(int32_t));
					memcpy(stg.mapq, fb.mapq + done, k);
				}
				engine_check(ctx, conga_reads_commit(ctx, k), "conga_reads_commit");
				done += k;
				if (fb.n_reads == 0)
					break;
			}
			engine_check(ctx, conga_split_reads_commit(ctx, fb.n_reads, fb.n_bytes), "conga_split_reads_commit");
			cnt += (int64_t) fb.n_reads;
			if (fb.n_reads == 0)
				break;
		}
		return cnt;
	}
	{
		// a BAM with an index: the whole chromosome decoded by several readers at once (reads.h: read_all)
		std::vector<int32_t> all_pos;
		std::vector<uint8_t> all_mapq;
		if (src->read_all(chr_index_bam, chrom_len, std::max(1, usable_cpus() / reader_share()), &all_pos, &all_mapq, &err)) {
			for (size_t done = 0;;) {
				conga_read_staging stg;
				engine_check(ctx, conga_reads_staging(ctx, &stg), "conga_reads_staging");
				const size_t k = std::min(stg.capacity, all_pos.size() - done);
				if (k) {
					memcpy(stg.pos, all_pos.data() + done, k * sizeof(int32_t));
					memcpy(stg.mapq, all_mapq.data() + done, k);
				}
				engine_check(ctx, conga_reads_commit(ctx, k), "conga_reads_commit");
				done += k;
				if (done >= all_pos.size())
					break;
			}
			return (int64_t) all_pos.size();
		}
		if (!err.empty())
			print_error("[CONGA INPUT ERROR] " + err);
	}
	for (;;) {
		conga_read_staging stg;
		engine_check(ctx, conga_reads_staging(ctx, &stg), "conga_reads_staging");
		read_batch b;
		if (!src->next(stg.capacity, &b, &err))
			print_error("[CONGA INPUT ERROR] " + err);
		if (b.n) {
			memcpy(stg.pos, b.pos, b.n * sizeof(int32_t));
			memcpy(stg.mapq, b.mapq, b.n);
		}
		engine_check(ctx, conga_reads_commit(ctx, b.n), "conga_reads_commit");
		cnt += (int64_t) b.n;
		if (b.n < stg.capacity)
			break;
	}
	return cnt;
}

// One chromosome of the loop of read_bam (bam_data.c:269-339), from selection to output.
struct chrom_job {
	int chr_index = 0;     // in the annotation
	int chr_index_bam = 0; // in the BAM header
	int64_t L = 0;
	int worker = 0;
	std::string messages;  // progress lines of the reference, in its order (held back when several workers run)
	chrom_svs cs;
	conga_chrom_stats st{};
	bool staged = false;
	bool counts_pending = false; // `messages` holds the placeholders below instead of the two counts of bam_data.c:218
};

// count_reads_bam's closing line prints the reads that passed `qual > mq_threshold` and the running number of split-read
// elements (bam_data.c:201-218, split_read.c:14).  Both are known exactly once the engine has computed, so with a MAPQ
// threshold or with split reads the chromosome's lines are held back and these two marks are filled in afterwards.
constexpr char kMarkReads = '\x01', kMarkSplit = '\x02';

void fill_counts(std::string *text, long long reads, long split_reads)
{
	for (size_t i = 0; i < text->size(); i++) {
		const char c = (*text)[i];
		if (c != kMarkReads && c != kMarkSplit)
			continue;
		const std::string v = (c == kMarkReads) ? std::to_string(reads) : std::to_string(split_reads);
		text->replace(i, 1, v);
		i += v.size() - 1;
	}
}

// Where the reference's stderr progress lines go: straight out with one worker, into the job's buffer with several
// (they are then printed in annotation order once every worker is done).
struct progress {
	std::string *buf; // nullptr: stderr
	void say(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
	{
		va_list ap;
		va_start(ap, fmt);
		if (!buf) {
			vfprintf(stderr, fmt, ap);
		} else {
			char tmp[1024];
			vsnprintf(tmp, sizeof tmp, fmt, ap);
			*buf += tmp;
		}
		va_end(ap);
	}
};

struct worker_timing {
	double ms_create = 0, ms_reads = 0, ms_compute = 0, ms_fetch = 0;
};

// The known SVs of the chromosome begun last and, when there are any, its mappability rows
// (likelihood.c:319-336,352-356; load_mappability_regions, svs.c:317-377, runs only for chromosomes that have SVs).
void attach_intervals(conga_ctx *ctx, const parameters *params, const bed_index &map_bed, const chrom_svs &cs)
{
	std::vector<int32_t> s, e;
	auto hand_over = [&](char type, const std::vector<sv_row> &rows) {
		s.resize(rows.size());
		e.resize(rows.size());
		for (size_t i = 0; i < rows.size(); i++) {
			s[i] = rows[i].start;
			e[i] = rows[i].end;
		}
		engine_check(ctx, conga_intervals(ctx, type, s.data(), e.data(), rows.size()), "conga_intervals");
	};
	hand_over(CONGA_DELETION, cs.dels);
	hand_over(CONGA_DUPLICATION, cs.dups);
	if (params->have_map && cs.dels.size() + cs.dups.size() > 0) {
		auto it = map_bed.rows.find(cs.chr_name);
		std::vector<int32_t> ms, me;
		const float *mv = nullptr;
		if (it != map_bed.rows.end()) {
			ms.resize(it->second.size());
			me.resize(it->second.size());
			for (size_t i = 0; i < it->second.size(); i++) {
				ms[i] = it->second[i].start;
				me[i] = it->second[i].end;
			}
			mv = map_bed.values.at(cs.chr_name).data();
		}
		engine_check(ctx, conga_mappability(ctx, ms.data(), me.data(), mv, ms.size()), "conga_mappability");
	}
}

// What the decode on the GPU needs from one BAM, got ready ahead of time (read_bam_cohort does it for sample k + 1 on a
// thread of its own while sample k is on the GPU): the opened file and, for the chromosomes the run will select in their
// order, the mapped stretch of the file with its block table and start points (bam_reader.cpp: device_plan).
struct planned_input {
	std::unique_ptr<read_source> src;
	std::vector<device_target> targets;
	file_piece bytes;
	std::vector<conga_bgzf_block> blocks;
	std::vector<conga_bam_segment> segments;
	bool planned = false;
	double ms_plan = 0;
	uint64_t ahead_ticket = 0; // conga_reads_bgzf_next_fd's: the engine was told about the bytes before their sample began
};

bool gpu_decode_wanted(const parameters *params)
{
	return knobs().gpu_bam != 0; // CONGA_GPU_BAM: 0 never, 1 always, unset: when it pays
}

// the chromosomes read_bam will work on, in its order (bam_data.c:269-291): (annotation index, BAM target)
std::vector<std::pair<int, int>> select_chromosomes(const parameters *params, const sonic *this_sonic, const read_source &src)
{
	std::vector<std::pair<int, int>> sel;
	for (int chr_index = 0; chr_index < this_sonic->number_of_chromosomes; chr_index++) {
		if (chr_index < params->first_chrom)
			chr_index = params->first_chrom;
		if (chr_index > params->last_chrom || chr_index >= this_sonic->number_of_chromosomes)
			break;
		const std::string &name = this_sonic->chromosome_names[chr_index];
		if (name.find('X') != std::string::npos || name.find('Y') != std::string::npos)
			continue;
		const int tid = find_chr_index_bam(name, src);
		if (tid != -1)
			sel.emplace_back(chr_index, tid);
	}
	return sel;
}

// engine: a context that will be given this input next (a cohort's kept one), or nullptr
// front_begun: set once the call in front of this input's own has begun (nullptr: there is none) -- see conga_reads_bgzf_next_go
std::unique_ptr<planned_input> plan_input(const parameters *params, const sonic *this_sonic, const std::string &path, conga_ctx *engine = nullptr,
		std::atomic<bool> *named = nullptr, std::atomic<uint64_t> *ticket_out = nullptr, const std::atomic<bool> *front_begun = nullptr)
{
	std::unique_ptr<planned_input> p(new planned_input);
	std::string err;
	p->src.reset(open_reads(path, &err));
	if (!p->src || !gpu_decode_wanted(params) || params->n_gpus != 1)
		return p; // (an input that does not open is reported by the run itself)
	const auto sel = select_chromosomes(params, this_sonic, *p->src);
	for (size_t i = 0; i < sel.size(); i++)
		p->targets.push_back(device_target{sel[i].second, this_sonic->chromosome_lengths[sel[i].first], (int) i});
	if (p->targets.empty())
		return p;
	const auto t0 = std::chrono::steady_clock::now();
	const int gpu_bam = knobs().gpu_bam;
	planned_input *raw = p.get();
	plan_hooks hooks;
	bool engine_table = false;
	if (engine) {
		hooks.named = [raw, engine, named, ticket_out, front_begun](const file_piece &bytes, const std::vector<uint64_t> &known_starts, uint64_t stop_at) {
			if (bytes.data == nullptr && bytes.fd >= 0)
				(void) conga_reads_bgzf_next_fd(engine, bytes.fd, bytes.file_off, bytes.size, known_starts.data(), known_starts.size(), stop_at,
						&raw->ahead_ticket);
			// Named between two calls, the bytes wait for the next call to begin (it may bring other bytes: the sample in front).  When
			// that sample's call HAS begun -- it may even be over: a sample that was inflated ahead is six milliseconds of call -- the
			// next call is this input's own, and the thread of the calls is about to wait for this plan: the bytes must go now.  (Read
			// AFTER the naming: a call that begins later finds the bytes named and takes them along.)
			if (raw->ahead_ticket && front_begun && front_begun->load())
				(void) conga_reads_bgzf_next_go(engine, raw->ahead_ticket);
			if (ticket_out)
				ticket_out->store(raw->ahead_ticket);
			if (named)
				named->store(true); // (the sample behind this one may name its bytes now)
		};
		hooks.table = [raw, engine, &engine_table](std::vector<conga_bgzf_block> *blocks) {
			const conga_bgzf_block *b = nullptr;
			size_t n = 0;
			if (!raw->ahead_ticket || conga_reads_bgzf_next_table(engine, raw->ahead_ticket, &b, &n) != CONGA_OK || n == 0)
				return false;
			blocks->assign(b, b + n); // (waits until the bytes are up: the sample in front is on the GPU meanwhile)
			engine_table = true;
			return true;
		};
	}
	p->planned = p->src->device_plan(p->targets, gpu_bam >= 0 ? 0 : gpu_bam_min_piece(), &p->bytes, &p->blocks, &p->segments, &err,
			engine ? &hooks : nullptr);
	p->ms_plan = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	if (p->planned && engine && p->ahead_ticket && !engine_table) // (the table was read here: the bytes named above can be inflated ahead with it)
		(void) conga_reads_bgzf_next_blocks(engine, p->ahead_ticket, p->blocks.data(), p->blocks.size());
	return p;
}

// An engine context that outlives one input (read_bam_cohort): what it holds, so that the next sample can tell whether the
// layout is the same (then only the reads are replaced) or has to be handed over again.
struct kept_engine {
	conga_ctx *ctx = nullptr;
	std::string layout_key;
	bool last_sample = false; // no BAM behind this one: the pinned staging can go while the context computes
	bool expect_cohort = false; // three samples or more: their bytes will be named ahead (read_bam_cohort)
	std::atomic<conga_ctx *> early_ctx{nullptr}; // the context as soon as it exists (the first sample's run makes it on a thread of its own)
	std::atomic<bool> *call_begins = nullptr; // raised right in front of this sample's conga_reads_bgzf* call (or when its reads are in
	                                          // without one): from then on the NEXT sample's bytes need not wait for a call (plan_input)
	bed_index dels_bed, dups_bed, map_bed; // --dels / --dups / --mappability as parsed for the first sample
	bool beds_loaded = false;
	// the producer of the packed hand-over and its pinned buffers (hand_over_packed: further samples from the host decoders)
	conga_packer *packer = nullptr;
	uint8_t *packed_pin = nullptr, *mapq_pin = nullptr;
	size_t packed_cap = 0, mapq_cap = 0;
};

// count_reads_bam's tuples of a whole sample, one array per chromosome of the context: where the source holds them (a tuple
// container's mapping) or in vectors of its own (a BAM through the host decoders)
struct whole_sample {
	std::vector<const int32_t *> pos;
	std::vector<const uint8_t *> mapq;
	std::vector<uint64_t> off;
	std::vector<std::vector<int32_t>> own_pos;
	std::vector<std::vector<uint8_t>> own_mapq;
};

// count_reads_bam (bam_data.c:192-221) for every chromosome of a FURTHER sample of a cohort at once, host decoders: the positions go
// through the library's producer (conga_packer_start_v: host threads turn them into differences of a few bits in pinned memory, 1.25
// bytes per read at 1x instead of 4) and over the link in one copy (conga_sample_reads_packed) -- no staging ring, no pass over the
// tuples on this thread.  false: not this way (a sample whose positions are not differences of
// sorted ones, no pinned memory to be had) -- the ring, chromosome by chromosome, as for the first sample.  `ws` must live until the
// compute's sync.
bool hand_over_packed(conga_ctx *ctx, kept_engine *keep, read_source *src, const parameters *params, const std::vector<chrom_job *> &mine,
		whole_sample *ws)
{
	const auto t0 = std::chrono::steady_clock::now();
	auto ms_since = [](std::chrono::steady_clock::time_point t) {
		return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
	};
	const size_t nc = mine.size();
	ws->pos.assign(nc, nullptr);
	ws->mapq.assign(nc, nullptr);
	ws->off.assign(nc + 1, 0);
	ws->own_pos.resize(nc);
	ws->own_mapq.resize(nc);
	const int threads = std::max(1, usable_cpus() / reader_share());
	std::string err;
	for (size_t i = 0; i < nc; i++) {
		size_t n = 0;
		if (!src->whole(mine[i]->chr_index_bam, mine[i]->L, &ws->pos[i], &ws->mapq[i], &n, &err)) {
			if (err.empty() && src->read_all(mine[i]->chr_index_bam, mine[i]->L, threads, &ws->own_pos[i], &ws->own_mapq[i], &err)) {
				ws->pos[i] = ws->own_pos[i].data();
				ws->mapq[i] = ws->own_mapq[i].data();
				n = ws->own_pos[i].size();
			} else if (!err.empty())
				print_error("[CONGA INPUT ERROR] " + err);
			else { // a source that is read front to back (a BAM without an index, a small one): the records as they come
				if (!src->begin(mine[i]->chr_index_bam, mine[i]->L, &err)) {
					fprintf(stderr, "Error: Iterator cannot be loaded (bam_itr_queryi)\n");
					exit(1);
				}
				constexpr size_t kAtOnce = (size_t) 1 << 20;
				for (;;) {
					read_batch b;
					if (!src->next(kAtOnce, &b, &err))
						print_error("[CONGA INPUT ERROR] " + err);
					ws->own_pos[i].insert(ws->own_pos[i].end(), b.pos, b.pos + b.n);
					ws->own_mapq[i].insert(ws->own_mapq[i].end(), b.mapq, b.mapq + b.n);
					if (b.n < kAtOnce)
						break;
				}
				ws->pos[i] = ws->own_pos[i].data();
				ws->mapq[i] = ws->own_mapq[i].data();
				n = ws->own_pos[i].size();
			}
		}
		ws->off[i + 1] = ws->off[i] + n;
	}
	const double ms_arrays = ms_since(t0);
	const uint64_t total = ws->off[nc];
	if (!keep->packer && !(keep->packer = conga_packer_create(std::min(16, threads))))
		return false;
	// (room for one exception in sixteen reads: a position-sorted sample has one in thousands at the width the producer picks)
	const size_t need = conga_pack_bound(total, (size_t) (total / 16) + 1024);
	if (need > keep->packed_cap) {
		conga_host_free(ctx, keep->packed_pin);
		keep->packed_cap = need + need / 8; // (samples of a cohort differ a little)
		if (!(keep->packed_pin = (uint8_t *) conga_host_alloc(ctx, keep->packed_cap))) {
			keep->packed_cap = 0;
			return false;
		}
	}
	const bool need_mapq = params->mq_threshold >= 0; // (otherwise the bytes are never looked at: include/conga_hip.h)
	if (need_mapq && total > keep->mapq_cap) {
		conga_host_free(ctx, keep->mapq_pin);
		keep->mapq_cap = (size_t) total + (size_t) total / 8;
		if (!(keep->mapq_pin = (uint8_t *) conga_host_alloc(ctx, keep->mapq_cap))) {
			keep->mapq_cap = 0;
			return false;
		}
	}
	const auto t1 = std::chrono::steady_clock::now();
	if (conga_packer_start_v(keep->packer, ws->pos.data(), ws->off.data(), (int) nc, 0, keep->packed_pin, keep->packed_cap) != CONGA_OK)
		return false;
	if (need_mapq) // beside the encode
		for (size_t i = 0; i < nc; i++)
			if (ws->off[i + 1] > ws->off[i])
				memcpy(keep->mapq_pin + ws->off[i], ws->mapq[i], (size_t) (ws->off[i + 1] - ws->off[i]));
	int width = 0;
	size_t n_esc = 0, bytes = 0;
	if (conga_packer_finish(keep->packer, &width, &n_esc, &bytes) != CONGA_OK)
		return false; // (more exceptions than fit: not a sorted sample -- through the ring the engine says what is wrong with it)
	const double ms_encode = ms_since(t1);
	engine_check(ctx, conga_sample_reads_packed(ctx, keep->packed_pin, width, nullptr, nullptr, n_esc, need_mapq ? keep->mapq_pin : nullptr,
			ws->off.data(), (int) nc), "conga_sample_reads_packed");
	if (knobs().timing)
		fprintf(stderr, "\n[timing] packed hand-over: %llu reads of %zu chromosomes, arrays %.1f ms, conga_packer (%d threads) %.1f ms -> %d-bit differences, "
				"%zu exceptions, %.1f MB over the link (%.1f MB as int32)\n", (unsigned long long) total, nc, ms_arrays, conga_packer_threads(keep->packer),
				ms_encode, width, n_esc, (double) (bytes + (need_mapq ? total : 0)) / 1e6, (double) total * (need_mapq ? 5 : 4) / 1e6);
	return true;
}

std::string layout_key_of(const std::vector<chrom_job *> &mine)
{
	std::string key;
	for (const chrom_job *j : mine) {
		key += j->cs.chr_name + ":" + std::to_string(j->L) + ":" + std::to_string(j->cs.dels.size()) + ":" + std::to_string(j->cs.dups.size());
		if (!j->cs.dels.empty())
			key += ":" + std::to_string(j->cs.dels.front().start) + "-" + std::to_string(j->cs.dels.back().end);
		if (!j->cs.dups.empty())
			key += ":" + std::to_string(j->cs.dups.front().start) + "-" + std::to_string(j->cs.dups.back().end);
		key += ";";
	}
	return key;
}

// The work of one context: stage every chromosome of `mine` (annotation order), one batch compute, fetch.
// One host thread per context, one context per GPU; chromosomes are independent in the
// reference (bam_data.c:269-339), so no worker ever needs another's data.
void run_worker(const parameters *params, const sonic *this_sonic, read_source *src, int device, bool buffered,
		const bed_index &map_bed, std::vector<chrom_job *> &mine, bool announce_compute, worker_timing *wt, kept_engine *keep = nullptr,
		planned_input *pre = nullptr)
{
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms_since = [&](std::chrono::steady_clock::time_point t) {
		return std::chrono::duration<double, std::milli>(now() - t).count();
	};
	const auto t_create = now();
	struct joined_thread {
		std::thread t;
		~joined_thread()
		{
			if (t.joinable())
				t.join();
		}
	} releaser;
	conga_opts opts;
	memset(&opts, 0, sizeof opts);
	opts.struct_size = sizeof opts;
	opts.mq_threshold = params->mq_threshold;
	opts.gc_step = this_sonic->gc_step;
	opts.flags = CONGA_FLAG_BATCH;
	{
		if (knobs().gpu_bam != 0) // (the decode may go to the GPU: let the engine get its staging ring meanwhile)
			opts.flags |= CONGA_FLAG_EXPECT_BGZF;
		if (keep && keep->expect_cohort)
			opts.flags |= CONGA_FLAG_EXPECT_COHORT; // (a list of several BAMs: the pipeline's buffers while the first one is on)
	}
	opts.min_read_length = params->min_read_length;
	// the reference's split-read gate: `!no_sr && dup_file` (svdepth.c:57, bam_data.c:207,306,331, likelihood.c:344)
	const bool split_reads = !params->no_sr && params->have_dups;
	// (the caller passes buffered = true whenever the counts of the closing line are not known while reading)
	const bool counts_known_now = params->mq_threshold < 0 && !split_reads;
	// The engine context (HIP runtime, streams, the staging ring) is made on a thread of its own while this one reads the
	// BAM's block table: neither needs the other.
	int status = 0;
	conga_ctx *ctx = keep ? keep->ctx : nullptr;
	// a kept context whose chromosomes, intervals and tracks are the ones this sample needs: only the reads change
	// (with split reads too: the reference sequences, the satellites and the 10-mer indexes built from them are the layout's)
	bool same_layout = ctx != nullptr && !mine.empty() && keep->layout_key == layout_key_of(mine);
	if (ctx != nullptr && !same_layout)
		engine_check(ctx, conga_reset(ctx), "conga_reset");
	std::thread creator;
	if (!ctx)
		creator = std::thread([&] { ctx = conga_create(device, &opts, &status); });
	auto need_ctx = [&]() {
		if (!creator.joinable())
			return;
		creator.join();
		if (!ctx) {
			fprintf(stderr, "\n[CONGA ENGINE ERROR] cannot create a context on HIP device %d: %s\n", device, conga_strerror(status));
			exit(CONGA_EXIT_COMMON);
		}
		wt->ms_create = ms_since(t_create);
		if (keep)
			keep->early_ctx.store(ctx); // (the second sample's planning thread is waiting for it: read_bam_cohort)
	};
	const auto t_loop = now();

	std::string err;
	std::vector<uint64_t> gpu_counts; // reads per chromosome when all of this worker's chromosomes were decoded on the GPU at once
	const int gpu_bam = knobs().gpu_bam; // CONGA_GPU_BAM: 0 never, 1 always, -1 (unset): when it pays
	// readReferenceSeq (common.c:423-463) and the satellite annotation (bam_data.c:96-97,207) for the chromosome begun last
	// (all of this worker's chromosomes are read side by side the first time one is asked for: a genome's FASTA is 3 GB of text)
	std::vector<std::string> ref_seqs;
	auto hand_over_reference = [&](const chrom_job *job) {
		if (ref_seqs.empty()) {
			ref_seqs.resize(mine.size());
			std::vector<std::string> errs(mine.size());
			std::atomic<size_t> next{0};
			auto load_some = [&] {
				for (size_t k; (k = next.fetch_add(1)) < mine.size();)
					if (!load_fasta_chrom(params->ref_genome, this_sonic->chromosome_names[mine[k]->chr_index], mine[k]->L, &ref_seqs[k], &errs[k])
							&& errs[k].empty())
						errs[k] = "cannot read chromosome " + this_sonic->chromosome_names[mine[k]->chr_index] + " from " + params->ref_genome;
			};
			std::vector<std::thread> pool;
			for (int t = 1; t < std::min<int>((int) mine.size(), std::max(1, usable_cpus() / reader_share())); t++)
				pool.emplace_back(load_some);
			load_some();
			for (std::thread &t : pool)
				t.join();
			for (const std::string &e : errs)
				if (!e.empty())
					print_error(e);
		}
		size_t k = 0;
		while (k < mine.size() && mine[k] != job)
			k++;
		std::string ref_seq;
		ref_seq.swap(ref_seqs[k]);
		engine_check(ctx, conga_reference(ctx, ref_seq.data(), (int64_t) ref_seq.size()), "conga_reference");
		engine_check(ctx, conga_satellites(ctx, this_sonic->sat_start[job->chr_index].data(),
				this_sonic->sat_end[job->chr_index].data(), this_sonic->sat_start[job->chr_index].size()), "conga_satellites");
	};
	if (!mine.empty() && gpu_bam != 0) {
		// All chromosomes of this context in ONE decode on the GPU (conga_reads_bgzf): a low-coverage genome is tens of
		// thousands of BGZF blocks as a whole, not per chromosome.  The chromosomes are opened (with their intervals and
		// tracks) first, then the file's stretch goes up as it is.
		std::vector<device_target> targets;
		for (size_t i = 0; i < mine.size(); i++)
			targets.push_back(device_target{mine[i]->chr_index_bam, mine[i]->L, (int) i});
		file_piece own_bytes;
		std::vector<conga_bgzf_block> own_blocks;
		std::vector<conga_bam_segment> own_segments;
		const auto t_plan = now();
		// planned ahead (cohort) for exactly these chromosomes in this order? then that plan is the one
		bool use_pre = pre != nullptr && pre->targets.size() == targets.size();
		for (size_t i = 0; use_pre && i < targets.size(); i++)
			use_pre = pre->targets[i].tid == targets[i].tid && pre->targets[i].chrom_len == targets[i].chrom_len && pre->targets[i].chrom == targets[i].chrom;
		const bool planned = use_pre ? pre->planned
				: src->device_plan(targets, gpu_bam >= 0 ? 0 : gpu_bam_min_piece(), &own_bytes, &own_blocks, &own_segments, &err);
		file_piece &bytes = use_pre ? pre->bytes : own_bytes;
		std::vector<conga_bgzf_block> &blocks = use_pre ? pre->blocks : own_blocks;
		std::vector<conga_bam_segment> &segments = use_pre ? pre->segments : own_segments;
		const double ms_plan = use_pre ? pre->ms_plan : ms_since(t_plan);
		need_ctx();
		if (planned) {
			const auto t_open = now();
			if (same_layout)
				engine_check(ctx, conga_sample_begin(ctx), "conga_sample_begin"); // the layout stays, the reads go
			else
				for (chrom_job *job : mine) {
					std::vector<uint8_t> gc_hist_w, gc_like_w;
					gc_window_arrays(this_sonic, job->chr_index, &gc_hist_w, &gc_like_w);
					engine_check(ctx, conga_chrom_begin(ctx, job->L, gc_hist_w.data(), gc_like_w.data(), (int64_t) gc_hist_w.size()),
							"conga_chrom_begin");
					if (split_reads)
						hand_over_reference(job); // (the engine then keeps the records' places in the inflated stream)
					attach_intervals(ctx, params, map_bed, job->cs);
				}
			gpu_counts.assign(mine.size(), 0);
			if (knobs().timing)
				fprintf(stderr, "\n[timing] block table + start points %.1f ms, chromosomes opened (GC tracks, intervals, tracks) %.1f ms\n",
						ms_plan, ms_since(t_open));
			if (keep && keep->call_begins)
				keep->call_begins->store(true);
			const int rc = bytes.data ? conga_reads_bgzf(ctx, bytes.data, bytes.size, blocks.data(), blocks.size(), segments.data(), segments.size(),
					gpu_counts.data())
					: conga_reads_bgzf_fd(ctx, bytes.fd, bytes.file_off, bytes.size, blocks.data(), blocks.size(), segments.data(), segments.size(),
					gpu_counts.data());
			if (rc != CONGA_OK) {
				if (rc != CONGA_ERR_DATA)
					engine_check(ctx, rc, "conga_reads_bgzf");
				fprintf(stderr, "\n[CONGA] decoding on the host: %s\n", conga_last_error(ctx));
				gpu_counts.clear();
				engine_check(ctx, conga_reset(ctx), "conga_reset");
				same_layout = false;
			} else if (!keep || keep->last_sample) {
				// the last BAM is in: its 96 MB of pinned staging go back now, beside the compute, instead of being taken down
				// with the process (which costs the driver four times as long, tools/exitcost.hip)
				conga_ctx *c = ctx;
				releaser.t = std::thread([c] { (void) conga_release_staging(c); });
			}
		}
		err.clear();
	}
	need_ctx();
	whole_sample ws; // (handed over as it lies: stays until the compute below is through)
	// (only where the chromosome loop below would decode on the host: an indexed BAM whose whole stretch was refused above -- too large
	// for one call -- still goes up chromosome by chromosome, count_reads_bam)
	const bool host_decoders = gpu_bam == 0 || src->index_path().empty();
	const bool packed_done = same_layout && gpu_counts.empty() && keep && !split_reads && knobs().host_packed && host_decoders && !mine.empty()
			&& hand_over_packed(ctx, keep, src, params, mine, &ws);
	if (same_layout && gpu_counts.empty() && !packed_done)
		engine_check(ctx, conga_sample_begin(ctx), "conga_sample_begin"); // (host decoders: the staging ring, chromosome by chromosome)
	for (size_t job_index = 0; job_index < mine.size(); job_index++) {
		chrom_job *job = mine[job_index];
		const bool on_gpu = !gpu_counts.empty() || packed_done; // opened, equipped and filled above: only the progress text is left
		progress out = {buffered ? &job->messages : nullptr};
		job->counts_pending = !counts_known_now;
		if (!buffered && !job->messages.empty()) {
			fputs(job->messages.c_str(), stderr); // what the selection pass had to say before this chromosome
			job->messages.clear();
		}
		const int64_t L = job->L;
		out.say("\n");
		out.say("Reading BAM [%s] - Chromosome: %s", src->sample_name().c_str(), src->target_name(job->chr_index_bam).c_str());

		if (!on_gpu && same_layout)
			engine_check(ctx, conga_sample_chrom(ctx, (int) job_index), "conga_sample_chrom");
		else if (!on_gpu) {
			// init_rd_per_chr + the GC side of calc_mean_per_chr (read_distribution.c:12-18,63-73)
			std::vector<uint8_t> gc_hist_w, gc_like_w;
			gc_window_arrays(this_sonic, job->chr_index, &gc_hist_w, &gc_like_w);
			engine_check(ctx, conga_chrom_begin(ctx, L, gc_hist_w.data(), gc_like_w.data(), (int64_t) gc_hist_w.size()),
					"conga_chrom_begin");
		}
		if (split_reads) {
			out.say("\nReading the Reference Genome");
			if (!on_gpu && !same_layout)
				hand_over_reference(job);
		}
		out.say("\n-->counting reads");
		const int64_t cnt_reads = packed_done ? (int64_t) (ws.off[job_index + 1] - ws.off[job_index])
				: on_gpu ? (int64_t) gpu_counts[job_index]
				: count_reads_bam(ctx, src, job->chr_index_bam, L, split_reads, (int) job_index);
		if (counts_known_now)
			out.say(" (%lld reads, %ld split-reads)\n", (long long) cnt_reads, 0L); // every record counts, no split reads
		else
			out.say(" (%c reads, %c split-reads)\n", kMarkReads, kMarkSplit);

		// find_SVs, loading half (likelihood.c:319-336); the rows were picked by the selection pass
		chrom_svs &cs = job->cs;
		out.say("\nLoading known SVs");
		out.say("(%d DELS, %d DUPS in chromosome %s - larger than the threshold %d)\n", (int) cs.dels.size(), (int) cs.dups.size(),
				cs.chr_name.c_str(), params->min_sv_size);
		if (params->have_map && cs.dels.size() + cs.dups.size() > 0)
			out.say("Finding mappability for each region\n");
		if (!on_gpu && !same_layout)
			attach_intervals(ctx, params, map_bed, cs);
		job->staged = true;
	}
	wt->ms_reads = ms_since(t_loop);
	if (keep && keep->call_begins)
		keep->call_begins->store(true); // (a sample that went another way: its reads are in)

	// ---- calc_mean_per_chr + find_depths for every chromosome of this context at once
	if (!mine.empty()) {
		if (announce_compute)
			fprintf(stderr, "\nCalculating Likelihoods\n");
		const auto t_compute = now();
		engine_check(ctx, conga_chrom_compute(ctx), "conga_chrom_compute");
		engine_check(ctx, conga_sync(ctx), "conga_sync");
		host_trace("compute + sync returned");
		wt->ms_compute = ms_since(t_compute);
		const auto t_fetch = now();
		for (size_t i = 0; i < mine.size(); i++) {
			chrom_svs &cs = mine[i]->cs;
			cs.del_res.resize(cs.dels.size());
			cs.dup_res.resize(cs.dups.size());
			float expected_rd[101];
			engine_check(ctx, conga_chrom_select(ctx, (int) i), "conga_chrom_select");
			engine_check(ctx, conga_chrom_fetch(ctx, cs.del_res.data(), cs.dup_res.data(), expected_rd, &mine[i]->st), "conga_chrom_fetch");
		}
		wt->ms_fetch = ms_since(t_fetch);
	}
	if (!keep || keep->last_sample) {
		// no BAM behind this one: whatever staging the decode on the GPU still holds goes back (a no-op when the releaser above
		// has done it), and the helper thread its first call left inside the HIP runtime is joined before the process leaves
		if (releaser.t.joinable())
			releaser.t.join();
		(void) conga_release_staging(ctx);
	}
	// Everything is fetched.  The process is about to end, and giving gigabytes of device and pinned memory back one
	// allocation at a time is a quarter of a second the operating system does for nothing: the context is left to it
	// (CONGA_CLEAN_EXIT=1: tear down in order, for leak checkers).
	if (keep) {
		keep->ctx = ctx; // (the next sample's)
		keep->layout_key = layout_key_of(mine);
	} else if (knobs().clean_exit)
		conga_destroy(ctx);
}

int read_bam_with(parameters *params, sonic *this_sonic, kept_engine *keep, planned_input *pre = nullptr);
int cohort_pipeline(parameters *params, sonic *this_sonic, const std::vector<std::pair<std::string, std::string>> &samples);

} // namespace

int read_bam(parameters *params, sonic *this_sonic)
{
	return read_bam_with(params, this_sonic, nullptr);
}

int read_bam_cohort(parameters *params, sonic *this_sonic)
{
	// the list: one BAM per line, optionally a tab (or blanks) and the sample's output prefix; '#' starts a comment
	std::vector<std::pair<std::string, std::string>> samples;
	{
		FILE *f = fopen(params->cohort_file.c_str(), "r");
		if (!f)
			print_error("[CONGA INPUT ERROR] Unable to open file " + params->cohort_file + " in read mode.");
		char line[8192];
		while (fgets(line, sizeof line, f)) {
			char *save = nullptr;
			const char *bam = strtok_r(line, " \t\r\n", &save);
			if (!bam || bam[0] == '#')
				continue;
			const char *prefix = strtok_r(nullptr, " \t\r\n", &save);
			std::string out;
			if (prefix)
				out = prefix;
			else { // <--out>.<file name without directory and .bam>
				std::string stem = bam;
				const size_t slash = stem.rfind('/');
				if (slash != std::string::npos)
					stem.erase(0, slash + 1);
				if (stem.size() > 4 && stem.compare(stem.size() - 4, 4, ".bam") == 0)
					stem.erase(stem.size() - 4);
				out = params->outdir + params->outprefix + "." + stem;
			}
			samples.emplace_back(bam, out);
		}
		fclose(f);
	}
	if (samples.empty())
		print_error("[CONGA INPUT ERROR] " + params->cohort_file + " names no BAM file.");
	map_bam_pieces = samples.size() < 2; // (a mapping per sample would have to be given back between samples: reads.h)
	// --gpus N with a cohort: the SAMPLES are dealt to N pipelines, round robin -- the reference runs one process per sample
	// (bam_data.c:253-339; svdepth.c:47-66), so a cohort shards by sample with nothing to exchange, and end to end a sample costs
	// its upload: every GPU brings its own PCIe link.  Each pipeline is what `--gpus 1` runs -- one kept
	// engine context on its device, its layout handed over once, the next samples' bytes named two deep -- on its own host thread;
	// a sample's three files are those of its own run whatever N is.  (A single BAM with --gpus N shards by chromosome: read_bam.)
	const int n_pipes = (int) std::min<size_t>((size_t) std::max(1, params->n_gpus), samples.size());
	if (n_pipes <= 1) {
		parameters one = *params;
		one.n_gpus = samples.size() == 1 ? std::max(1, params->n_gpus) : 1; // (one BAM: its chromosomes are sharded, read_bam_with)
		return cohort_pipeline(&one, this_sonic, samples);
	}
	const int n_dev = std::max(1, conga_device_count());
	if (n_dev < n_pipes)
		fprintf(stderr, "\n[CONGA] --gpus %d with %d visible HIP device(s): pipelines share devices\n", n_pipes, n_dev);
	set_reader_share(n_pipes); // (host threads of every pipeline: its share of the cores)
	std::vector<std::vector<std::pair<std::string, std::string>>> dealt((size_t) n_pipes);
	for (size_t k = 0; k < samples.size(); k++)
		dealt[k % (size_t) n_pipes].push_back(samples[k]);
	std::vector<int> rcs((size_t) n_pipes, 0);
	std::vector<std::thread> pipes;
	for (int w = 0; w < n_pipes; w++)
		pipes.emplace_back([&, w] {
			parameters mine = *params;
			mine.n_gpus = 1;
			mine.device = (params->device + w) % n_dev;
			rcs[(size_t) w] = cohort_pipeline(&mine, this_sonic, dealt[(size_t) w]);
		});
	for (std::thread &t : pipes)
		t.join();
	for (int rc : rcs)
		if (rc != 0)
			return rc;
	return 0;
}

namespace {

// One pipeline of a cohort: the samples of `samples` one after the other through ONE kept engine context on params->device.
int cohort_pipeline(parameters *params, sonic *this_sonic, const std::vector<std::pair<std::string, std::string>> &samples)
{
	kept_engine keep;
	const std::string outdir = params->outdir, outprefix = params->outprefix;
	// The samples behind the one on the GPU are got ready meanwhile, two deep: their files are opened and -- once the engine
	// exists -- their bytes named to it from the index alone (conga_reads_bgzf_next_fd), so that the engine's upload thread
	// brings sample k + 1 and then k + 2 up back to back, reads their block tables off the bytes and inflates them ahead; a
	// sample's planning thread sleeps in conga_reads_bgzf_next_table until its bytes are up, then places the start points.
	const size_t n_samples = samples.size();
	std::vector<std::unique_ptr<planned_input>> plans(n_samples);
	std::vector<std::thread> planners(n_samples);
	std::unique_ptr<std::atomic<bool>[]> named(new std::atomic<bool>[n_samples]); // sample j's bytes are named (or will not be)
	for (size_t j = 0; j < n_samples; j++)
		named[j] = false;
	std::unique_ptr<std::atomic<uint64_t>[]> tickets(new std::atomic<uint64_t>[n_samples]); // conga_reads_bgzf_next_fd's, 0: not named
	for (size_t j = 0; j < n_samples; j++)
		tickets[j] = 0;
	std::unique_ptr<std::atomic<bool>[]> begun(new std::atomic<bool>[n_samples]); // sample j's run (and with it its call) has begun
	for (size_t j = 0; j < n_samples; j++)
		begun[j] = false;
	std::atomic<bool> first_sample_done{false};
	auto launch = [&](size_t j, conga_ctx *engine_now, size_t ahead_of, bool wait_for_engine = false) { // ahead_of: the sample right behind the one on the GPU
		if (j >= n_samples || planners[j].joinable() || plans[j])
			return;
		planners[j] = std::thread([&, j, engine_now, ahead_of, wait_for_engine] {
			// (bytes go up in the order they were named: sample j's not before sample j - 1's)
			conga_ctx *engine = engine_now;
			if (!engine && wait_for_engine) { // the first sample is on: its context appears a few hundred milliseconds into it
				for (int spin = 0; spin < 20000 && !(engine = keep.early_ctx.load()) && !first_sample_done.load(); spin++)
					std::this_thread::sleep_for(std::chrono::microseconds(200));
				if (!engine)
					engine = keep.early_ctx.load();
			}
			while (engine && j > 0 && !named[j - 1].load())
				std::this_thread::sleep_for(std::chrono::microseconds(200));
			// (... and only behind NAMED bytes, or behind the sample on the GPU: a sample whose bytes the engine was not told about
			// brings them with its call, and a call must not find two named stretches in front of its own)
			conga_ctx *tell = engine && (j <= ahead_of || tickets[j - 1].load() != 0) ? engine : nullptr;
			plans[j] = plan_input(params, this_sonic, samples[j].first, tell, &named[j], &tickets[j], j > 0 ? &begun[j - 1] : nullptr);
			named[j] = true;
		});
	};
	// How many samples behind the one on the GPU have their bytes named to the engine (CONGA_COHORT_AHEAD: measurement switch).
	// Two where the GPU has slack beside the link -- a 1x genome: 35-40 ms of upload against ~40 ms of inflate, walks and compute
	// that wait for it; 49 ms per sample against 56 with none named.  One with split reads, and the engine only brings its bytes
	// up (a context that holds reference text does not inflate ahead: the sample in front maps its split reads on the inflated
	// stream where it lies, a second one would be another 45 GB and share the machine with that stage): the link, which is what
	// a 5x genome with sequences waits for, never stands still -- 319 ms per sample of a cohort of twelve against 417 with none
	// named and 370-400 with the next sample inflated ahead as well (profiles/r03j_rp_cohort_ahead.log).
	const bool with_split_reads = !params->no_sr && params->have_dups;
	const int ahead_depth = knobs().cohort_ahead >= 0 ? knobs().cohort_ahead : with_split_reads ? 1 : 2;
	keep.expect_cohort = n_samples >= 3 && ahead_depth >= 1;
	plans[0] = plan_input(params, this_sonic, samples[0].first);
	named[0] = true;
	plan_beside_upload = n_samples > 1; // (from here on a plan runs beside a sample's upload: reads.h)
	std::thread cleaner; // gives the sample before's mapping back (3 GB of touched pages: ~75 ms of munmap) beside this sample's work
	const auto t_cohort = std::chrono::steady_clock::now();
	auto give_up_plans = [&](size_t from) { // (an error ends the run: nothing of ours may still be running, no named bytes left behind)
		for (size_t j = from; j < n_samples; j++) {
			if (planners[j].joinable())
				planners[j].join();
			conga_ctx *engine = keep.ctx ? keep.ctx : keep.early_ctx.load();
			if (engine && plans[j] && plans[j]->ahead_ticket)
				(void) conga_reads_bgzf_forget(engine, plans[j]->ahead_ticket);
		}
	};
	for (size_t k = 0; k < n_samples; k++) {
		if (planners[k].joinable())
			planners[k].join();
		std::unique_ptr<planned_input> mine_now = std::move(plans[k]);
		// (keep.ctx: made by the first sample's run, the same from then on; the second sample's planning thread waits for it to appear)
		launch(k + 1, ahead_depth >= 1 ? keep.ctx : nullptr, k + 1, ahead_depth >= 1 && k == 0 && params->n_gpus == 1);
		if (keep.ctx && ahead_depth >= 2)
			launch(k + 2, keep.ctx, k + 1);
		params->bam_file = samples[k].first;
		params->outdir.clear(); // (a prefix from the list is taken as it is; the default one already carries --out's directory)
		params->outprefix = samples[k].second;
		keep.last_sample = k + 1 == n_samples;
		fprintf(stderr, "\n[CONGA] sample %zu of %zu: %s\n", k + 1, n_samples, params->bam_file.c_str());
		host_trace("begins", k + 1);
		keep.call_begins = &begun[k]; // (raised by run_worker right in front of the sample's call)
		// several contexts (--gpus N) are made per sample; one context is kept from sample to sample
		const int rc = read_bam_with(params, this_sonic, params->n_gpus == 1 ? &keep : nullptr, mine_now.get());
		first_sample_done = true;
		host_trace("files written", k + 1);
		const auto t_join = std::chrono::steady_clock::now();
		if (k + 1 < n_samples && planners[k + 1].joinable())
			planners[k + 1].join();
		if (knobs().timing && k + 1 < n_samples)
			fprintf(stderr, "[timing] waited %.1f ms more for the next sample's plan (file opened, bytes up, block table, start points)\n",
					std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_join).count());
		if (cleaner.joinable())
			cleaner.join();
		if (knobs().timing) // (what a further sample costs, read off one process's own clock: bench.py's end-to-end legs)
			fprintf(stderr, "[timing] cohort: sample %zu of %zu is done %.1f ms after the first one began\n", k + 1, n_samples,
					std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cohort).count());
		// (bytes the engine was told about and never asked for -- this sample was decoded on the host after all, or failed --
		// are given up before their descriptor goes; for bytes taken up this does nothing)
		if (keep.ctx && mine_now && mine_now->ahead_ticket)
			(void) conga_reads_bgzf_forget(keep.ctx, mine_now->ahead_ticket);
		if (rc != 0) {
			give_up_plans(k + 1);
			if (cleaner.joinable())
				cleaner.join();
			return rc;
		}
		if (k + 1 < n_samples) {
			planned_input *done = mine_now.release();
			cleaner = std::thread([done] { delete done; });
		} else if (!knobs().clean_exit)
			(void) mine_now.release(); // the last one's mapping goes with the process
	}
	if (cleaner.joinable())
		cleaner.join();
	params->outdir = outdir;
	params->outprefix = outprefix;
	if (keep.packer)
		conga_packer_destroy(keep.packer); // (its threads)
	if (keep.ctx && knobs().clean_exit) {
		conga_host_free(keep.ctx, keep.packed_pin);
		conga_host_free(keep.ctx, keep.mapq_pin);
		conga_destroy(keep.ctx);
	}
	return 0;
}

int read_bam_with(parameters *params, sonic *this_sonic, kept_engine *keep, planned_input *pre)
{
	FILE *fpDel = nullptr, *fpDup = nullptr, *fpSVs = nullptr;

	// ---- output files and headers (bam_data.c:232-250)
	const std::string svfile = params->outdir + params->outprefix + "_svs.bed";
	fprintf(stderr, "\nOutput SV file: %s\n", svfile.c_str());
	fpSVs = safe_fopen(svfile, "w");
	fprintf(fpSVs, "#CHR\tSTART_SV\tEND_SV\tSV_TYPE\tCOPY_NUMBER\tLIKELIHOOD\tREAD_PAIR\tMAPPABILITY\n");
	if (params->have_dels) {
		const std::string f = params->outdir + params->outprefix + "_dels.bed";
		fprintf(stderr, "Output Del file: %s\n", f.c_str());
		fpDel = safe_fopen(f, "w");
		fprintf(fpDel, "#CHR\tSTART_SV\tEND_SV\tCOPY_NUMBER\tLIKELIHOOD\tREAD_PAIR\tMAPPABILITY\tOBSERVED_READS\tEXPECTED_READS\n");
	}
	if (params->have_dups) {
		const std::string f = params->outdir + params->outprefix + "_dups.bed";
		fprintf(stderr, "Output DUP file: %s\n", f.c_str());
		fpDup = safe_fopen(f, "w");
		fprintf(fpDup, "#CHR\tSTART_SV\tEND_SV\tCOPY_NUMBER\tLIKELIHOOD\tREAD_PAIR\tMAPPABILITY\tOBSERVED_READS\tEXPECTED_READS\n");
	}

	// CONGA_TIMING=1: wall time of the host phases on stderr (where an end-to-end run spends its time)
	const bool timing = knobs().timing;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms_since = [&](std::chrono::steady_clock::time_point t) {
		return std::chrono::duration<double, std::milli>(now() - t).count();
	};
	const auto t_start = now();

	// ---- inputs (bam_data.c:253-267); the BED files are parsed once instead of once per chromosome
	std::string err;
	std::unique_ptr<read_source> src(pre && pre->src ? pre->src.release() : open_reads(params->bam_file, &err));
	if (!src)
		print_error(err);
	// (a cohort's samples share the three files: parsed for the first one, kept with the engine -- a mappability track is millions of rows)
	bed_index own_dels, own_dups, own_map;
	bed_index &dels_bed = keep ? keep->dels_bed : own_dels, &dups_bed = keep ? keep->dups_bed : own_dups, &map_bed = keep ? keep->map_bed : own_map;
	if (!keep || !keep->beds_loaded) {
		if (params->have_dels && !load_bed(params->del_file, false, &dels_bed))
			print_error("[CONGA INPUT ERROR] Unable to open file " + params->del_file + " in read mode.");
		if (params->have_dups && !load_bed(params->dup_file, false, &dups_bed))
			print_error("[CONGA INPUT ERROR] Unable to open file " + params->dup_file + " in read mode.");
		if (params->have_map && !load_bed(params->mappability_file, true, &map_bed))
			print_error("[CONGA INPUT ERROR] Unable to open file " + params->mappability_file + " in read mode.");
		if (keep)
			keep->beds_loaded = true;
	}
	const double ms_inputs = ms_since(t_start);

	// ---- chromosome selection (bam_data.c:269-291) and the known SVs of each (likelihood.c:319-331)
	std::vector<chrom_job> jobs;
	std::string pending; // messages of skipped chromosomes: they belong in front of the next one that runs
	for (int chr_index = 0; chr_index < this_sonic->number_of_chromosomes; chr_index++) {
		if (chr_index < params->first_chrom)
			chr_index = params->first_chrom;
		if (chr_index > params->last_chrom || chr_index >= this_sonic->number_of_chromosomes)
			break;
		const std::string &name = this_sonic->chromosome_names[chr_index];
		if (name.find('X') != std::string::npos || name.find('Y') != std::string::npos)
			continue; // strstr(name, "X") / "Y" (bam_data.c:280)
		const int chr_index_bam = find_chr_index_bam(name, *src);
		if (chr_index_bam == -1) {
			pending += "\nCannot find chromosome name " + name + " in BAM/CRAM " + src->sample_name();
			continue;
		}
		chrom_job job;
		job.chr_index = chr_index;
		job.chr_index_bam = chr_index_bam;
		job.L = this_sonic->chromosome_lengths[chr_index];
		job.messages.swap(pending);
		// BED rows are matched against the BAM's target name
		job.cs.chr_name = src->target_name(chr_index_bam);
		if (params->have_dels)
			job.cs.dels = known_SVs_for(dels_bed, job.cs.chr_name, params->min_sv_size);
		if (params->have_dups)
			job.cs.dups = known_SVs_for(dups_bed, job.cs.chr_name, params->min_sv_size);
		jobs.push_back(std::move(job));
	}

	// ---- chromosomes -> contexts: longest-processing-time-first on L + 2 * sum(interval length), the cost model of
	// conga_amd/shard.py.  Worker k drives HIP device (--device + k) modulo the visible devices.
	int n_workers = std::max(1, std::min(params->n_gpus, (int) std::max<size_t>(jobs.size(), 1)));
	// (with one worker nothing here needs the HIP runtime yet: it is still coming up on its own thread, main.cpp)
	const int n_dev = n_workers > 1 ? conga_device_count() : 1;
	if (n_workers > 1 && n_dev > 0 && n_dev < n_workers)
		fprintf(stderr, "\n[CONGA] --gpus %d with %d visible HIP device(s): contexts share devices\n", n_workers, n_dev);
	std::vector<std::vector<chrom_job *>> mine((size_t) n_workers);
	{
		std::vector<size_t> order(jobs.size());
		std::vector<double> cost(jobs.size());
		for (size_t i = 0; i < jobs.size(); i++) {
			order[i] = i;
			double sum_len = 0;
			for (const sv_row &r : jobs[i].cs.dels)
				sum_len += (double) r.end - (double) r.start;
			for (const sv_row &r : jobs[i].cs.dups)
				sum_len += (double) r.end - (double) r.start;
			cost[i] = (double) jobs[i].L + 2.0 * sum_len;
		}
		std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
		std::vector<double> load((size_t) n_workers, 0.0);
		for (size_t i : order) {
			const size_t w = (size_t) (std::min_element(load.begin(), load.end()) - load.begin());
			jobs[i].worker = (int) w;
			load[w] += cost[i];
		}
		for (chrom_job &j : jobs)
			mine[(size_t) j.worker].push_back(&j); // annotation order within a worker
	}

	const double ms_select = ms_since(t_start) - ms_inputs;
	const auto t_work = now();
	std::vector<worker_timing> wt((size_t) n_workers);
	const bool hold_lines = params->mq_threshold >= 0 || (!params->no_sr && params->have_dups);
	auto print_held_lines = [&]() {
		long split_total = 0; // split_read_count is never reset between chromosomes (split_read.c:14)
		for (chrom_job &j : jobs) {
			split_total += (long) j.st.split_elements;
			if (j.counts_pending)
				fill_counts(&j.messages, (long long) j.st.reads_counted, split_total);
			fputs(j.messages.c_str(), stderr);
		}
		if (!jobs.empty())
			fprintf(stderr, "\nCalculating Likelihoods\n");
	};
	if (n_workers == 1) {
		run_worker(params, this_sonic, src.get(), params->device, hold_lines, map_bed, mine[0], !hold_lines, &wt[0], keep, pre);
		if (hold_lines)
			print_held_lines();
	} else {
		set_reader_share(n_workers); // each worker's BAM reader gets its share of the inflate threads
		std::vector<std::thread> threads;
		for (int w = 0; w < n_workers; w++) {
			threads.emplace_back([&, w] {
				std::string werr;
				std::unique_ptr<read_source> wsrc(open_reads(params->bam_file, &werr));
				if (!wsrc)
					print_error(werr);
				const int device = n_dev > 0 ? (params->device + w) % n_dev : params->device + w;
				run_worker(params, this_sonic, wsrc.get(), device, true, map_bed, mine[(size_t) w], false, &wt[(size_t) w]);
			});
		}
		for (std::thread &t : threads)
			t.join();
		print_held_lines();
	}
	if (!pending.empty())
		fputs(pending.c_str(), stderr);
	const double ms_work = ms_since(t_work);

	// ---- output in annotation order (rank 0's job in SURVEY.md section 8e)
	const auto t_out = now();
	const bool split_reads = !params->no_sr && params->have_dups;
	// The rows of a chromosome are formatted into memory, all chromosomes side by side (40 000 rows through fprintf are
	// 10 ms on one thread), and then written in annotation order: the files are what one pass in order would have written.
	struct formatted { // the three files' rows and the progress line of the chromosome
		char *text[4] = {nullptr, nullptr, nullptr, nullptr};
		size_t size[4] = {0, 0, 0, 0};
	};
	std::vector<formatted> done(jobs.size());
	{
		std::atomic<size_t> next_job{0};
		auto format_some = [&] {
			for (;;) {
				const size_t k = next_job.fetch_add(1);
				if (k >= jobs.size())
					return;
				const chrom_svs &cs = jobs[k].cs;
				if (cs.dels.size() + cs.dups.size() == 0)
					continue; // find_SVs returns before output_SVs when the chromosome has no SV (likelihood.c:332-336)
				FILE *m[4];
				for (int f = 0; f < 4; f++)
					m[f] = open_memstream(&done[k].text[f], &done[k].size[f]);
				if (!m[0] || !m[1] || !m[2] || !m[3])
					print_error("out of memory while formatting the output");
				output_SVs(params, cs, m[0], fpDel ? m[1] : nullptr, fpDup ? m[2] : nullptr, m[3]);
				for (int f = 0; f < 4; f++)
					fclose(m[f]);
			}
		};
		const int n_fmt = (int) std::min<size_t>(jobs.size(), (size_t) std::max(1, std::min(16, usable_cpus())));
		std::vector<std::thread> fmt;
		for (int t = 1; t < n_fmt; t++)
			fmt.emplace_back(format_some);
		format_some();
		for (std::thread &t : fmt)
			t.join();
	}
	for (size_t k = 0; k < jobs.size(); k++) {
		const chrom_job &job = jobs[k];
		// calc_mu_per_chr's log line (read_distribution.c:41)
		fprintf(logFile, "Read Count:%li  Window count:%li mean=%f\n", (long) job.st.rd_sum, (long) job.L, job.st.mean);
		if (split_reads)
			fprintf(stderr, "\nCONGA paired %lld single-end reads\n", (long long) (job.st.split_del_rows + job.st.split_dup_rows));
		FILE *out[4] = {fpSVs, fpDel, fpDup, stderr};
		for (int f = 0; f < 4; f++) {
			if (out[f] && done[k].size[f])
				fwrite(done[k].text[f], 1, done[k].size[f], out[f]);
			free(done[k].text[f]);
		}
	}
	const double ms_output = ms_since(t_out);
	if (timing) {
		if (n_workers == 1)
			fprintf(stderr, "\n[timing] open + BED parsing %.1f ms, engine create %.1f ms, chromosome loop (annotation, read decode + "
					"staging, intervals) %.1f ms, layout + compute %.1f ms, fetch + output %.1f ms, total %.1f ms (selection %.1f ms, worker "
					"%.1f ms)\n", ms_inputs, wt[0].ms_create, wt[0].ms_reads, wt[0].ms_compute, wt[0].ms_fetch + ms_output, ms_since(t_start),
					ms_select, ms_work);
		else {
			fprintf(stderr, "\n[timing] open + BED parsing %.1f ms, %d workers %.1f ms, output %.1f ms, total %.1f ms\n", ms_inputs,
					n_workers, ms_work, ms_output, ms_since(t_start));
			for (int w = 0; w < n_workers; w++)
				fprintf(stderr, "[timing] worker %d: %zu chromosomes, engine create %.1f ms, chromosome loop %.1f ms, layout + compute "
						"%.1f ms, fetch %.1f ms\n", w, mine[(size_t) w].size(), wt[(size_t) w].ms_create, wt[(size_t) w].ms_reads,
						wt[(size_t) w].ms_compute, wt[(size_t) w].ms_fetch);
		}
	}

	fprintf(stderr, "\n");
	if (fpDel)
		fclose(fpDel);
	if (fpDup)
		fclose(fpDup);
	fclose(fpSVs);
	return 0;
}

} // namespace

} // namespace conga_host
