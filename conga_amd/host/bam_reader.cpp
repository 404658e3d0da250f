// bam_reader.cpp -- minimal coordinate-sorted BAM source (BGZF + BAM records, optional .bai seek).
//
// htslib is an un-vendored submodule of the reference and absent from this image, so the BAM side of
// count_reads_bam (bam_data.c:192-221) is re-implemented from the SAM/BAM specification: only the fixed
// 32-byte record core is decoded (refID, pos, mapq, flag, l_seq) -- the depth path reads nothing else.
// Record set (parity unpinned): all records with refID == tid and 0 <= pos < L, in
// file order, which is what sam_itr_queryi(idx, tid, 0, L) + sam_itr_next yield (bam_data.c:293,201).
// CRAM is not supported.  Inflate is zlib on a pool of worker threads with a 96-block read-ahead
// (CONGA_BAM_THREADS overrides the worker count; 0 = inflate inline).
#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "inflate_fast.h"
#include "knobs.h"
#include "reads.h"

namespace conga_host {

namespace {

// BGZF reader with read-ahead: the calling thread reads raw blocks from the file (cheap, page cache) and a small
// pool of workers inflates them; blocks are consumed strictly in file order.
class bgzf_reader {
public:
	// workers < 0: a pool sized for the host (the sequential reader); 0: inflate on the calling thread (what each of the
	// parallel segment readers of bam_file::read_all does)
	explicit bgzf_reader(int workers = -1)
	{
		int n = std::max(2, usable_cpus() / reader_share());
		if (knobs().bam_threads >= 0)
			n = knobs().bam_threads + 1;
		n_workers_ = std::max(0, std::min(n - 1, 24)); // one record-walking thread keeps up with ~24 inflating ones
		if (workers >= 0)
			n_workers_ = workers;
		for (int i = 0; i < n_workers_; i++)
			workers_.emplace_back([this] { work(); });
	}
	bgzf_reader(const bgzf_reader &) = delete;
	bgzf_reader &operator=(const bgzf_reader &) = delete;
	~bgzf_reader()
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			quit_ = true;
		}
		cv_work_.notify_all();
		for (auto &t : workers_)
			t.join();
		if (f_)
			fclose(f_);
	}
	bool open(const std::string &path)
	{
		drain();
		if (f_)
			fclose(f_);
		f_ = fopen(path.c_str(), "rb");
		reset_state();
		return f_ != nullptr;
	}
	// virtual offset = compressed block offset << 16 | offset inside the inflated block
	bool seek(uint64_t voffset)
	{
		drain();
		if (fseeko(f_, (off_t) (voffset >> 16), SEEK_SET) != 0)
			return false;
		reset_state();
		if (!load_block())
			return false;
		at_ = (size_t) (voffset & 0xFFFF);
		return at_ <= cur_size();
	}
	// read exactly n bytes; false on EOF / error
	bool read(void *dst, size_t n)
	{
		uint8_t *d = (uint8_t *) dst;
		while (n) {
			if (at_ == cur_size()) {
				if (!load_block() || cur_size() == 0)
					return false;
			}
			const size_t k = std::min(n, cur_size() - at_);
			memcpy(d, cur_->out.data() + at_, k);
			at_ += k;
			d += k;
			n -= k;
		}
		return true;
	}
	bool skip(size_t n)
	{
		while (n) {
			if (at_ == cur_size()) {
				if (!load_block() || cur_size() == 0)
					return false;
			}
			const size_t k = std::min(n, cur_size() - at_);
			at_ += k;
			n -= k;
		}
		return true;
	}
	bool at_eof()
	{
		while (at_ == cur_size()) {
			if (file_eof_ && queue_.empty())
				return true;
			if (!load_block())
				return true;
			if (cur_size() == 0 && file_eof_ && queue_.empty())
				return true;
		}
		return false;
	}
	// virtual offset of the next byte (start of the next block when the current one is used up); kNoOffset at the end
	static constexpr uint64_t kNoOffset = ~0ull;
	uint64_t tell()
	{
		if (at_eof())
			return kNoOffset;
		return ((uint64_t) cur_->coff << 16) | (uint64_t) at_;
	}
	const std::string &error() const { return err_; }

private:
	struct task {
		std::vector<uint8_t> cdata, out;
		uint64_t coff = 0; // file offset of the block
		uint32_t isize = 0, crc = 0;
		bool done = false, failed = false;
	};
	static constexpr size_t kAhead = 96;

	size_t cur_size() const { return cur_ ? cur_->out.size() : 0; }

	void reset_state()
	{
		cur_.reset();
		at_ = 0;
		file_eof_ = false;
		err_.clear();
	}

	// wait for every dispatched block and forget them (before seek / reopen)
	void drain()
	{
		std::unique_lock<std::mutex> lk(mu_);
		for (auto &t : queue_)
			cv_done_.wait(lk, [&] { return t->done; });
		queue_.clear();
		todo_.clear();
	}

	static bool inflate_block(task &t)
	{
		t.out.resize(t.isize);
		// the block decoder of inflate_fast.cpp first (CONGA_ZLIB_INFLATE=1: zlib only); whatever it produces has to pass
		// the block's CRC32, and a block it refuses or gets wrong goes through zlib before anything is reported
		const bool zlib_only = knobs().zlib_inflate;
		if (!zlib_only && inflate_raw(t.cdata.data(), t.cdata.size(), t.out.data(), t.out.size())
				&& (uint32_t) crc32(crc32(0L, Z_NULL, 0), t.out.data(), (uInt) t.out.size()) == t.crc)
			return true;
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		if (inflateInit2(&zs, -15) != Z_OK)
			return false;
		zs.next_in = t.cdata.data();
		zs.avail_in = (uInt) t.cdata.size();
		zs.next_out = t.out.data();
		zs.avail_out = t.isize;
		const int rc = inflate(&zs, Z_FINISH);
		inflateEnd(&zs);
		if (rc != Z_STREAM_END || zs.avail_out != 0)
			return false;
		// the gzip trailer's CRC32 of the inflated bytes (htslib's bgzf layer refuses a block that fails it too)
		return (uint32_t) crc32(crc32(0L, Z_NULL, 0), t.out.data(), (uInt) t.out.size()) == t.crc;
	}

	void work()
	{
		for (;;) {
			std::shared_ptr<task> t;
			{
				std::unique_lock<std::mutex> lk(mu_);
				cv_work_.wait(lk, [&] { return quit_ || !todo_.empty(); });
				if (quit_)
					return;
				t = todo_.front();
				todo_.pop_front();
			}
			const bool ok = inflate_block(*t);
			{
				std::lock_guard<std::mutex> g(mu_);
				t->failed = !ok;
				t->done = true;
			}
			cv_done_.notify_all();
		}
	}

	// read one raw block from the file into a task; false on error; sets file_eof_ at the end
	bool read_raw(std::shared_ptr<task> *out)
	{
		for (;;) { // skip empty blocks (the EOF marker is one)
			uint8_t h[12];
			const off_t block_at = ftello(f_);
			const size_t got = fread(h, 1, 12, f_);
			if (got == 0) {
				file_eof_ = true;
				out->reset();
				return true;
			}
			if (got != 12 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
				err_ = "not a BGZF block";
				return false;
			}
			const unsigned xlen = h[10] | (h[11] << 8);
			std::vector<uint8_t> extra(xlen);
			if (fread(extra.data(), 1, xlen, f_) != xlen) {
				err_ = "truncated BGZF header";
				return false;
			}
			int bsize = -1;
			for (unsigned i = 0; i + 4 <= xlen;) {
				const unsigned slen = extra[i + 2] | (extra[i + 3] << 8);
				if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen)
					bsize = extra[i + 4] | (extra[i + 5] << 8);
				i += 4 + slen;
			}
			if (bsize < 0) {
				err_ = "BGZF block without BC field";
				return false;
			}
			const int cdata = bsize - (int) xlen - 19; // total block = bsize + 1; header 12 + xlen; trailer 8
			if (cdata < 0) {
				err_ = "bad BGZF block size";
				return false;
			}
			auto t = std::make_shared<task>();
			t->coff = (uint64_t) block_at;
			t->cdata.resize((size_t) cdata + 8);
			if (fread(t->cdata.data(), 1, t->cdata.size(), f_) != t->cdata.size()) {
				err_ = "truncated BGZF block";
				return false;
			}
			memcpy(&t->crc, t->cdata.data() + cdata, 4);
			memcpy(&t->isize, t->cdata.data() + cdata + 4, 4);
			t->cdata.resize((size_t) cdata);
			if (t->isize > 65536u) { // the BGZF limit: anything larger is a damaged (or hostile) trailer, not an allocation size
				err_ = "bad BGZF block (ISIZE above 64 KiB)";
				return false;
			}
			if (t->isize == 0)
				continue;
			*out = t;
			return true;
		}
	}

	// make the next block current (empty current block at end of file)
	bool load_block()
	{
		at_ = 0;
		// keep the read-ahead queue full
		const size_t ahead = n_workers_ ? kAhead : 1; // (inflating on the calling thread: nothing to run ahead with)
		while (!file_eof_ && queue_.size() < ahead) {
			std::shared_ptr<task> t;
			if (!read_raw(&t))
				return false;
			if (!t)
				break;
			if (n_workers_ == 0) {
				t->failed = !inflate_block(*t);
				t->done = true;
				queue_.push_back(t);
			} else {
				{
					std::lock_guard<std::mutex> g(mu_);
					queue_.push_back(t);
					todo_.push_back(t);
				}
				cv_work_.notify_one();
			}
		}
		if (queue_.empty()) {
			cur_.reset();
			return true;
		}
		std::shared_ptr<task> t;
		{
			std::unique_lock<std::mutex> lk(mu_);
			t = queue_.front();
			cv_done_.wait(lk, [&] { return t->done; });
			queue_.pop_front();
		}
		if (t->failed) {
			err_ = "BGZF block does not inflate to its recorded size and CRC32";
			return false;
		}
		cur_ = t;
		return true;
	}

	FILE *f_ = nullptr;
	std::shared_ptr<task> cur_;
	size_t at_ = 0;
	bool file_eof_ = false;
	std::string err_;
	int n_workers_ = 0;
	std::vector<std::thread> workers_;
	std::mutex mu_;
	std::condition_variable cv_work_, cv_done_;
	std::deque<std::shared_ptr<task>> queue_, todo_; // queue_: in file order (consumer); todo_: not yet inflated
	bool quit_ = false;
};

class bam_file : public read_source {
public:
	bool open(const std::string &path, std::string *err)
	{
		path_ = path;
		if (!bgzf_.open(path) || !read_header(err)) {
			if (err->empty())
				*err = "[CONGA INPUT ERROR] Unable to open file " + path + " in read mode.";
			return false;
		}
		// sample.bam.bai (samtools) or sample.bai (Picard); without either the file is read front to back
		if (!load_bai(path + ".bai") && path.size() > 4 && path.compare(path.size() - 4, 4, ".bam") == 0)
			load_bai(path.substr(0, path.size() - 4) + ".bai");
		return true;
	}
	int n_targets() const override { return (int) names_.size(); }
	const std::string &target_name(int tid) const override { return names_[tid]; }
	const std::string &sample_name() const override { return sample_; }
	std::string index_path() const override { return ref_beg_.empty() ? "" : bai_path_; }

	bool begin(int tid, int64_t chrom_len, std::string *err) override
	{
		tid_ = tid;
		len_ = chrom_len;
		done_ = false;
		if (tid < (int) ref_beg_.size() && ref_beg_[tid] != 0) {
			if (!bgzf_.seek(ref_beg_[tid])) {
				*err = "BAM seek failed";
				return false;
			}
			have_pending_ = false;
			last_ref_ = -2;
		} else if (last_ref_ > tid || last_ref_ == -1) {
			// no index and the file position is already past this target: start over
			have_pending_ = false;
			if (!bgzf_.open(path_) || !read_header(err))
				return false;
		}
		// otherwise keep scanning forward; a record already read for a later target stays pending
		return true;
	}

	bool next(size_t max_n, read_batch *out, std::string *err) override
	{
		pos_.clear();
		mapq_.clear();
		while (!done_ && pos_.size() < max_n) {
			core c;
			if (have_pending_) {
				c = pending_;
				have_pending_ = false;
			} else if (!read_core(&c, err)) {
				if (!err->empty())
					return false;
				done_ = true; // end of file
				break;
			}
			last_ref_ = c.ref_id;
			if (c.ref_id >= 0 && c.ref_id < tid_)
				continue; // earlier target (sequential scan without an index)
			if (c.ref_id != tid_) {
				// sorted file: a later target (or the unplaced tail, refID -1) ends this one
				pending_ = c;
				have_pending_ = true;
				done_ = true;
				break;
			}
			if (c.pos < 0)
				continue;
			if (c.pos >= len_) {
				done_ = true;
				break;
			}
			pos_.push_back(c.pos);
			mapq_.push_back(c.mapq);
		}
		out->pos = pos_.data();
		out->mapq = mapq_.data();
		out->n = pos_.size();
		return true;
	}

	// Parallel decode of one target.  The linear index gives, for every 16 kb window, the virtual offset of the first
	// record that overlaps it; records are sorted by position, so the records that START in [w_a, w_b) * 16 kb are a
	// contiguous run somewhere behind offset[w_a].  The target's stretch of the file is cut into segments of about equal
	// compressed size at window boundaries; each segment gets its own reader (own file handle, inflating on its own
	// thread), skips the records in front of its first window and stops at the first record of the next segment.  The
	// offsets of an index are not trusted blindly: segment k must stop exactly where segment k + 1 found its first
	// record (virtual offsets compared), otherwise the target is read sequentially.
	bool read_all(int tid, int64_t chrom_len, int threads, std::vector<int32_t> *pos, std::vector<uint8_t> *mapq,
			std::string *err) override
	{
		if (tid < 0 || tid >= (int) linear_.size() || threads < 2 || ref_beg_[(size_t) tid] == 0)
			return false;
		std::vector<uint64_t> lin = linear_[(size_t) tid];
		const size_t n_win = std::min(lin.size(), (size_t) ((chrom_len + 16383) >> 14));
		lin.resize(n_win);
		for (size_t w = 1; w < n_win; w++) // windows nothing overlaps carry 0 (or, from htslib, the previous value)
			if (lin[w] == 0 || lin[w] < lin[w - 1])
				lin[w] = lin[w - 1];
		size_t w_first = 0;
		while (w_first < n_win && lin[w_first] == 0)
			w_first++;
		if (w_first >= n_win)
			return false;
		const uint64_t c_lo = lin[w_first] >> 16, c_hi = lin[n_win - 1] >> 16;
		int want = threads;
		if (knobs().bam_segments >= 0)
			want = knobs().bam_segments; // (tests: any number of segments on a small file; 0 / 1: off)
		else
			want = (int) std::min<uint64_t>((uint64_t) std::min(threads, 128), (c_hi - c_lo) / (4u << 20)); // >= 4 MiB of file per segment
		if (want < 2)
			return false;
		struct segment {
			int64_t lo, hi;   // positions [lo, hi)
			uint64_t at;      // where to start reading
			uint64_t v_first = bgzf_reader::kNoOffset, v_stop = bgzf_reader::kNoOffset;
			std::vector<int32_t> pos;
			std::vector<uint8_t> mapq;
			std::string err;
		};
		std::vector<segment> segs;
		{
			std::vector<size_t> cuts = {0}; // windows at which a segment starts (the first one covers everything in front too)
			for (int k = 1; k < want; k++) {
				const uint64_t target = c_lo + (c_hi - c_lo) * (uint64_t) k / (uint64_t) want;
				size_t lo = w_first, hi = n_win; // first window whose offset lies at or behind `target`
				while (lo < hi) {
					const size_t mid = (lo + hi) / 2;
					if ((lin[mid] >> 16) < target)
						lo = mid + 1;
					else
						hi = mid;
				}
				if (lo < n_win && lo > cuts.back())
					cuts.push_back(lo);
			}
			for (size_t k = 0; k < cuts.size(); k++) {
				segment sg;
				sg.lo = k == 0 ? 0 : (int64_t) cuts[k] << 14;
				sg.hi = k + 1 < cuts.size() ? (int64_t) cuts[k + 1] << 14 : chrom_len;
				sg.at = k == 0 ? ref_beg_[(size_t) tid] : lin[cuts[k]];
				segs.push_back(std::move(sg));
			}
		}
		if (segs.size() < 2)
			return false;
		auto run = [&](segment &sg) {
			bgzf_reader bz(0);
			if (!bz.open(path_) || !bz.seek(sg.at)) {
				sg.err = "BAM seek failed";
				return;
			}
			for (;;) {
				const uint64_t v = bz.tell();
				if (v == bgzf_reader::kNoOffset) {
					if (!bz.error().empty())
						sg.err = bz.error();
					break; // end of file
				}
				int32_t block_size;
				uint8_t b[32];
				if (!bz.read(&block_size, 4) || block_size < 32 || !bz.read(b, 32)) {
					sg.err = bz.error().empty() ? "truncated BAM record" : bz.error();
					break;
				}
				int32_t ref_id, p;
				memcpy(&ref_id, b, 4);
				memcpy(&p, b + 4, 4);
				if (!(ref_id >= 0 && ref_id < tid) && (ref_id != tid || p >= sg.hi)) {
					// the record that ends the segment: only its first fields count (as for the walk on the GPU, kernels_bam.hip.h:
					// what lies behind them -- another target's record, the unplaced tail -- is not this target's to judge)
					sg.v_stop = v;
					if (sg.v_first == bgzf_reader::kNoOffset)
						sg.v_first = v;
					break;
				}
				if (!bz.skip((size_t) block_size - 32)) {
					sg.err = bz.error().empty() ? "truncated BAM record" : bz.error();
					break;
				}
				if (ref_id >= 0 && ref_id < tid)
					continue; // (a chunk may begin with the tail of the previous target)
				if (p < sg.lo)
					continue; // starts in front of this segment: the previous one's
				if (sg.v_first == bgzf_reader::kNoOffset)
					sg.v_first = v;
				if (p < 0)
					continue;
				sg.pos.push_back(p);
				sg.mapq.push_back(b[9]);
			}
		};
		{
			std::vector<std::thread> pool;
			for (size_t k = 1; k < segs.size(); k++)
				pool.emplace_back([&, k] { run(segs[k]); });
			run(segs[0]);
			for (std::thread &t : pool)
				t.join();
		}
		for (const segment &sg : segs)
			if (!sg.err.empty()) {
				*err = sg.err;
				return false;
			}
		for (size_t k = 0; k + 1 < segs.size(); k++)
			if (segs[k].v_stop != segs[k + 1].v_first) {
				fprintf(stderr, "\n[CONGA] %s: the linear index of target %d does not line up with the records "
						"(segment %zu); reading it sequentially\n", bai_path_.c_str(), tid, k);
				return false;
			}
		size_t total = 0;
		for (const segment &sg : segs)
			total += sg.pos.size();
		pos->clear();
		mapq->clear();
		pos->reserve(total);
		mapq->reserve(total);
		for (const segment &sg : segs) {
			pos->insert(pos->end(), sg.pos.begin(), sg.pos.end());
			mapq->insert(mapq->end(), sg.mapq.begin(), sg.mapq.end());
		}
		// the sequential iterator would now stand behind this target
		have_pending_ = false;
		last_ref_ = -2;
		done_ = true;
		return true;
	}

	// The targets' stretch of the file for conga_reads_bgzf: from the block of the first record of the first of them in
	// the file through the block in which the next target with records behind the last of them begins (so that the walk
	// sees a record that ends it; to the end of the file when there is none), the table of those blocks, and for every
	// target one start point per distinct linear-index offset.
	bool device_plan(const std::vector<device_target> &targets, uint64_t min_piece_bytes, file_piece *bytes,
			std::vector<conga_bgzf_block> *blocks, std::vector<conga_bam_segment> *segments, std::string *err,
			const plan_hooks *hooks = nullptr) override
	{
		if (targets.empty() || linear_.empty())
			return false;
		uint64_t c_lo = ~0ull;
		int last_tid = -1;
		bool any = false;
		for (const device_target &t : targets) {
			if (t.tid < 0 || t.tid >= (int) linear_.size() || t.chrom_len <= 0 || t.chrom_len > INT32_MAX)
				return false;
			if (ref_doubt_[(size_t) t.tid])
				return false; // (the host decoders scan for this target; so must the run that would have decoded on the GPU)
			if (ref_beg_[(size_t) t.tid] == 0)
				continue; // no records: no start points, it gets no reads
			any = true;
			c_lo = std::min(c_lo, ref_beg_[(size_t) t.tid] >> 16);
			last_tid = std::max(last_tid, t.tid);
		}
		if (!any)
			return false;
		uint64_t c_end = 0; // file offset of the last block to take; 0: to the end of the file
		for (size_t t = (size_t) last_tid + 1; t < ref_beg_.size(); t++)
			if (ref_beg_[t] != 0) {
				c_end = ref_beg_[t] >> 16;
				break;
			}
		struct stat st;
		if (stat(path_.c_str(), &st) != 0 || (uint64_t) st.st_size <= c_lo)
			return false;
		uint64_t stop = (uint64_t) st.st_size;
		if (c_end && c_end + 65536 + 18 < stop)
			stop = c_end + 65536 + 18; // enough for the whole block that starts at c_end
		uint64_t max_piece = 32ull << 30; // (a 5x genome with sequences is 13 GB of file and 27 GB inflated, resident together: 288 GB of HBM)
		if (knobs().gpu_bam_max_mb > 0)
			max_piece = (uint64_t) (knobs().gpu_bam_max_mb * 1048576.0); // (fractions allowed: tests)
		if (stop - c_lo < min_piece_bytes || stop - c_lo > max_piece)
			return false; // (nothing has been read yet)
		// The stretch is mapped and its pages touched on all cores (this runs beside the HIP runtime's start, or beside the
		// sample before in a cohort): the upload's host threads then fill the pinned pieces at memcpy's pace -- 10 ms each
		// against 30-40 ms with pread (conga_reads_bgzf_fd: no mapping, 26 bytes of every block read for the table below),
		// and the pieces are up after 60 ms instead of 80-90 (profiles/r02c_upload_mmap_vs_pread.log).
		// A process that goes on to other samples (`conga --cohort`) has to give the mapping back -- 75 ms of munmap for 3 GB of
		// touched pages, on its critical path or as TLB shootdowns under the next sample's threads -- and is better off with
		// pread: ten genomes in 1.76 s against 1.90-1.99 s (map_bam_pieces, bam_data.cpp).  CONGA_BAM_MMAP=0 / 1 decides for both.
		const int mm = knobs().bam_mmap;
		if ((mm >= 0 ? mm != 0 : map_bam_pieces) ? !bytes->open(path_, c_lo, stop) : !bytes->open_fd(path_, c_lo, stop))
			return false;
		// known block starts inside the stretch, from the linear indexes of the targets
		std::vector<uint64_t> starts;
		for (const device_target &t : targets)
			if (ref_beg_[(size_t) t.tid] != 0)
				for (uint64_t v : linear_[(size_t) t.tid])
					if (v != 0 && (v >> 16) > c_lo && (v >> 16) < (c_end ? c_end : stop))
						starts.push_back(v >> 16);
		std::sort(starts.begin(), starts.end());
		starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
		// (which bytes the GPU will be given is known from the index alone: a cohort lets the engine start on them now -- and
		// read the block table off them on their way, conga_reads_bgzf_next_fd)
		if (hooks && hooks->named) {
			std::vector<uint64_t> rel{0};
			for (uint64_t v : starts)
				rel.push_back(v - c_lo);
			hooks->named(*bytes, rel, c_end ? c_end - c_lo + 1 : 0); // (one more than the offset: 0 says "none", include/conga_hip.h)
		}
		// ---- block table.  A BGZF file is a chain (every header says where the next block starts), but the index knows
		// thousands of block starts along it: the stretch is cut at some of them and every part is walked by its own thread;
		// a part must arrive exactly at the next part's start, otherwise the index is not trusted and the chain is walked
		// from the front.
		struct found {
			conga_bgzf_block b;
			uint64_t file_off;
		};
		// 1: *b filled but for its trailer, *next set; 0: the piece ends inside this block; -1: not a block.  h = the header's
		// first eighteen bytes (everything a BGZF writer's header has; a longer extra field is read behind them)
		auto block_header = [&](size_t at, const uint8_t *h, conga_bgzf_block *b, size_t *next) -> int {
			if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4))
				return -1;
			const unsigned xlen = h[10] | (h[11] << 8);
			int bsize = -1;
			if (xlen == 6 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0)
				bsize = h[16] | (h[17] << 8); // (what every BGZF writer produces)
			else {
				if (at + 12 + xlen > bytes->size)
					return 0;
				std::vector<uint8_t> extra(xlen);
				if (!bytes->read_at(at + 12, extra.data(), xlen))
					return 0;
				for (unsigned i = 0; i + 4 <= xlen;) {
					const unsigned slen = extra[i + 2] | (extra[i + 3] << 8);
					if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2 && i + 6 <= xlen)
						bsize = extra[i + 4] | (extra[i + 5] << 8);
					i += 4 + slen;
				}
			}
			if (bsize < 0 || bsize + 1 < (int) (12 + xlen + 8))
				return -1;
			if (at + (size_t) bsize + 1 > bytes->size)
				return 0; // the piece ends inside this block (only behind c_end)
			memset(b, 0, sizeof *b);
			b->data_off = at + 12 + xlen;
			b->data_len = (uint32_t) ((size_t) bsize + 1 - 12 - xlen - 8);
			*next = at + (size_t) bsize + 1;
			return 1;
		};
		// walks [from, until) of the piece; *arrived = where it stopped; false: something that is not a block.  A block's
		// trailer (CRC32, ISIZE) and the next block's header lie side by side: ONE read of 26 bytes per block (with the bytes
		// behind a descriptor, a cohort's way, every read is a system call -- 0.8 M of them for a 5x genome's 407 000 blocks).
		auto walk = [&](size_t from, size_t until, std::vector<found> *out, size_t *arrived) -> bool {
			size_t at = from;
			uint8_t h[18];
			bool have_header = false;
			while (at < until) {
				if (!have_header && (at + 18 > bytes->size || !bytes->read_at(at, h, 18)))
					break;
				have_header = false;
				conga_bgzf_block b;
				size_t next = 0;
				const int rc = block_header(at, h, &b, &next);
				if (rc < 0)
					return false;
				if (rc == 0)
					break;
				uint8_t tn[26];
				const size_t want = std::min<size_t>(26, bytes->size - (next - 8));
				if (!bytes->read_at(next - 8, tn, want))
					break;
				memcpy(&b.crc32, tn, 4);
				memcpy(&b.inflated_len, tn + 4, 4);
				if (want == 26) {
					memcpy(h, tn + 8, 18);
					have_header = true;
				}
				if (b.inflated_len)
					out->push_back(found{b, c_lo + at});
				const uint64_t this_off = c_lo + at;
				at = next;
				if (c_end && this_off >= c_end)
					break; // the block in which the next target begins is in: enough
			}
			*arrived = at;
			return true;
		};
		std::vector<found> all;
		bool walked = false;
		// the table the engine read off the bytes, if it did (a cohort's sample whose bytes were named ahead)
		std::vector<conga_bgzf_block> given;
		const bool have_given = hooks && hooks->table && hooks->table(&given) && !given.empty();
		const bool check_given = have_given && knobs().check_table; // (tests: both, and they must agree)
		if (have_given && !check_given) {
			for (const conga_bgzf_block &b : given)
				all.push_back(found{b, c_lo + b.data_off - 18});
			walked = true;
		}
		if (!walked) {
			// (beside another sample's upload -- a cohort plans sample k + 1 while sample k goes up -- six threads: the
			// upload's threads and these share one CPU quota, and a throttled upload is what the GPU then waits for)
			const int n_threads = std::min(plan_beside_upload.load() ? 6 : 16, std::max(1, usable_cpus() / reader_share()));
			// (CONGA_BAM_PARALLEL_MIN_KB: the tests walk small files in parts too)
			const long min_kb = knobs().parallel_min_kb;
			const size_t min_bytes = min_kb >= 0 ? (size_t) min_kb << 10 : (size_t) 64 << 20;
			if (n_threads > 1 && starts.size() >= (size_t) n_threads * 4 && bytes->size > min_bytes) {
				std::vector<size_t> cut{0}; // piece offsets of the parts' starts
				for (int k = 1; k < n_threads; k++)
					cut.push_back((size_t) (starts[starts.size() * (size_t) k / (size_t) n_threads] - c_lo));
				cut.erase(std::unique(cut.begin(), cut.end()), cut.end());
				const size_t parts = cut.size();
				std::vector<std::vector<found>> got(parts);
				std::vector<size_t> arrived(parts, 0);
				std::vector<char> ok(parts, 0);
				std::vector<std::thread> pool;
				for (size_t k = 0; k < parts; k++)
					pool.emplace_back([&, k] {
						ok[k] = walk(cut[k], k + 1 < parts ? cut[k + 1] : bytes->size, &got[k], &arrived[k]) ? 1 : 0;
					});
				for (std::thread &th : pool)
					th.join();
				walked = true;
				for (size_t k = 0; k < parts && walked; k++)
					walked = ok[k] && (k + 1 == parts || arrived[k] == cut[k + 1]);
				if (walked)
					for (size_t k = 0; k < parts; k++)
						all.insert(all.end(), got[k].begin(), got[k].end());
				if (knobs().timing)
					fprintf(stderr, "[timing] block table: %zu parts walked side by side%s\n", parts, walked ? "" : " (a part did not arrive at the next one's start: walked from the front instead)");
			}
		}
		if (!walked) {
			all.clear();
			size_t arrived = 0;
			if (!walk(0, bytes->size, &all, &arrived)) {
				*err = "not a BGZF block";
				return false;
			}
		}
		if (check_given) {
			bool same = given.size() == all.size();
			for (size_t k = 0; same && k < all.size(); k++)
				same = memcmp(&given[k], &all[k].b, sizeof(conga_bgzf_block)) == 0;
			if (!same) {
				fprintf(stderr, "\n[CONGA] the block table the engine read off the bytes (%zu blocks) is not the one read from the file (%zu)\n",
						given.size(), all.size());
				abort();
			}
			fprintf(stderr, "\n[CONGA] block table: the engine's and the file's agree (%zu blocks)\n", all.size());
		}
		blocks->clear();
		std::vector<uint64_t> file_off, inflated_off; // per kept block
		uint64_t total = 0;
		for (const found &f : all) {
			blocks->push_back(f.b);
			file_off.push_back(f.file_off);
			inflated_off.push_back(total);
			total += f.b.inflated_len;
		}
		if (blocks->empty())
			return false;
		auto absolute = [&](uint64_t v, uint64_t *out) {
			const uint64_t c = v >> 16, u = v & 0xFFFF;
			const size_t k = (size_t) (std::lower_bound(file_off.begin(), file_off.end(), c) - file_off.begin());
			if (k < file_off.size() && file_off[k] == c && u <= (*blocks)[k].inflated_len) {
				*out = inflated_off[k] + u;
				return true;
			}
			if (u == 0) { // an empty block (not kept): the next one's start
				*out = k < file_off.size() ? inflated_off[k] : total;
				return true;
			}
			return false;
		};
		// start points, target by target
		segments->clear();
		for (const device_target &t : targets) {
			if (ref_beg_[(size_t) t.tid] == 0)
				continue;
			std::vector<uint64_t> lin = linear_[(size_t) t.tid];
			const size_t n_win = std::min(lin.size(), (size_t) ((t.chrom_len + 16383) >> 14));
			lin.resize(n_win);
			for (size_t w = 1; w < n_win; w++)
				if (lin[w] == 0 || lin[w] < lin[w - 1])
					lin[w] = lin[w - 1];
			conga_bam_segment first;
			first.pos_lo = 0;
			first.pos_hi = (int32_t) t.chrom_len;
			first.ref_id = t.tid;
			first.chrom = t.chrom;
			if (!absolute(ref_beg_[(size_t) t.tid], &first.start))
				return false;
			segments->push_back(first);
			uint64_t prev_v = ref_beg_[(size_t) t.tid];
			for (size_t w = 1; w < n_win; w++) {
				if (lin[w] == 0 || lin[w] <= prev_v)
					continue;
				conga_bam_segment sg = first;
				if (!absolute(lin[w], &sg.start))
					return false;
				sg.pos_lo = (int32_t) (w << 14);
				if ((int64_t) sg.pos_lo >= t.chrom_len)
					break;
				segments->back().pos_hi = sg.pos_lo;
				segments->push_back(sg);
				prev_v = lin[w];
			}
		}
		return !segments->empty();
	}

	bool next_full(full_batch *fb, std::string *err) override
	{
		fb->n_reads = fb->n_bytes = 0;
		keep_body_ = true;
		while (!done_ && fb->n_reads < fb->cap_reads) {
			core c;
			if (have_pending_) {
				c = pending_;
				body_.swap(pending_body_);
				have_pending_ = false;
			} else if (!read_core(&c, err)) {
				if (!err->empty())
					return false;
				done_ = true;
				break;
			}
			last_ref_ = c.ref_id;
			if (c.ref_id >= 0 && c.ref_id < tid_)
				continue;
			if (c.ref_id != tid_ || c.pos >= len_) {
				pending_ = c;
				pending_body_ = body_;
				have_pending_ = true;
				done_ = true;
				break;
			}
			if (c.pos < 0)
				continue;
			const size_t l = (size_t) (c.l_seq > 0 ? c.l_seq : 0);
			const size_t need = (l + 1) / 2 + l;
			// body_ = read_name | cigar | seq | qual | aux
			const size_t seq_at = (size_t) c.l_read_name + (size_t) c.n_cigar * 4;
			if (seq_at + need > body_.size()) {
				*err = "corrupt BAM record";
				return false;
			}
			if (fb->n_bytes + need > fb->cap_bytes) { // no room: hand the record over next time
				pending_ = c;
				pending_body_ = body_;
				have_pending_ = true;
				break;
			}
			const size_t k = fb->n_reads++;
			fb->pos[k] = c.pos;
			fb->mapq[k] = c.mapq;
			fb->flag[k] = c.flag;
			fb->l_qseq[k] = c.l_seq;
			fb->data_off[k] = fb->n_bytes;
			memcpy(fb->data + fb->n_bytes, body_.data() + seq_at, need);
			fb->n_bytes += need;
		}
		keep_body_ = false;
		return true;
	}

private:
	struct core {
		int32_t ref_id, pos;
		uint8_t mapq, l_read_name;
		uint16_t flag, n_cigar;
		int32_t l_seq;
	};

	bool read_header(std::string *err)
	{
		char magic[4];
		int32_t l_text, n_ref;
		if (!bgzf_.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0 || !bgzf_.read(&l_text, 4) || l_text < 0) {
			*err = path_ + ": not a BAM file" + (bgzf_.error().empty() ? "" : " (" + bgzf_.error() + ")");
			return false;
		}
		// sizes come from the file: bound them by what the file can hold (deflate expands at most 1032 : 1) before allocating
		uint64_t file_bytes = 0;
		{
			struct stat sb;
			if (stat(path_.c_str(), &sb) == 0 && sb.st_size > 0)
				file_bytes = (uint64_t) sb.st_size;
		}
		const uint64_t most = file_bytes ? file_bytes * 1032u + 64u : (uint64_t) INT32_MAX;
		if ((uint64_t) l_text > most) {
			*err = path_ + ": BAM header text longer than the file can hold";
			return false;
		}
		std::string text((size_t) l_text, '\0');
		if (!bgzf_.read(&text[0], (size_t) l_text) || !bgzf_.read(&n_ref, 4) || n_ref < 0) {
			*err = path_ + ": truncated BAM header";
			return false;
		}
		names_.clear();
		for (int i = 0; i < n_ref; i++) {
			int32_t l_name, l_ref;
			if (!bgzf_.read(&l_name, 4) || l_name <= 0 || (uint64_t) l_name > most) {
				*err = path_ + ": truncated BAM header";
				return false;
			}
			std::string name((size_t) l_name, '\0');
			if (!bgzf_.read(&name[0], (size_t) l_name) || !bgzf_.read(&l_ref, 4)) {
				*err = path_ + ": truncated BAM header";
				return false;
			}
			name.resize(strlen(name.c_str()));
			names_.push_back(name);
		}
		// get_sample_name (common.c:325-352): first token (split on tab / newline) that starts with "SM"
		sample_.clear();
		size_t i = 0;
		while (i < text.size()) {
			size_t j = text.find_first_of("\t\n", i);
			if (j == std::string::npos)
				j = text.size();
			if (j - i >= 3 && text[i] == 'S' && text[i + 1] == 'M') {
				sample_ = text.substr(i + 3, j - i - 3);
				break;
			}
			i = j + 1;
		}
		last_ref_ = -2;
		return true;
	}

	// next record's fixed core; false with empty *err at end of file
	bool read_core(core *c, std::string *err)
	{
		if (bgzf_.at_eof()) {
			if (!bgzf_.error().empty())
				*err = bgzf_.error();
			return false;
		}
		int32_t block_size;
		uint8_t b[32];
		if (!bgzf_.read(&block_size, 4) || block_size < 32 || !bgzf_.read(b, 32)) {
			*err = bgzf_.error().empty() ? "truncated BAM record" : bgzf_.error();
			return false;
		}
		memcpy(&c->ref_id, b, 4);
		memcpy(&c->pos, b + 4, 4);
		c->l_read_name = b[8];
		c->mapq = b[9];
		memcpy(&c->n_cigar, b + 12, 2);
		memcpy(&c->flag, b + 14, 2);
		memcpy(&c->l_seq, b + 16, 4);
		const size_t rest = (size_t) block_size - 32;
		bool ok;
		if (keep_body_) {
			body_.resize(rest);
			ok = bgzf_.read(body_.data(), rest);
		} else
			ok = bgzf_.skip(rest);
		if (!ok) {
			*err = bgzf_.error().empty() ? "truncated BAM record" : bgzf_.error();
			return false;
		}
		return true;
	}

	// .bai: only the smallest chunk start per reference is kept (whole-chromosome queries)
	bool load_bai(const std::string &path)
	{
		FILE *f = fopen(path.c_str(), "rb");
		if (!f)
			return false;
		char magic[4];
		int32_t n_ref;
		std::vector<uint64_t> beg, fin; // per reference: smallest chunk begin, largest chunk end
		std::vector<std::vector<uint64_t>> lin; // per reference: smallest virtual offset of a record overlapping each 16 kb window
		std::vector<uint8_t> doubt;     // per reference: the index names records but its two views of where they begin differ
		bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "BAI\1", 4) == 0 && fread(&n_ref, 4, 1, f) == 1 && n_ref >= 0;
		for (int r = 0; ok && r < n_ref; r++) {
			int32_t n_bin;
			uint64_t first = 0, last = 0;
			bool any_chunk = false;
			ok = fread(&n_bin, 4, 1, f) == 1 && n_bin >= 0;
			for (int b = 0; ok && b < n_bin; b++) {
				uint32_t bin;
				int32_t n_chunk;
				ok = fread(&bin, 4, 1, f) == 1 && fread(&n_chunk, 4, 1, f) == 1 && n_chunk >= 0;
				for (int k = 0; ok && k < n_chunk; k++) {
					uint64_t ce[2];
					ok = fread(ce, 8, 2, f) == 2;
					if (ok && bin != 37450)
						any_chunk = true;
					if (ok && bin != 37450 && ce[0] != 0 && (first == 0 || ce[0] < first))
						first = ce[0];
					if (ok && bin != 37450 && ce[1] > last)
						last = ce[1];
				}
			}
			int32_t n_intv;
			ok = ok && fread(&n_intv, 4, 1, f) == 1 && n_intv >= 0 && n_intv <= (1 << 17);
			std::vector<uint64_t> iv((size_t) (ok ? n_intv : 0));
			ok = ok && (iv.empty() || fread(iv.data(), 8, iv.size(), f) == iv.size());
			// Where a target's records begin is in the index twice: the smallest chunk begin of its bins and the first entry of its
			// linear index (the first record overlaps its own window).  A whole index agrees with itself; one that does not
			// (tools/bam_fuzz.py index: a zeroed chunk made the decode on the GPU take the target for empty while the sequential
			// reader scanned forward and found its reads) is not used for this target: every decoder then scans for it.
			uint64_t first_lin = 0;
			for (size_t w = 0; w < iv.size() && first_lin == 0; w++)
				first_lin = iv[w];
			const bool names_records = any_chunk || first_lin != 0;
			const bool in_doubt = ok && names_records && (first == 0 || first != first_lin);
			if (in_doubt)
				first = 0;
			doubt.push_back(in_doubt ? 1 : 0);
			beg.push_back(first);
			fin.push_back(last);
			lin.push_back(std::move(iv));
		}
		fclose(f);
		if (ok && (int) beg.size() == n_targets()) {
			ref_beg_ = beg;
			ref_end_ = fin;
			ref_doubt_ = doubt;
			linear_ = lin;
			bai_path_ = path;
		}
		return true;
	}

	std::string path_, sample_, bai_path_;
	bgzf_reader bgzf_;
	std::vector<std::string> names_;
	std::vector<uint64_t> ref_beg_, ref_end_;
	std::vector<std::vector<uint64_t>> linear_;
	std::vector<uint8_t> ref_doubt_; // targets whose index entries contradict each other: scanned for, never planned
	std::vector<int32_t> pos_;
	std::vector<uint8_t> mapq_;
	int tid_ = -1, last_ref_ = -2;
	int64_t len_ = 0;
	bool done_ = false, have_pending_ = false, keep_body_ = false;
	core pending_{};
	std::vector<uint8_t> body_, pending_body_;
};

} // namespace

read_source *open_bam(const std::string &path, std::string *err)
{
	bam_file *b = new bam_file();
	if (!b->open(path, err)) {
		delete b;
		return nullptr;
	}
	return b;
}

} // namespace conga_host
