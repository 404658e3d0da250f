#include "reads.h"

#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <sched.h>

#include <atomic>
#include <thread>
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>

namespace conga_host {

read_source *open_bam(const std::string &path, std::string *err); // bam_reader.cpp

namespace {

// Memory-mapped read-tuple container (layout: conga_amd/formats.py write_tuples).  CONGATP2 pads every array to a
// 16-byte boundary; in a CONGATP1 file the arrays sit wherever the header ends, so positions that are not 4-byte
// aligned are copied once at open.
class tuple_file : public read_source {
public:
	~tuple_file() override
	{
		if (map_ && map_ != MAP_FAILED)
			munmap(map_, size_);
		if (fd_ >= 0)
			close(fd_);
	}
	bool open(const std::string &path, std::string *err)
	{
		fd_ = ::open(path.c_str(), O_RDONLY);
		struct stat st;
		if (fd_ < 0 || fstat(fd_, &st) != 0) {
			*err = "cannot open " + path;
			return false;
		}
		size_ = (size_t) st.st_size;
		map_ = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
		if (map_ == MAP_FAILED) {
			*err = "cannot map " + path;
			return false;
		}
		const uint8_t *p = (const uint8_t *) map_, *end = p + size_;
		auto need = [&](size_t n) { return (size_t) (end - p) >= n; };
		if (!need(8) || (memcmp(p, "CONGATP1", 8) != 0 && memcmp(p, "CONGATP2", 8) != 0)) {
			*err = path + ": not a CONGATP1 / CONGATP2 read-tuple container";
			return false;
		}
		const bool aligned = p[7] == '2';
		const uint8_t *const base = p;
		auto align16 = [&]() {
			if (aligned)
				p = base + (((size_t) (p - base) + 15) & ~(size_t) 15);
			return p <= end;
		};
		p += 8;
		uint16_t ln;
		if (!need(2))
			return trunc(err);
		memcpy(&ln, p, 2);
		p += 2;
		if (!need(ln))
			return trunc(err);
		sample_.assign((const char *) p, ln);
		p += ln;
		int32_t n;
		if (!need(4))
			return trunc(err);
		memcpy(&n, p, 4);
		p += 4;
		if (n < 0)
			return trunc(err);
		for (int c = 0; c < n; c++) {
			if (!need(2))
				return trunc(err);
			memcpy(&ln, p, 2);
			p += 2;
			if (!need((size_t) ln + 17))
				return trunc(err);
			entry e;
			e.name.assign((const char *) p, ln);
			p += ln;
			memcpy(&e.length, p, 8);
			memcpy(&e.n, p + 8, 8);
			e.ext = p[16];
			p += 17;
			if (e.n < 0)
				return trunc(err);
			chroms_.push_back(e);
		}
		for (entry &e : chroms_) {
			// pos int32[n], mapq uint8[n] and, with ext, flag uint16[n], l_qseq int32[n] (not used by this reader)
			const size_t widths[4] = {4, 1, 2, 4};
			for (int k = 0; k < (e.ext ? 4 : 2); k++) {
				if (!align16() || !need((size_t) e.n * widths[k]))
					return trunc(err);
				if (k == 0) {
					if (((uintptr_t) p & 3u) == 0)
						e.pos = (const int32_t *) p;
					else {
						e.pos_copy.resize((size_t) e.n);
						memcpy(e.pos_copy.data(), p, (size_t) e.n * 4);
						e.pos = e.pos_copy.data();
					}
				} else if (k == 1)
					e.mapq = p;
				p += (size_t) e.n * widths[k];
			}
		}
		return true;
	}
	int n_targets() const override { return (int) chroms_.size(); }
	const std::string &target_name(int tid) const override { return chroms_[tid].name; }
	const std::string &sample_name() const override { return sample_; }
	bool begin(int tid, int64_t chrom_len, std::string *err) override
	{
		if (tid < 0 || tid >= (int) chroms_.size()) {
			*err = "bad target id";
			return false;
		}
		cur_ = tid;
		off_ = 0;
		len_ = chrom_len;
		return true;
	}
	bool next(size_t max_n, read_batch *out, std::string *) override
	{
		const entry &e = chroms_[cur_];
		// records are coordinate-sorted: skip pos < 0, stop at the first pos >= L (region [0, L))
		while (off_ < e.n && e.pos[off_] < 0)
			off_++;
		size_t n = 0;
		while (off_ + (int64_t) n < e.n && n < max_n && e.pos[off_ + n] < len_)
			n++;
		out->pos = e.pos + off_;
		out->mapq = e.mapq + off_;
		out->n = n;
		off_ += (int64_t) n;
		return true;
	}
	bool whole(int tid, int64_t chrom_len, const int32_t **pos, const uint8_t **mapq, size_t *n, std::string *err) override
	{
		if (tid < 0 || tid >= (int) chroms_.size()) {
			*err = "bad target id";
			return false;
		}
		// what begin() + next() yield -- the records behind the leading pos < 0, up to the first pos >= L -- found by two searches
		// instead of a walk over every position (the walk is 100 MB of a 1x genome read once more)
		const entry &e = chroms_[tid];
		const int32_t *a = e.pos, *b = e.pos + e.n;
		if (e.n > 0 && a[0] < 0)
			a = std::lower_bound(a, b, (int32_t) 0);
		if (b > a && (int64_t) b[-1] >= chrom_len)
			b = std::partition_point(a, b, [&](int32_t p) { return (int64_t) p < chrom_len; });
		*pos = a;
		*mapq = e.mapq + (a - e.pos);
		*n = (size_t) (b - a);
		return true;
	}

private:
	struct entry {
		std::string name;
		int64_t length = 0, n = 0;
		int ext = 0;
		const int32_t *pos = nullptr;
		const uint8_t *mapq = nullptr;
		std::vector<int32_t> pos_copy; // CONGATP1 with unaligned positions
	};
	bool trunc(std::string *err)
	{
		*err = "truncated read-tuple container";
		return false;
	}
	int fd_ = -1;
	void *map_ = nullptr;
	size_t size_ = 0;
	std::string sample_;
	std::vector<entry> chroms_;
	int cur_ = -1;
	int64_t off_ = 0, len_ = 0;
};

} // namespace

bool map_bam_pieces = true;

file_piece::~file_piece()
{
	if (map && map != MAP_FAILED) {
		// Gigabytes of touched pages: dropping the page-table entries is what takes the time (~75 ms for 3 GB), and munmap does it
		// holding the address space's lock for WRITING -- every malloc that grows the heap and every page fault of the process
		// waits.  MADV_DONTNEED drops them under the read lock; the munmap behind it has nothing left to do.  (`conga --cohort`
		// gives a sample's mapping back on a thread of its own while the next sample is at work.)
		(void) madvise(map, map_len, MADV_DONTNEED);
		munmap(map, map_len);
	}
	if (fd >= 0)
		close(fd);
}

bool file_piece::open_fd(const std::string &path, uint64_t from, uint64_t to)
{
	fd = ::open(path.c_str(), O_RDONLY);
	if (fd < 0)
		return false;
	file_off = from;
	size = (size_t) (to - from);
	data = nullptr;
	return true;
}

bool file_piece::read_at(uint64_t off, void *dst, size_t n) const
{
	if (off > size || n > size - off)
		return false;
	if (data) {
		memcpy(dst, data + off, n);
		return true;
	}
	uint8_t *p = static_cast<uint8_t *>(dst);
	while (n) {
		const ssize_t got = pread(fd, p, n, (off_t) (file_off + off));
		if (got <= 0)
			return false;
		p += got;
		off += (uint64_t) got;
		n -= (size_t) got;
	}
	return true;
}

bool file_piece::open(const std::string &path, uint64_t from, uint64_t to)
{
	const int fd = ::open(path.c_str(), O_RDONLY);
	if (fd < 0)
		return false;
	const uint64_t page = (uint64_t) sysconf(_SC_PAGESIZE), base = from & ~(page - 1);
	map_len = (size_t) (to - base);
	map = mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, (off_t) base);
	close(fd);
	if (map == MAP_FAILED) {
		map = nullptr;
		return false;
	}
	(void) madvise(map, map_len, MADV_SEQUENTIAL);
	data = (const uint8_t *) map + (from - base);
	size = (size_t) (to - from);
	// Map the pages in now, on every core the process may use: the block-table scan and above all the upload would
	// otherwise take one page fault per 4 KB on a single thread.
	const int n_threads = std::min(16, std::max(1, usable_cpus() / reader_share()));
	if (n_threads > 1 && map_len > (64u << 20)) {
		std::vector<std::thread> pool;
		const size_t per = ((map_len / (size_t) n_threads) + page - 1) & ~(size_t) (page - 1);
		for (int t = 0; t < n_threads; t++)
			pool.emplace_back([=] {
				const volatile uint8_t *p = (const volatile uint8_t *) map;
				const size_t lo = (size_t) t * per, hi = std::min(map_len, lo + per);
				unsigned sink = 0;
				for (size_t at = lo; at < hi; at += page)
					sink += p[at];
				(void) sink;
			});
		for (std::thread &th : pool)
			th.join();
	}
	return true;
}

namespace {
std::atomic<int> g_reader_share{1};
}
void set_reader_share(int n) { g_reader_share = n < 1 ? 1 : n; }
int reader_share() { return g_reader_share; }

// CPUs this process may actually use: the affinity mask, cut down to a cgroup v2 / v1 CPU quota when one is set
// (a container on a 256-thread host is often allowed 16).
std::atomic<bool> plan_beside_upload{false};

int usable_cpus()
{
	int n = (int) std::thread::hardware_concurrency();
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
		n = std::min(n, (int) CPU_COUNT(&set));
	long long quota = -1, period = -1;
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[32] = "";
		if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0)
			quota = atoll(q);
		fclose(f);
	} else {
		FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
		if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1) {
		} else {
			quota = -1;
		}
		if (fq)
			fclose(fq);
		if (fp)
			fclose(fp);
	}
	if (quota > 0 && period > 0)
		n = std::min(n, (int) std::max(1LL, (quota + period - 1) / period));
	return std::max(n, 1);
}

read_source *open_reads(const std::string &path, std::string *err)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) {
		*err = "[CONGA INPUT ERROR] Unable to open file " + path + " in read mode.";
		return nullptr;
	}
	unsigned char magic[8] = {0};
	const size_t got = fread(magic, 1, 8, f);
	fclose(f);
	if (got == 8 && (memcmp(magic, "CONGATP1", 8) == 0 || memcmp(magic, "CONGATP2", 8) == 0)) {
		tuple_file *t = new tuple_file();
		if (!t->open(path, err)) {
			delete t;
			return nullptr;
		}
		return t;
	}
	if (got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b) // gzip member: BGZF-compressed BAM
		return open_bam(path, err);
	*err = path + ": neither a BAM nor a CONGATP1 / CONGATP2 read-tuple container (CRAM is not supported)";
	return nullptr;
}

bool load_fasta_chrom(const std::string &fasta_path, const std::string &name, int64_t chrom_len, std::string *seq,
		std::string *err)
{
	FILE *f = fopen(fasta_path.c_str(), "rb");
	if (!f) {
		*err = "[CONGA INPUT ERROR] Unable to open file " + fasta_path + " in read mode.";
		return false;
	}
	seq->clear();
	seq->reserve((size_t) chrom_len);
	bool found = false;
	// .fai: name, length, offset, bases per line, bytes per line
	FILE *fai = fopen((fasta_path + ".fai").c_str(), "r");
	if (fai) {
		char nm[1024];
		long long len, off, lb, lw;
		while (fscanf(fai, "%1023s %lld %lld %lld %lld", nm, &len, &off, &lb, &lw) == 5) {
			if (name == nm && lb > 0 && lw >= lb) {
				found = true;
				const long long want = std::min<long long>(len, chrom_len);
				std::vector<char> line((size_t) lw);
				fseeko(f, (off_t) off, SEEK_SET);
				while ((long long) seq->size() < want) {
					const size_t got = fread(line.data(), 1, (size_t) lw, f);
					if (got == 0)
						break;
					const size_t take = std::min<size_t>(std::min<size_t>(got, (size_t) lb), (size_t) (want - (long long) seq->size()));
					seq->append(line.data(), take);
				}
				break;
			}
		}
		fclose(fai);
	}
	if (!found) { // no index: scan for ">name"
		std::vector<char> buf(1 << 16);
		bool in_rec = false;
		while (fgets(buf.data(), (int) buf.size(), f)) {
			if (buf[0] == '>') {
				if (in_rec)
					break;
				size_t k = 1;
				while (buf[k] && !isspace((unsigned char) buf[k]))
					k++;
				in_rec = found = (std::string(buf.data() + 1, k - 1) == name);
				continue;
			}
			if (!in_rec)
				continue;
			for (const char *c = buf.data(); *c && (int64_t) seq->size() < chrom_len; c++)
				if (!isspace((unsigned char) *c))
					seq->push_back(*c);
		}
	}
	fclose(f);
	if (!found) {
		*err = "chromosome " + name + " is not in " + fasta_path;
		return false;
	}
	seq->resize((size_t) chrom_len, 'N');
	return true;
}

int find_chr_index_bam(const std::string &chromosome_name, const read_source &src)
{
	for (int i = 0; i < src.n_targets(); i++)
		if (src.target_name(i) == chromosome_name)
			return i;
	return -1;
}

} // namespace conga_host
