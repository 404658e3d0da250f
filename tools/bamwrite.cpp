// bamwrite.cpp -- a coordinate-sorted BAM + its .bai from raw arrays, written by all the host's cores.
//
// Test / bench infrastructure (the reference reads BAMs through htslib, bam_data.c:253-259; nothing here is on the product's
// path): bench.py's end-to-end legs need a whole-genome BAM on the GPU box (3 GB at 1x) within seconds, which the numpy
// writer of conga_amd/formats.py (one zlib stream at a time) cannot deliver.  Every record has the same layout per
// chromosome -- 12-byte read name, one <l>M CIGAR operation, l bases, l qualities -- so the uncompressed stream is a function
// of the record index, every BGZF block can be built and deflated by any thread, and the index follows from the block sizes.
//
//   bamwrite OUT.bam MANIFEST [--level 1] [--threads N] [--payload 65280]
// MANIFEST, one line per chromosome in file order (the @SQ order):
//   NAME LENGTH N POS.i32 MAPQ.u8 FLAG.u16|- SEQ.u8|- QUAL.u8|- L_SEQ
// POS / MAPQ / FLAG: raw little-endian arrays of N entries (FLAG "-": all 0).  SEQ: N x ((L_SEQ + 1) / 2) bytes as bam_get_seq
// lays them out; QUAL: N x L_SEQ bytes.  SEQ / QUAL "-": pseudo-random bases and run-structured qualities (a BAM of real reads
// deflates to about a third; an all-'A' one to a fiftieth, which would flatter every decoder).
// The .bai holds, per reference, one chunk in bin 0 and the 16 kb linear index (the layout of formats.write_bam_fast).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Mapped {
	const uint8_t *p = nullptr;
	size_t n = 0;
	bool open(const std::string &path, size_t want)
	{
		if (path == "-")
			return true;
		const int fd = ::open(path.c_str(), O_RDONLY);
		struct stat st;
		if (fd < 0 || fstat(fd, &st) != 0 || (size_t) st.st_size < want) {
			fprintf(stderr, "bamwrite: %s: cannot open, or shorter than %zu bytes\n", path.c_str(), want);
			return false;
		}
		n = (size_t) st.st_size;
		if (n) {
			void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
			if (m == MAP_FAILED)
				return false;
			p = static_cast<const uint8_t *>(m);
		}
		close(fd);
		return true;
	}
};

struct Chrom {
	std::string name;
	int64_t length = 0, n = 0;
	int l_seq = 100;
	Mapped pos, mapq, flag, seq, qual;
	uint64_t start = 0;   // of its first record in the uncompressed stream
	uint64_t serial0 = 0; // number of its first record (read names)
	size_t rec_size() const { return 4 + 32 + 12 + 4 + (size_t) (l_seq + 1) / 2 + (size_t) l_seq; }
};

// the cores this process may really use: affinity mask and cgroup CPU quota (a box that shows 256 hardware threads and grants 16)
int usable_cpus()
{
	int n = (int) std::max(1u, std::thread::hardware_concurrency());
	cpu_set_t set;
	if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0)
		n = std::min(n, (int) CPU_COUNT(&set));
	if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char q[32] = "";
		long long period = -1;
		if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
			n = std::min(n, (int) std::max(1LL, (atoll(q) + period - 1) / period));
		fclose(f);
	}
	return std::max(n, 1);
}

uint64_t mix(uint64_t x) // splitmix64
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

int reg2bin(int64_t beg, int64_t end) // SAM specification, section 5.3
{
	--end;
	if (beg >> 14 == end >> 14) return (int) (((1 << 15) - 1) / 7 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (int) (((1 << 12) - 1) / 7 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (int) (((1 << 9) - 1) / 7 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (int) (((1 << 6) - 1) / 7 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (int) (((1 << 3) - 1) / 7 + (beg >> 26));
	return 0;
}

void put32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
void put16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }

void build_record(const Chrom &c, int tid, int64_t j, uint8_t *out)
{
	const int l = c.l_seq;
	const size_t rs = c.rec_size();
	int32_t pos;
	memcpy(&pos, c.pos.p + 4 * j, 4);
	const uint8_t mapq = c.mapq.p[j];
	uint16_t flag = 0;
	if (c.flag.p)
		memcpy(&flag, c.flag.p + 2 * j, 2);
	put32(out, (uint32_t) (rs - 4));
	put32(out + 4, (uint32_t) tid);
	put32(out + 8, (uint32_t) pos);
	out[12] = 12;
	out[13] = mapq;
	put16(out + 14, (uint16_t) reg2bin(pos, (int64_t) pos + std::max(l, 1)));
	put16(out + 16, 1);
	put16(out + 18, flag);
	put32(out + 20, (uint32_t) l);
	put32(out + 24, 0xFFFFFFFFu);
	put32(out + 28, 0xFFFFFFFFu);
	put32(out + 32, 0);
	char name[16];
	snprintf(name, sizeof name, "r%010llu", (unsigned long long) ((c.serial0 + (uint64_t) j) % 10000000000ull));
	memcpy(out + 36, name, 12);
	put32(out + 48, ((uint32_t) l << 4) | 0u);
	uint8_t *sq = out + 52, *ql = sq + (l + 1) / 2;
	if (c.seq.p)
		memcpy(sq, c.seq.p + (size_t) j * (size_t) ((l + 1) / 2), (size_t) (l + 1) / 2);
	else {
		static const uint8_t code[4] = {1, 2, 4, 8};
		uint64_t r = 0;
		for (int i = 0; i < l; i++) {
			if ((i & 31) == 0)
				r = mix((c.serial0 + (uint64_t) j) * 64 + (uint64_t) (i >> 5));
			const uint8_t b = code[r & 3];
			r >>= 2;
			if (i & 1)
				sq[i >> 1] |= b;
			else
				sq[i >> 1] = (uint8_t) (b << 4);
		}
	}
	if (c.qual.p)
		memcpy(ql, c.qual.p + (size_t) j * (size_t) l, (size_t) l);
	else {
		// binned qualities in runs (what a sequencer of the last decade writes): mostly 37, stretches of 25 / 11, a tail of 2
		static const uint8_t level[8] = {37, 37, 37, 37, 37, 25, 25, 11};
		uint64_t r = mix((c.serial0 + (uint64_t) j) ^ 0x5151515151ull);
		int i = 0;
		while (i < l) {
			const uint8_t q = level[r & 7];
			int run = 4 + (int) ((r >> 3) & 31);
			r = mix(r);
			for (; run > 0 && i < l; run--)
				ql[i++] = q;
		}
		if ((r & 15) == 0)
			for (int k = std::max(0, l - 8); k < l; k++)
				ql[k] = 2;
	}
}

} // namespace

int main(int argc, char **argv)
{
	if (argc < 3) {
		fprintf(stderr, "usage: bamwrite OUT.bam MANIFEST [--level 1] [--threads N] [--payload 65280] [--sample NAME]\n");
		return 2;
	}
	const std::string out_path = argv[1], manifest = argv[2];
	int level = 1, threads = usable_cpus();
	size_t payload = 65280;
	std::string sample = "S";
	for (int i = 3; i + 1 < argc; i += 2) {
		const std::string k = argv[i];
		if (k == "--level") level = atoi(argv[i + 1]);
		else if (k == "--threads") threads = std::max(1, atoi(argv[i + 1]));
		else if (k == "--payload") payload = (size_t) std::max(1024, std::min(atoi(argv[i + 1]), 65280));
		else if (k == "--sample") sample = argv[i + 1];
	}
	std::vector<Chrom> chroms;
	{
		FILE *f = fopen(manifest.c_str(), "r");
		if (!f) {
			fprintf(stderr, "bamwrite: cannot open %s\n", manifest.c_str());
			return 1;
		}
		char nm[256], p1[1024], p2[1024], p3[1024], p4[1024], p5[1024];
		long long len, n;
		int l;
		while (fscanf(f, "%255s %lld %lld %1023s %1023s %1023s %1023s %1023s %d", nm, &len, &n, p1, p2, p3, p4, p5, &l) == 9) {
			chroms.emplace_back();
			Chrom &c = chroms.back();
			c.name = nm;
			c.length = len;
			c.n = n;
			c.l_seq = l;
			if (l < 0 || l > 60000 || n < 0 || !c.pos.open(p1, (size_t) n * 4) || !c.mapq.open(p2, (size_t) n) || !c.flag.open(p3, (size_t) n * 2)
					|| !c.seq.open(p4, (size_t) n * (size_t) ((l + 1) / 2)) || !c.qual.open(p5, (size_t) n * (size_t) l))
				return 1;
			if (n && (!c.pos.p || !c.mapq.p)) {
				fprintf(stderr, "bamwrite: chromosome %s needs POS and MAPQ\n", nm);
				return 1;
			}
		}
		fclose(f);
	}
	// ---- the header (BAM specification 4.2)
	std::string head;
	{
		std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
		for (const Chrom &c : chroms)
			text += "@SQ\tSN:" + c.name + "\tLN:" + std::to_string(c.length) + "\n";
		text += "@RG\tID:rg1\tSM:" + sample + "\tPL:ILLUMINA\n";
		head = "BAM\1";
		const int32_t l_text = (int32_t) text.size(), n_ref = (int32_t) chroms.size();
		head.append(reinterpret_cast<const char *>(&l_text), 4);
		head += text;
		head.append(reinterpret_cast<const char *>(&n_ref), 4);
		for (const Chrom &c : chroms) {
			const int32_t l_name = (int32_t) c.name.size() + 1, l_ref = (int32_t) c.length;
			head.append(reinterpret_cast<const char *>(&l_name), 4);
			head.append(c.name.c_str(), (size_t) l_name);
			head.append(reinterpret_cast<const char *>(&l_ref), 4);
		}
	}
	uint64_t total = head.size(), serial = 0;
	for (Chrom &c : chroms) {
		c.start = total;
		c.serial0 = serial;
		total += (uint64_t) c.n * c.rec_size();
		serial += (uint64_t) c.n;
	}
	const size_t n_blocks = (size_t) ((total + payload - 1) / payload);
	std::vector<std::string> packed(n_blocks);
	std::vector<uint32_t> csize(n_blocks, 0);
	std::vector<uint8_t> done(n_blocks, 0);
	std::mutex mu;
	std::condition_variable cv;
	std::atomic<size_t> next{0};
	size_t written = 0; // blocks the writer has taken (workers stay at most `window` blocks ahead of it)
	const size_t window = 8192;
	bool failed = false;

	auto fill = [&](uint64_t from, uint64_t to, uint8_t *dst) { // the uncompressed stream's bytes [from, to)
		uint64_t at = from;
		if (at < head.size()) {
			const uint64_t k = std::min<uint64_t>(to, head.size()) - at;
			memcpy(dst, head.data() + at, (size_t) k);
			at += k;
		}
		std::vector<uint8_t> rec;
		while (at < to) {
			// the chromosome that holds `at`
			size_t ci = 0;
			while (ci + 1 < chroms.size() && chroms[ci + 1].start <= at)
				ci++;
			while (chroms[ci].n == 0 || at >= chroms[ci].start + (uint64_t) chroms[ci].n * chroms[ci].rec_size())
				ci++;
			const Chrom &c = chroms[ci];
			const size_t rs = c.rec_size();
			rec.resize(rs);
			const int64_t j = (int64_t) ((at - c.start) / rs);
			const uint64_t r0 = c.start + (uint64_t) j * rs;
			build_record(c, (int) ci, j, rec.data());
			const uint64_t a = at - r0, b = std::min<uint64_t>(to, r0 + rs) - r0;
			memcpy(dst + (at - from), rec.data() + a, (size_t) (b - a));
			at = r0 + b;
		}
	};
	auto worker = [&]() {
		std::vector<uint8_t> raw(payload), comp(payload + 1024);
		for (;;) {
			const size_t b = next.fetch_add(1);
			if (b >= n_blocks)
				return;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return failed || b < written + window; });
				if (failed)
					return;
			}
			const uint64_t from = (uint64_t) b * payload, to = std::min<uint64_t>(total, from + payload);
			const size_t n = (size_t) (to - from);
			fill(from, to, raw.data());
			z_stream zs;
			memset(&zs, 0, sizeof zs);
			bool ok = deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
			size_t clen = 0;
			if (ok) {
				zs.next_in = raw.data();
				zs.avail_in = (uInt) n;
				zs.next_out = comp.data();
				zs.avail_out = (uInt) comp.size();
				ok = deflate(&zs, Z_FINISH) == Z_STREAM_END;
				clen = comp.size() - zs.avail_out;
				deflateEnd(&zs);
			}
			if (ok && clen + 26 > 65536) { // (does not happen with a payload of 65280 and a real level; stored blocks would)
				ok = false;
			}
			std::string blk;
			if (ok) {
				static const uint8_t hd[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
				blk.assign(reinterpret_cast<const char *>(hd), 16);
				const uint16_t bsize = (uint16_t) (clen + 25);
				blk.append(reinterpret_cast<const char *>(&bsize), 2);
				blk.append(reinterpret_cast<const char *>(comp.data()), clen);
				const uint32_t crc = (uint32_t) crc32(crc32(0L, Z_NULL, 0), raw.data(), (uInt) n), isize = (uint32_t) n;
				blk.append(reinterpret_cast<const char *>(&crc), 4);
				blk.append(reinterpret_cast<const char *>(&isize), 4);
			}
			{
				std::lock_guard<std::mutex> g(mu);
				if (!ok)
					failed = true;
				packed[b].swap(blk);
				done[b] = 1;
			}
			cv.notify_all();
			if (!ok)
				return;
		}
	};
	FILE *out = fopen(out_path.c_str(), "wb");
	if (!out) {
		fprintf(stderr, "bamwrite: cannot write %s\n", out_path.c_str());
		return 1;
	}
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; t++)
		pool.emplace_back(worker);
	std::vector<uint64_t> block_off(n_blocks + 1, 0);
	uint64_t file_at = 0;
	for (size_t b = 0; b < n_blocks; b++) {
		std::string blk;
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return failed || done[b]; });
			if (failed)
				break;
			blk.swap(packed[b]);
			written = b + 1;
		}
		cv.notify_all();
		block_off[b] = file_at;
		csize[b] = (uint32_t) blk.size();
		if (fwrite(blk.data(), 1, blk.size(), out) != blk.size()) {
			std::lock_guard<std::mutex> g(mu);
			failed = true;
			break;
		}
		file_at += blk.size();
	}
	{
		std::lock_guard<std::mutex> g(mu);
		if (failed)
			written = n_blocks + window; // (let every worker go)
	}
	cv.notify_all();
	for (std::thread &t : pool)
		t.join();
	if (failed) {
		fprintf(stderr, "bamwrite: failed (deflate, or the disk)\n");
		return 1;
	}
	block_off[n_blocks] = file_at; // the EOF block: where a virtual offset behind the last record points
	static const uint8_t eof_block[28] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	fwrite(eof_block, 1, sizeof eof_block, out);
	if (fclose(out) != 0)
		return 1;

	// ---- the index (SAM specification 5.2): per reference one chunk in bin 0 and the 16 kb linear index
	auto voffset = [&](uint64_t stream_at) {
		const uint64_t b = stream_at / payload;
		return (block_off[(size_t) b] << 16) | (stream_at % payload);
	};
	FILE *bai = fopen((out_path + ".bai").c_str(), "wb");
	if (!bai)
		return 1;
	const int32_t n_ref = (int32_t) chroms.size();
	fwrite("BAI\1", 1, 4, bai);
	fwrite(&n_ref, 4, 1, bai);
	for (const Chrom &c : chroms) {
		if (c.n == 0) {
			const int32_t zero[2] = {0, 0};
			fwrite(zero, 4, 2, bai);
			continue;
		}
		const int32_t *pos = reinterpret_cast<const int32_t *>(c.pos.p);
		const size_t rs = c.rec_size();
		const int32_t one = 1;
		const uint32_t bin0 = 0;
		const uint64_t chunk[2] = {voffset(c.start), voffset(c.start + (uint64_t) c.n * rs)};
		fwrite(&one, 4, 1, bai);
		fwrite(&bin0, 4, 1, bai);
		fwrite(&one, 4, 1, bai);
		fwrite(chunk, 8, 2, bai);
		const int32_t n_intv = (int32_t) (((int64_t) pos[c.n - 1] + std::max(c.l_seq, 1) - 1) >> 14) + 1;
		std::vector<uint64_t> lin((size_t) n_intv, 0);
		int64_t j = 0;
		for (int32_t w = 0; w < n_intv; w++) {
			// the first record overlapping the window: the first one that ends behind the window's start
			const int64_t need = (int64_t) w * 16384 - std::max(c.l_seq, 1) + 1;
			while (j < c.n && (int64_t) pos[j] < need)
				j++;
			if (j < c.n && w >= (pos[0] >> 14))
				lin[(size_t) w] = voffset(c.start + (uint64_t) j * rs);
		}
		fwrite(&n_intv, 4, 1, bai);
		fwrite(lin.data(), 8, lin.size(), bai);
	}
	const uint64_t no_coor = 0;
	fwrite(&no_coor, 8, 1, bai);
	fclose(bai);
	fprintf(stderr, "bamwrite: %llu records, %zu blocks, %.1f MB -> %.1f MB, %d threads\n", (unsigned long long) serial, n_blocks, total / 1e6,
			(file_at + 28) / 1e6, threads);
	return 0;
}
