// reads.h -- the record source of count_reads_bam (bam_data.c:192-221).
// Two containers are understood: this implementation's read-tuple container (.ctp, conga_amd/formats.py)
// and coordinate-sorted BAM (bam_reader.cpp; needs the .bai only for random access, not used here).
#pragma once
#include <cstdint>
#include <atomic>
#include <functional>
#include <string>
#include <vector>

#include "../../include/conga_hip.h"

namespace conga_host {

struct read_batch {
	const int32_t *pos;
	const uint8_t *mapq;
	size_t n;
};

// Everything find_split_reads touches of a record (split_read.c:206-354), laid out as conga_split_staging wants it.
struct full_batch {
	int32_t *pos;
	uint8_t *mapq;
	uint16_t *flag;
	int32_t *l_qseq;
	uint64_t *data_off;
	uint8_t *data; // per record: (l_qseq + 1) / 2 bytes of packed sequence, then l_qseq quality bytes
	size_t cap_reads, cap_bytes;
	size_t n_reads, n_bytes; // filled by next_full
};

// A stretch of a file, memory-mapped (the page cache is the only copy on the host; the GPU upload reads from it).
// whether device_plan() maps the stretch of the BAM it hands over (a single run) or leaves it to pread (a cohort)
extern bool map_bam_pieces;

struct file_piece {
	const uint8_t *data = nullptr; // the mapped bytes (open), or nullptr when only the descriptor is held (open_fd)
	size_t size = 0;
	void *map = nullptr;
	size_t map_len = 0;
	int fd = -1;                   // open_fd: the file, read with pread -- no mapping, no page faults, nothing to unmap
	uint64_t file_off = 0;         // of the piece's first byte
	bool open_fd(const std::string &path, uint64_t from, uint64_t to);
	bool read_at(uint64_t off, void *dst, size_t n) const; // n bytes at `off` of the PIECE (either form)
	file_piece() {}
	file_piece(const file_piece &) = delete;
	file_piece &operator=(const file_piece &) = delete;
	~file_piece();
	bool open(const std::string &path, uint64_t from, uint64_t to);
};

// What a caller that keeps an engine context across inputs (a cohort) wants to know while an input is being planned:
//   named   the stretch of the file is fixed (bytes: the open descriptor), with the offsets inside it at which the index knows
//           a BGZF block to begin (ascending, the first one 0) and where the table will end (0: at the stretch's end)
//   table   called before the block table is read from the file: true = *blocks holds it already (the engine read it off the
//           bytes on their way to the GPU, conga_reads_bgzf_next_table), data_off relative to the stretch
struct plan_hooks {
	std::function<void(const file_piece &bytes, const std::vector<uint64_t> &known_starts, uint64_t stop_at)> named;
	std::function<bool(std::vector<conga_bgzf_block> *blocks)> table;
};

struct device_target {
	int tid;           // in the BAM header
	int64_t chrom_len; // from the annotation
	int chrom;         // index of the chromosome in the engine context
};

class read_source {
public:
	virtual ~read_source() {}
	// header: bam_hdr_t.n_targets / target_name (find_chr_index_bam, common.c:289-300)
	virtual int n_targets() const = 0;
	virtual const std::string &target_name(int tid) const = 0;
	virtual const std::string &sample_name() const = 0; // @RG SM (get_sample_name, common.c:325-352)
	virtual std::string index_path() const { return ""; } // the index in use for seeking ("" = none: read front to back)
	// Iterate the records of target `tid` with 0 <= pos < chrom_len, in file order, at most max_n at a time
	// (the analogue of sam_itr_queryi(idx, tid, 0, L) + sam_itr_next: bam_data.c:293,201).
	virtual bool begin(int tid, int64_t chrom_len, std::string *err) = 0;
	virtual bool next(size_t max_n, read_batch *out, std::string *err) = 0;
	// The same records as begin() + next() until exhaustion, all at once, decoded by up to `threads` readers working on
	// disjoint position ranges of the target (a BAM with the linear offsets of its .bai).  false with an empty *err: not
	// available for this source / target -- iterate instead; false with a message: the file or its index is broken.
	virtual bool read_all(int tid, int64_t chrom_len, int threads, std::vector<int32_t> *pos, std::vector<uint8_t> *mapq,
			std::string *err)
	{
		return false;
	}
	// The same records where they already lie as arrays (a container that holds them so): valid until the source goes.  false with an
	// empty *err: this source has no such arrays -- read_all() or iterate.  (Position order is taken for granted when the ends are cut
	// to [0, chrom_len): the engine refuses reads that are out of order.)
	virtual bool whole(int tid, int64_t chrom_len, const int32_t **pos, const uint8_t **mapq, size_t *n, std::string *err)
	{
		return false;
	}
	// What conga_reads_bgzf (include/conga_hip.h) needs to decode the same records on the GPU: the stretch of the file that
	// holds the targets (in the order of their chromosomes in the context), as it is, the table of its BGZF blocks, and start
	// points from the index's linear offsets.  false with an empty *err: not
	// available (no index, not a BAM, a piece of the file smaller than min_piece_bytes or too large) -- decode on the host.
	virtual bool device_plan(const std::vector<device_target> &targets, uint64_t min_piece_bytes, file_piece *bytes,
			std::vector<conga_bgzf_block> *blocks, std::vector<conga_bam_segment> *segments, std::string *err,
			const plan_hooks *hooks = nullptr)
	{
		return false;
	}
	// Same iteration, whole records (--rp).  Sources without sequences return false.
	virtual bool next_full(full_batch *fb, std::string *err)
	{
		*err = "this input holds no read sequences: --rp needs a BAM";
		return false;
	}
};

// readReferenceSeq (common.c:423-463): one chromosome of the --ref FASTA (through its .fai when there is one),
// exactly chrom_len bases; a shorter FASTA record is padded with 'N'.
bool load_fasta_chrom(const std::string &fasta_path, const std::string &name, int64_t chrom_len, std::string *seq,
		std::string *err);

read_source *open_reads(const std::string &path, std::string *err);
// Readers opened from now on size their inflate pool for 1/n of the host's threads (n readers run side by side).
extern std::atomic<bool> plan_beside_upload; // device_plan() runs while another sample's bytes go up (read_bam_cohort)
void set_reader_share(int n);
int reader_share();
int usable_cpus(); // affinity mask and cgroup CPU quota taken into account
int find_chr_index_bam(const std::string &chromosome_name, const read_source &src); // common.c:289-300

} // namespace conga_host
