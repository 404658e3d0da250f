// knobs.h -- every switch the `conga` executable takes from the environment, read once (the first call of knobs()).
// None of them is part of the reference's interface (cmdline.c knows options only); INTEGRATION.md lists them.  The engine's own
// switches are conga_amd/csrc/engine_knobs.h.
#pragma once
#include <stdlib.h>

namespace conga_host {

struct host_knobs {
	int gpu_bam = -1;            // CONGA_GPU_BAM: 0 never / 1 always decode the BAM on the GPU (-1: when it pays)
	bool timing = false;         // CONGA_TIMING: wall time of the host phases on stderr
	long long t0_ns = 0;         // CONGA_T0_NS: the caller's clock when it started the process (with CONGA_TIMING)
	bool clean_exit = false;     // CONGA_CLEAN_EXIT: tear everything down in order at the end (leak checkers)
	int cohort_ahead = -1;       // CONGA_COHORT_AHEAD: samples named ahead to the engine (-1: two, one with split reads)
	int bam_threads = -1;        // CONGA_BAM_THREADS: inflating threads of the sequential BAM reader (-1: by the cores)
	bool zlib_inflate = false;   // CONGA_ZLIB_INFLATE: every block through zlib (not the reader's own decoder)
	int bam_segments = -1;       // CONGA_BAM_SEGMENTS: segments of the index-guided parallel decode (-1: by size and cores; 0 / 1: off)
	double gpu_bam_max_mb = 0;   // CONGA_GPU_BAM_MAX_MB: largest piece of file one GPU decode takes (0: 32 GB)
	int bam_mmap = -1;           // CONGA_BAM_MMAP: 0 pread / 1 map the file's stretch (-1: map a single run, pread a cohort)
	bool check_table = false;    // CONGA_BGZF_CHECK_TABLE: read the block table from the file as well and compare with the engine's
	long parallel_min_kb = -1;   // CONGA_BAM_PARALLEL_MIN_KB: from this size on the block table is walked in parts (-1: default)
	bool bed_literal = false;    // CONGA_BED_LITERAL: the reference-literal fgets / strtok BED reader
	bool trace = false;          // CONGA_DEBUG=1 CONGA_BGZF_TRACE=1: the sample loop's events with the engine's trace clock (bz_sched.h: trace)
	bool host_packed = true;     // CONGA_HOST_PACKED=0: a cohort's further samples from the host decoders go through the staging ring,
	                             // chromosome by chromosome, instead of conga_packer_* + conga_sample_reads_packed (tests compare the two)
};

inline const host_knobs &knobs()
{
	static const host_knobs k = [] {
		host_knobs h;
		if (const char *e = getenv("CONGA_GPU_BAM"))
			h.gpu_bam = atoi(e) != 0 ? 1 : 0;
		h.timing = getenv("CONGA_TIMING") != nullptr;
		if (const char *e = getenv("CONGA_T0_NS"))
			h.t0_ns = atoll(e);
		h.clean_exit = getenv("CONGA_CLEAN_EXIT") != nullptr;
		if (const char *e = getenv("CONGA_COHORT_AHEAD"))
			h.cohort_ahead = atoi(e);
		if (const char *e = getenv("CONGA_BAM_THREADS"))
			h.bam_threads = atoi(e);
		h.zlib_inflate = getenv("CONGA_ZLIB_INFLATE") != nullptr;
		if (const char *e = getenv("CONGA_BAM_SEGMENTS"))
			h.bam_segments = atoi(e);
		if (const char *e = getenv("CONGA_GPU_BAM_MAX_MB"))
			h.gpu_bam_max_mb = atof(e);
		if (const char *e = getenv("CONGA_BAM_MMAP"))
			h.bam_mmap = atoi(e) != 0 ? 1 : 0;
		h.check_table = getenv("CONGA_BGZF_CHECK_TABLE") != nullptr;
		if (const char *e = getenv("CONGA_BAM_PARALLEL_MIN_KB"))
			h.parallel_min_kb = atol(e);
		h.bed_literal = getenv("CONGA_BED_LITERAL") != nullptr;
		h.trace = getenv("CONGA_BGZF_TRACE") != nullptr && getenv("CONGA_DEBUG") != nullptr && atoi(getenv("CONGA_DEBUG")) != 0;
		if (const char *e = getenv("CONGA_HOST_PACKED"))
			h.host_packed = atoi(e) != 0;
		return h;
	}();
	return k;
}

} // namespace conga_host
