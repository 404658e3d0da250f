// params.h -- run parameters of the `conga` command line.
// Mirrors the fields of the reference's `parameters` (common.h:62-86) that the depth / likelihood
// path reads; names are kept so the driver reads like the reference's.
#pragma once
#include <string>

namespace conga_host {

// exit / return codes of the reference (common.h:25-37)
enum { CONGA_EXIT_SUCCESS = 0, CONGA_EXIT_COMMON = 1, CONGA_EXIT_PARAM_ERROR = 3 };
enum { RETURN_ERROR = 0, RETURN_SUCCESS = 1 };

struct parameters {
	std::string ref_genome;       // --ref   (required by the reference even when unused: cmdline.c:142-146)
	std::string outdir;           // --out up to and including the last '/' (common.c:45-83)
	std::string outprefix;        // rest of --out
	std::string low_map_regions;  // --exclude (parsed, no effect on outputs: SURVEY.md section 2)
	std::string dup_file;         // --dups
	std::string del_file;         // --dels
	std::string bam_file;         // --input
	std::string sonic_file;       // --sonic
	std::string sonic_info;       // --sonic-info
	std::string mappability_file; // --mappability
	bool have_outprefix = false, have_ref = false, have_dels = false, have_dups = false, have_map = false;
	int min_sv_size = 0;     // --min-sv-size, <= 0 -> 1000 (cmdline.c:156-160)
	int min_read_length = 0; // --min-read-length, <= 0 -> 60 (cmdline.c:162-166)
	int first_chrom = 0;     // --first-chr
	int last_chrom = -1;     // --last-chr
	float c_score = 0.5f;    // --c-score (cmdline.c:168-174)
	int mq_threshold = -1;   // --min-mapq (cmdline.c:188-194)
	int rp_support = 10;     // --rp (cmdline.c:176-186)
	int no_sr = 1;           // 0 when --rp was given
	// extensions of this implementation (not in the reference)
	int device = 0;                  // --device
	int n_gpus = 1;                  // --gpus N : N contexts (one host thread + one HIP device each), chromosomes sharded longest-first
	std::string dump_intervals_chr;  // --dump-intervals CHR : print the kept, sorted SV rows and exit (no GPU)
	std::string dump_mappability_chr; // --dump-mappability CHR : print the parsed mappability rows of CHR and exit (no GPU)
	std::string cohort_file;         // --cohort FILE : one BAM per line (optionally: tab, output prefix); every sample is genotyped
	                                 // in this one process against the same call set -- the engine context, its device layout
	                                 // and the HIP runtime are kept from sample to sample (cohort mode of include/conga_hip.h)
	bool dump_reads = false;         // --dump-reads : per chromosome, count and checksums of the records the BAM loop would count (no GPU)
};

} // namespace conga_host
