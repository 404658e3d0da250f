// cmdline.cpp -- the `conga` command line: same options, defaults and messages as the reference
// (cmdline.c:20-42 option table, :134-206 checks and defaults; App. C of SURVEY.md).
#include "cmdline.h"

#include <algorithm>

#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#ifndef CONGA_VERSION
#define CONGA_VERSION "1.0-mi355x"
#endif
#ifndef CONGA_UPDATE
#define CONGA_UPDATE "round 2"
#endif
#ifndef BUILD_DATE
#define BUILD_DATE __DATE__
#endif

namespace conga_host {

namespace {
enum { OPT_FIRST_CHROM = 10001, OPT_LAST_CHROM = 10002, OPT_DEVICE = 10003, OPT_DUMP = 10004, OPT_DUMP_READS = 10005, OPT_DUMP_MAP = 10006, OPT_GPUS = 10007, OPT_COHORT = 10008 };
}

void print_help(void)
{
	// text of cmdline.c:210-234 (the misspelt "--mapability" is the reference's)
	fprintf(stdout, "\n\t... CONGA (COpy Number variation Genotyping in Ancient genomes) ...\n\n");
	fprintf(stdout, "\tVersion %s\n\tLast update: %s, build date: %s\n\n", CONGA_VERSION, CONGA_UPDATE, BUILD_DATE);
	fprintf(stdout, "\tParameters:\n");
	fprintf(stdout, "\t--input [BAM file]        : Input file in sorted and indexed BAM format (required).\n");
	fprintf(stdout, "\t--out [output prefix]     : Prefix for the output file names (required).\n");
	fprintf(stdout, "\t--ref [reference genome]  : Reference genome in FASTA format (required).\n");
	fprintf(stdout, "\t--sonic [sonic file]      : SONIC file that contains assembly annotations (required).\n");
	fprintf(stdout, "\t--dels [BED file]         : Known deletion SVs in BED format\n");
	fprintf(stdout, "\t--dups [BED file]         : Known duplication SVs in BED format\n");
	fprintf(stdout, "\t--first-chr [chr index]   : The index of the first chromosome for genotyping in your BAM\n");
	fprintf(stdout, "\t--last-chr [chr index]    : The index of the last chromosome for genotyping in your BAM\n");
	fprintf(stdout, "\t--mapability [BED file]   : Mappability file in BED format\n");
	fprintf(stdout, "\t--min-read-length [INT]   : Minimum length of a read to be processed for RP (default: 60 bps)\n");
	fprintf(stdout, "\t--min-sv-size [INT]       : Minimum length of a CNV (default: 1000 bps)\n");
	fprintf(stdout, "\t--min-mapq [INT]          : Minimum mapping quality filter for reads (default: no-filter)\n");
	fprintf(stdout, "\t--c-score [FLOAT]         : Minimum c-score to filter variants (More conservative with lower values, default: 0.5).\n");
	fprintf(stdout, "\t--rp [INT]                : Enable split-read and set minimum read-pair support for a duplication (Suggested for >5x only).");
	fprintf(stdout, "\n\n\tInformation:\n");
	fprintf(stdout, "\t--version                 : Print version and exit.\n");
	fprintf(stdout, "\t--help                    : Print this help screen and exit.\n\n");
	fprintf(stderr, "\n\t* For more information, please consult https://github.com/asylvz/CONGA\n\n");
}

// common.c:45-83: everything up to and including the last '/' of --out is the directory
void get_working_directory(parameters *params)
{
	const size_t slash = params->outprefix.rfind('/');
	if (slash == std::string::npos) {
		params->outdir.clear();
		return;
	}
	const std::string prefix = params->outprefix.substr(slash + 1);
	fprintf(stderr, "prefix: %s\n", prefix.c_str());
	params->outdir = params->outprefix.substr(0, slash + 1);
	fprintf(stderr, "prefix2: %s\n", prefix.c_str());
	params->outprefix = prefix;
}

int parse_cmd_line(int argc, char **argv, parameters *params)
{
	static int no_sr_flag = 1;
	static struct option long_options[] = {
		{"c-score", required_argument, 0, 'a'},
		{"min-read-length", required_argument, 0, 'b'},
		{"dels", required_argument, 0, 'd'},
		{"min-mapq", required_argument, 0, 'e'},
		{"ref", required_argument, 0, 'f'},
		{"help", no_argument, 0, 'h'},
		{"input", required_argument, 0, 'i'},
		{"rp", required_argument, 0, 'j'},
		{"min-sv-size", required_argument, 0, 'l'},
		{"mappability", required_argument, 0, 'm'},
		{"sonic-info", required_argument, 0, 'n'},
		{"out", required_argument, 0, 'o'},
		{"sonic", required_argument, 0, 's'},
		{"dups", required_argument, 0, 'u'},
		{"version", no_argument, 0, 'v'},
		{"exclude", required_argument, 0, 'x'},
		{"no-sr", no_argument, &no_sr_flag, 1},
		{"first-chr", required_argument, 0, OPT_FIRST_CHROM},
		{"last-chr", required_argument, 0, OPT_LAST_CHROM},
		{"device", required_argument, 0, OPT_DEVICE},
		{"gpus", required_argument, 0, OPT_GPUS},
		{"dump-intervals", required_argument, 0, OPT_DUMP},
		{"cohort", required_argument, 0, OPT_COHORT},
		{"dump-reads", no_argument, 0, OPT_DUMP_READS},
		{"dump-mappability", required_argument, 0, OPT_DUMP_MAP},
		{0, 0, 0, 0}};

	if (argc == 1) {
		print_help();
		return 0;
	}

	bool have_c = false, have_mq = false, have_rp = false, load_sonic = false;
	std::string c_score, min_mapping_qual, min_rp_support;
	int index = 0, o;
	optind = 1;
	// same short-option string as cmdline.c:50
	while ((o = getopt_long(argc, argv, "hvb:i:f:g:d:r:o:m:c:s:a:e:n:j:k:u:x", long_options, &index)) != -1) {
		switch (o) {
		case 'a': c_score = optarg; have_c = true; break;
		case 'b': params->min_read_length = atoi(optarg); break;
		case 'd': params->del_file = optarg; params->have_dels = true; break;
		case 'e': min_mapping_qual = optarg; have_mq = true; break;
		case 'f': params->ref_genome = optarg; params->have_ref = true; break;
		case 'h': print_help(); return CONGA_EXIT_SUCCESS;
		case 'i': params->bam_file = optarg; break;
		case 'j': min_rp_support = optarg; have_rp = true; break;
		case 'l': params->min_sv_size = atoi(optarg); break;
		case 'm': params->mappability_file = optarg; params->have_map = true; break;
		case 'n': params->sonic_info = optarg; break;
		case 'o': params->outprefix = optarg; params->have_outprefix = true; break;
		case 's': params->sonic_file = optarg; load_sonic = true; break;
		case 'u': params->dup_file = optarg; params->have_dups = true; break;
		case 'v':
			fprintf(stderr, "\n\tCONGA Version %s\n\tLast update: %s, build date: %s\n", CONGA_VERSION, CONGA_UPDATE, BUILD_DATE);
			fprintf(stderr, "\tFor more information, check https://github.com/asylvz/CONGA\n\n");
			return CONGA_EXIT_SUCCESS;
		case 'x': if (optarg) params->low_map_regions = optarg; break;
		case OPT_FIRST_CHROM: params->first_chrom = atoi(optarg); break;
		case OPT_LAST_CHROM: params->last_chrom = atoi(optarg); break;
		case OPT_DEVICE: params->device = atoi(optarg); break;
		case OPT_GPUS: params->n_gpus = std::max(1, atoi(optarg)); break;
		case OPT_COHORT: params->cohort_file = optarg; break;
		case OPT_DUMP: params->dump_intervals_chr = optarg; break;
		case OPT_DUMP_READS: params->dump_reads = true; break;
		case OPT_DUMP_MAP: params->dump_mappability_chr = optarg; break;
		default: break;
		}
	}

	if (!params->have_outprefix) {
		fprintf(stderr, "[CONGA CMDLINE ERROR] Please enter the output file name prefix using the --out option.\n");
		return CONGA_EXIT_PARAM_ERROR;
	}
	if (!params->have_ref) {
		fprintf(stderr, "[CONGA CMDLINE ERROR] Please enter reference genome file (FASTA) using the --ref option.\n");
		return CONGA_EXIT_PARAM_ERROR;
	}
	if (params->sonic_file.empty() && load_sonic) {
		fprintf(stderr, "[CONGA CMDLINE ERROR] Please enter the SONIC file (BED) using the --sonic option.\n");
		return CONGA_EXIT_PARAM_ERROR;
	}
	if (params->min_sv_size <= 0) {
		params->min_sv_size = 1000;
		fprintf(stderr, "Minimum size of an SV is set to %d\n", params->min_sv_size);
	}
	if (params->min_read_length <= 0) {
		params->min_read_length = 60;
		fprintf(stderr, "Minimum size of a read is set to %d\n", params->min_read_length);
	}
	params->c_score = have_c ? (float) atof(c_score.c_str()) : 0.5f;
	if (!have_rp) {
		params->rp_support = 10;
		params->no_sr = 1;
	} else {
		params->rp_support = atoi(min_rp_support.c_str());
		params->no_sr = 0;
	}
	params->mq_threshold = have_mq ? atoi(min_mapping_qual.c_str()) : -1;
	if (params->sonic_info.empty())
		params->sonic_info = params->ref_genome;

	get_working_directory(params);
	fprintf(stderr, "[CONGA INFO] Working directory: %s\n", params->outdir.c_str());
	return RETURN_SUCCESS;
}

} // namespace conga_host
