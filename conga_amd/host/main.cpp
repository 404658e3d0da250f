// main.cpp -- the `conga` executable; mirrors main (svdepth.c:16-74).
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/conga_hip.h"
#include "knobs.h"
#include "annotation.h"
#include "bam_data.h"
#include "cmdline.h"
#include "likelihood.h"
#include "reads.h"
#include "svs.h"

using namespace conga_host;

int main(int argc, char **argv)
{
	// CONGA_T0_NS (tools/e2e_quick.sh): the caller's clock just before the process was started -- what of a run's wall time lies
	// in front of main and behind _exit is then visible
	auto since_t0 = [](const char *what) {
		if (!knobs().t0_ns || !knobs().timing)
			return;
		struct timespec ts;
		clock_gettime(CLOCK_REALTIME, &ts);
		const long long now = (long long) ts.tv_sec * 1000000000ll + ts.tv_nsec;
		fprintf(stderr, "[timing] %s: %.1f ms after the caller's clock\n", what, (double) (now - knobs().t0_ns) * 1e-6);
	};
	since_t0("main entered");
	time_t rawtime;
	time(&rawtime);
	struct tm *timeinfo = localtime(&rawtime);

	// simple log in the current directory, as the reference keeps (svdepth.c:30-31)
	logFile = fopen("conga.log", "w");
	if (!logFile) {
		fprintf(stderr, "\n[CONGA INPUT ERROR] Unable to open file conga.log in write mode.\nInvoke parameter -h for help.\n");
		return CONGA_EXIT_COMMON;
	}
	fprintf(logFile, "#CreationDate=%d.%d.%d\n\n", timeinfo->tm_year + 1900, timeinfo->tm_mon + 1, timeinfo->tm_mday);

	parameters params;
	const int rv = parse_cmd_line(argc, argv, &params);
	if (rv == CONGA_EXIT_PARAM_ERROR)
		return EXIT_FAILURE;
	if (rv == CONGA_EXIT_SUCCESS)
		return EXIT_SUCCESS;

	// print_params (common.c:86-100)
	printf("\n");
	printf("%-30s%s\n", "BAM input:", params.bam_file.c_str());
	fprintf(logFile, "%-30s%s\n", "BAM input:", params.bam_file.c_str());
	printf("%-30s%s\n", "Reference genome:", params.ref_genome.c_str());
	printf("%-30s%s\n", "SONIC file:", params.sonic_file.c_str());
	fprintf(logFile, "%-30s%s\n", "Reference genome:", params.ref_genome.c_str());
	fprintf(logFile, "%-30s%s\n", "SONIC file:", params.sonic_file.c_str());
	fprintf(logFile, "%-30s%d\n", "First chrom:", params.first_chrom);
	fprintf(logFile, "%-30s%d\n", "Last chrom:", params.last_chrom);

	// --dump-intervals CHR: the rows load_known_SVs + qsort would hand to the engine (no GPU needed)
	if (!params.dump_intervals_chr.empty()) {
		for (int t = 0; t < 2; t++) {
			const bool have = t == 0 ? params.have_dels : params.have_dups;
			if (!have)
				continue;
			bed_index bed;
			const std::string &path = t == 0 ? params.del_file : params.dup_file;
			if (!load_bed(path, false, &bed)) {
				fprintf(stderr, "\n[CONGA INPUT ERROR] Unable to open file %s in read mode.\nInvoke parameter -h for help.\n", path.c_str());
				return CONGA_EXIT_COMMON;
			}
			for (const sv_row &r : known_SVs_for(bed, params.dump_intervals_chr, params.min_sv_size))
				printf("%s\t%s\t%d\t%d\n", t == 0 ? "DEL" : "DUP", params.dump_intervals_chr.c_str(), r.start, r.end);
		}
		fclose(logFile);
		return EXIT_SUCCESS;
	}

	if (!params.dump_mappability_chr.empty()) {
		bed_index bed;
		if (!params.have_map || !load_bed(params.mappability_file, true, &bed)) {
			fprintf(stderr, "\n[CONGA INPUT ERROR] Unable to open file %s in read mode.\nInvoke parameter -h for help.\n", params.mappability_file.c_str());
			return CONGA_EXIT_COMMON;
		}
		const auto &rows = bed.rows[params.dump_mappability_chr];
		const auto &vals = bed.values[params.dump_mappability_chr];
		for (size_t i = 0; i < rows.size(); i++)
			printf("MAP\t%d\t%d\t%a\n", rows[i].start, rows[i].end, (double) vals[i]);
		fclose(logFile);
		return EXIT_SUCCESS;
	}

	// Bringing the HIP runtime up is the better part of a tenth of a second and needs nothing from the inputs: it starts now,
	// on a thread of its own, while the annotation, the BED files and the BAM index are read.  (exit() on an input error
	// waits for it: leaving while the runtime initialises is not safe.)
	if (!params.dump_reads) {
		static std::thread warm;
		warm = std::thread([] { (void) conga_device_count(); });
		atexit([] {
			if (warm.joinable())
				warm.join();
		});
	}
	std::string err;
	std::unique_ptr<sonic> this_sonic(sonic_load(params.sonic_file, &err));
	if (!this_sonic) {
		fprintf(stderr, "\n[CONGA INPUT ERROR] %s\nInvoke parameter -h for help.\n", err.c_str());
		return CONGA_EXIT_COMMON;
	}
	if (params.last_chrom < params.first_chrom)
		params.last_chrom = this_sonic->number_of_chromosomes - 1; // svdepth.c:49-50

	// --dump-reads: what count_reads_bam would be handed, per chromosome (no GPU needed)
	if (params.dump_reads) {
		std::unique_ptr<read_source> src(open_reads(params.bam_file, &err));
		if (!src) {
			fprintf(stderr, "\n%s\nInvoke parameter -h for help.\n", err.c_str());
			return CONGA_EXIT_COMMON;
		}
		printf("sample\t%s\n", src->sample_name().c_str());
		printf("index\t%s\n", src->index_path().empty() ? "none" : src->index_path().c_str());
		for (int c = params.first_chrom; c <= params.last_chrom && c < this_sonic->number_of_chromosomes; c++) {
			const std::string &name = this_sonic->chromosome_names[c];
			if (name.find('X') != std::string::npos || name.find('Y') != std::string::npos)
				continue;
			const int tid = find_chr_index_bam(name, *src);
			if (tid < 0) {
				printf("%s\tmissing\n", name.c_str());
				continue;
			}
			long long n = 0, sum_pos = 0, sum_mapq = 0;
			{
				std::vector<int32_t> all_pos;
				std::vector<uint8_t> all_mapq;
				std::string perr;
				if (src->read_all(tid, this_sonic->chromosome_lengths[c], usable_cpus(), &all_pos, &all_mapq, &perr)) {
					for (size_t i = 0; i < all_pos.size(); i++) {
						sum_pos += all_pos[i];
						sum_mapq += all_mapq[i];
					}
					printf("%s\t%lld\t%lld\t%lld\n", name.c_str(), (long long) all_pos.size(), sum_pos, sum_mapq);
					continue;
				}
				if (!perr.empty()) {
					fprintf(stderr, "%s\n", perr.c_str());
					return CONGA_EXIT_COMMON;
				}
			}
			if (!src->begin(tid, this_sonic->chromosome_lengths[c], &err)) {
				fprintf(stderr, "%s\n", err.c_str());
				return CONGA_EXIT_COMMON;
			}
			for (;;) {
				read_batch b;
				if (!src->next(1 << 16, &b, &err)) {
					fprintf(stderr, "%s\n", err.c_str());
					return CONGA_EXIT_COMMON;
				}
				for (size_t i = 0; i < b.n; i++) {
					sum_pos += b.pos[i];
					sum_mapq += b.mapq[i];
				}
				n += (long long) b.n;
				if (b.n < (size_t) (1 << 16))
					break;
			}
			printf("%s\t%lld\t%lld\t%lld\n", name.c_str(), n, sum_pos, sum_mapq);
		}
		fclose(logFile);
		return EXIT_SUCCESS;
	}

	since_t0("inputs read");
	const int rc = params.cohort_file.empty() ? read_bam(&params, this_sonic.get()) : read_bam_cohort(&params, this_sonic.get());
	if (rc != 0)
		return rc;

	char username[1000] = "";
	if (getlogin_r(username, sizeof username - 1) != 0)
		username[0] = '\0';
	fprintf(stderr, "\nThank you %s. I found %d DELs and %d DUPs. Hope to see you again...\n", username, total_dels, total_dups);
	fclose(logFile);
	if (!knobs().clean_exit) {
		// outputs are written and closed: leave without the HIP runtime's and the contexts' teardown (bam_data.cpp)
		since_t0("leaving");
		fflush(nullptr);
		_exit(EXIT_SUCCESS);
	}
	return EXIT_SUCCESS;
}
