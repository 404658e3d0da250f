#include "likelihood.h"

namespace conga_host {

int total_dels = 0;
int total_dups = 0;

void output_SVs(const parameters *params, const chrom_svs &svs, FILE *fpSVs, FILE *fp_del, FILE *fp_dup, FILE *progress)
{
	const char *chr = svs.chr_name.c_str();
	const bool with_map = params->have_map;
	const float c = params->c_score; // a float in the reference: comparisons promote it, 1 / c is a float division
	int sv_cnt_del = 0, sv_cnt_dup = 0;

	if (params->have_dels) {
		for (size_t i = 0; i < svs.dels.size(); i++) {
			const sv_row &v = svs.dels[i];
			const conga_result &r = svs.del_res[i];
			const char *called = (r.copy_number == 2) ? "1/1" : "0/1";
			const bool is_call = r.score < c;
			const char *gt = is_call ? called : (r.score <= (1 / c) ? "N/A" : "0/0");
			if (with_map) {
				fprintf(fp_del, "%s\t%d\t%d\t%s\t%.2f\t%d\t%.2lf\t%d\t%.1f\n", chr, v.start, v.end, gt, r.score, r.border_rp,
						r.mappability, r.observed, r.expected);
				if (is_call && r.mappability > 0.5) {
					fprintf(fpSVs, "%s\t%d\t%d\tDEL\t%s\t%.2f\t%d\t%.2lf\n", chr, v.start, v.end, called, r.score, r.border_rp,
							r.mappability);
					sv_cnt_del++;
				}
			} else {
				fprintf(fp_del, "%s\t%d\t%d\t%s\t%.2f\t%d\tN/A\t%d\t%.1f\n", chr, v.start, v.end, gt, r.score, r.border_rp,
						r.observed, r.expected);
				if (is_call) { // seven columns (likelihood.c:225)
					fprintf(fpSVs, "%s\t%d\t%d\tDEL\t%s\t%.2f\t%d\n", chr, v.start, v.end, called, r.score, r.border_rp);
					sv_cnt_del++;
				}
			}
		}
	}

	if (params->have_dups) {
		for (size_t i = 0; i < svs.dups.size(); i++) {
			const sv_row &v = svs.dups[i];
			const conga_result &r = svs.dup_res[i];
			const char *called = (r.copy_number == 2) ? "1/1" : "0/1";
			if (with_map)
				fprintf(fp_dup, "%s\t%d\t%d\t%s\t%.2lf\t%d\t%.2lf\t%d\t%.1f\n", chr, v.start, v.end, called, r.score, r.rp,
						r.mappability, r.observed, r.expected);
			else
				fprintf(fp_dup, "%s\t%d\t%d\t%s\t%.2lf\t%d\tN/A\t%d\t%.1f\n", chr, v.start, v.end, called, r.score, r.rp,
						r.observed, r.expected);
			bool keep = r.score < c;
			if (!params->no_sr)
				keep = keep || (r.rp > params->rp_support);
			if (with_map)
				keep = keep && (r.mappability > 0.5);
			if (keep) {
				if (with_map)
					fprintf(fpSVs, "%s\t%d\t%d\tDUP\t%s\t%.2lf\t%d\t%.2lf\n", chr, v.start, v.end, called, r.score, r.rp,
							r.mappability);
				else
					fprintf(fpSVs, "%s\t%d\t%d\tDUP\t%s\t%.2lf\t%d\tN/A\n", chr, v.start, v.end, called, r.score, r.rp);
				sv_cnt_dup++;
			}
		}
	}

	fprintf(progress ? progress : stderr, "\nFound %d DELs - %d DUPs\n\n", sv_cnt_del, sv_cnt_dup);
	// (chromosomes are formatted side by side, bam_data.cpp: the totals are added atomically)
	__atomic_fetch_add(&total_dels, sv_cnt_del, __ATOMIC_RELAXED);
	__atomic_fetch_add(&total_dups, sv_cnt_dup, __ATOMIC_RELAXED);
}

} // namespace conga_host
