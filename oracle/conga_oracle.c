/*
 * conga_oracle.c -- serial CPU restatement of CONGA's read-depth / likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see conga_oracle.h).  PARITY UNPINNED: the reference has no
 * tests or fixtures for this path and cannot be built in this image (htslib + sonic absent).
 *
 * Written from the arithmetic spec in SURVEY.md App. A; every function cites the reference
 * lines it restates.  Loops are deliberately per base, like the reference's, because this file
 * is also the timed CPU baseline ("port") of bench.py.
 *
 * Documented deviations (all are undefined behaviour in the reference):
 *   - records with pos outside [0, L) are skipped instead of written out of bounds;
 *   - interval / mappability positions >= L read depth 0 / are not painted;
 *   - BED rows with fewer than three columns are skipped instead of crashing in atoi(NULL).
 */
#include "conga_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ROW_DELIMS " \t\r\n" /* svs.h:7 */
#define LINE_CHUNK 512       /* svs.c:11 */
#define WRONGMAP_WINDOW 100      /* likelihood.h:17 */
#define WRONGMAP_WINDOW_DEL 5000 /* likelihood.h:18 */

int oracle_gc(const uint8_t *gc_w, int64_t n_win, int step, int64_t i)
{
	int64_t w = i / step;
	if (w >= n_win)
		w = n_win - 1;
	if (w < 0)
		w = 0;
	return (int) gc_w[w];
}

/* read_distribution.c:12-18 (alloc + zero) and the counting loop of bam_data.c:201-216:
 * `if (core.qual > mq_threshold) read_depth[core.pos]++` on a short array. */
int64_t oracle_count_reads(int16_t *read_depth, int64_t L, const int32_t *pos, const uint8_t *mapq,
		int64_t n, int mq_threshold)
{
	int64_t i, counted = 0;

	memset(read_depth, 0, (size_t) L * sizeof(int16_t));
	for (i = 0; i < n; i++) {
		if ((int) mapq[i] > mq_threshold) {
			int64_t p = pos[i];
			if (p < 0 || p >= L)
				continue;
			/* short++ : promote, add, convert back (gcc wraps modulo 2^16) */
			read_depth[p] = (int16_t) (read_depth[p] + 1);
			counted++;
		}
	}
	return counted;
}

/* read_distribution.c:27-46: mean = (double) sum / (double) L, stored in a float. */
float oracle_calc_mu_per_chr(const int16_t *read_depth, int64_t L, int64_t *rd_cnt_out)
{
	int64_t i;
	long rd_cnt = 0, window_total = 0;
	float mean;

	for (i = 0; i < L; i++) {
		rd_cnt += (long) read_depth[i];
		window_total++;
	}
	mean = (double) rd_cnt / (double) window_total;
	if (rd_cnt_out)
		*rd_cnt_out = rd_cnt;
	return mean;
}

/* read_distribution.c:49-84 */
void oracle_calc_mean_per_chr(const int16_t *read_depth, int64_t L, const uint8_t *gc_hist_w,
		int64_t n_win, int step, float expected_read_depth[101], int64_t rd_per_gc_out[101],
		int32_t window_per_gc_out[101])
{
	long rd_per_gc[101];
	int window_per_gc[101];
	int64_t i;
	int g;

	for (g = 0; g < 101; g++) {
		rd_per_gc[g] = 0;
		window_per_gc[g] = 0;
	}

	/* :63-73 -- the window handed to sonic is (i, min(i + 100, L)); under the assumed sonic
	 * rule only its start selects the GC byte. */
	for (i = 0; i < L; i++) {
		g = oracle_gc(gc_hist_w, n_win, step, i);
		rd_per_gc[g] += (long) read_depth[i];
		window_per_gc[g]++;
	}

	/* :75-83 */
	expected_read_depth[0] = 0.0f;
	for (g = 1; g < 101; g++) {
		float e = (float) rd_per_gc[g] / (window_per_gc[g]);
		if (isnan(e) || isinf(e))
			e = 0;
		expected_read_depth[g] = e;
	}

	for (g = 0; g < 101; g++) {
		if (rd_per_gc_out)
			rd_per_gc_out[g] = rd_per_gc[g];
		if (window_per_gc_out)
			window_per_gc_out[g] = window_per_gc[g];
	}
}

/* read_distribution.c:20-24 (zero) + the paint loop svs.c:363-371: rows in file order,
 * `for (i = start; i <= end; i++) mappability[i] = value` -- END INCLUSIVE, later rows win. */
void oracle_paint_mappability(float *mappability, int64_t L, const int32_t *start, const int32_t *end,
		const float *val, int64_t m)
{
	int64_t r, i;

	memset(mappability, 0, (size_t) L * sizeof(float));
	for (r = 0; r < m; r++) {
		int64_t s = start[r], e = end[r];
		if (s < 0)
			s = 0;
		if (e > L - 1)
			e = L - 1;
		for (i = s; i <= e; i++)
			mappability[i] = val[r];
	}
}

/* likelihood.c:96-105 */
double oracle_lpoisson(int observed, double lambda)
{
	if (lambda == 0.0)
		lambda = 0.01;
	return observed * log(lambda) - lambda - lgamma(observed + 1);
}

/* common.c:262-268 -- the reference's max() takes and returns int, so both log-likelihoods
 * are truncated toward zero before the comparison (likelihood.c:138,160). */
static int int_max(int x, int y)
{
	return (x < y) ? y : x;
}

/* likelihood.c:131-168 */
void oracle_score(int observed_rd, float expected_rd, char type, oracle_sv *sv)
{
	if (type == ORACLE_DELETION) {
		sv->lhomo = oracle_lpoisson(observed_rd, 0.0);
		sv->lhetero = oracle_lpoisson(observed_rd, 0.5 * expected_rd); /* double * float */
		sv->lnone = oracle_lpoisson(observed_rd, expected_rd);
		sv->likelihood_score = int_max(sv->lhomo, sv->lhetero) / sv->lnone;
		sv->observed_rd_sv = observed_rd;
		sv->expected_rd_sv = expected_rd;
		/* :146 int < float : the int is converted to float */
		sv->copy_number = (observed_rd < (expected_rd / 4)) ? 2 : 1;
	} else if (type == ORACLE_DUPLICATION) {
		sv->lhomo = oracle_lpoisson(observed_rd, 2 * expected_rd); /* int * float, in float */
		sv->lhetero = oracle_lpoisson(observed_rd, 1.5 * expected_rd);
		sv->lnone = oracle_lpoisson(observed_rd, expected_rd);
		sv->likelihood_score = int_max(sv->lhomo, sv->lhetero) / sv->lnone;
		sv->observed_rd_sv = observed_rd;
		sv->expected_rd_sv = expected_rd;
		sv->copy_number = (sv->lhomo > sv->lhetero) ? 2 : 1;
	}
}

/* likelihood.c:108-169 */
void oracle_calculate_likelihood_CNV(const int16_t *read_depth, const float *mappability, int64_t L,
		const uint8_t *gc_like_w, int64_t n_win, int step, const float expected_read_depth[101],
		char type, oracle_sv *sv)
{
	int gc_val;
	int64_t i;
	float expected_rd = 0;
	int observed_rd = 0;
	double mappability_score = 0;

	/* :115-124 -- serial, single precision for expected_rd */
	for (i = sv->start; i < sv->end; i++) {
		gc_val = oracle_gc(gc_like_w, n_win, step, i);
		expected_rd += expected_read_depth[gc_val];
		if (i >= 0 && i < L) {
			observed_rd += read_depth[i];
			if (mappability != NULL)
				mappability_score += mappability[i];
		}
	}

	/* :127-128 */
	if (mappability != NULL)
		sv->mappability = mappability_score / (double) (sv->end - sv->start);

	oracle_score(observed_rd, expected_rd, type, sv);
}

/* likelihood.c:290-308 */
void oracle_find_depths(const int16_t *read_depth, const float *mappability, int64_t L,
		const uint8_t *gc_like_w, int64_t n_win, int step, const float expected_read_depth[101],
		char type, oracle_sv *svs, int64_t count)
{
	int64_t c;
	for (c = 0; c < count; c++)
		oracle_calculate_likelihood_CNV(read_depth, mappability, L, gc_like_w, n_win, step,
				expected_read_depth, type, &svs[c]);
}

/* common.c:199-215 */
static int cmp_start_then_end(const void *a, const void *b)
{
	const oracle_sv *x = (const oracle_sv *) a, *y = (const oracle_sv *) b;
	if (x->start > y->start)
		return 1;
	if (x->start == y->start)
		return x->end - y->end;
	return -1;
}

/* likelihood.c:324-328 */
void oracle_sort_svs(oracle_sv *svs, int64_t count)
{
	qsort(svs, (size_t) count, sizeof(oracle_sv), cmp_start_then_end);
}

static int chunk_is_blank(const char *line)
{
	size_t k, len = strlen(line);
	for (k = 0; k < len; k++)
		if (!isspace((unsigned char) line[k]))
			return 0;
	return 1;
}

/* svs.c:7-240.  The reference reads each file twice (count, then fill) in 512-byte fgets chunks
 * and keeps a row when strcmp(chr) == 0 and end - start >= min_sv_size; one growing pass gives
 * the same rows in the same order. */
int64_t oracle_load_known_SVs(const char *bed_path, const char *chr, int min_sv_size, oracle_sv **out)
{
	FILE *f = fopen(bed_path, "r");
	char line[LINE_CHUNK];
	int64_t n = 0, cap = 1024;
	oracle_sv *arr;

	*out = NULL;
	if (!f)
		return -1;
	arr = (oracle_sv *) malloc((size_t) cap * sizeof(oracle_sv));
	while (fgets(line, LINE_CHUNK, f) != NULL) {
		char *chr_name, *tok_s, *tok_e;
		int start_sv, end_sv;

		if (chunk_is_blank(line))
			continue;
		chr_name = strtok(line, ROW_DELIMS);
		tok_s = strtok(NULL, ROW_DELIMS);
		tok_e = strtok(NULL, ROW_DELIMS);
		if (!chr_name || !tok_s || !tok_e)
			continue;
		start_sv = atoi(tok_s);
		end_sv = atoi(tok_e);
		if (strcmp(chr_name, chr) != 0 || (end_sv - start_sv) < min_sv_size)
			continue;
		if (n == cap) {
			cap *= 2;
			arr = (oracle_sv *) realloc(arr, (size_t) cap * sizeof(oracle_sv));
		}
		memset(&arr[n], 0, sizeof(oracle_sv));
		arr[n].start = start_sv;
		arr[n].end = end_sv;
		n++;
	}
	fclose(f);
	*out = arr;
	return n;
}

/* svs.c:317-377: the whole file is parsed for every chromosome; rows of other chromosomes are
 * dropped after the first token; value = atof() narrowed to float. */
int64_t oracle_load_mappability_regions(const char *bed_path, const char *chr, float *mappability, int64_t L)
{
	FILE *f = fopen(bed_path, "r");
	char line[LINE_CHUNK];
	int64_t painted = 0, i;

	if (!f)
		return -1;
	memset(mappability, 0, (size_t) L * sizeof(float));
	while (fgets(line, LINE_CHUNK, f) != NULL) {
		char *chr_name, *tok_s, *tok_e, *tok_v;
		int64_t s, e;
		float v;

		if (chunk_is_blank(line))
			continue;
		chr_name = strtok(line, ROW_DELIMS);
		if (!chr_name || strcmp(chr_name, chr) != 0)
			continue;
		tok_s = strtok(NULL, ROW_DELIMS);
		tok_e = strtok(NULL, ROW_DELIMS);
		tok_v = strtok(NULL, ROW_DELIMS);
		if (!tok_s || !tok_e || !tok_v)
			continue;
		s = atoi(tok_s);
		e = atoi(tok_e);
		v = atof(tok_v);
		if (s < 0)
			s = 0;
		if (e > L - 1)
			e = L - 1;
		for (i = s; i <= e; i++)
			mappability[i] = v;
		painted++;
	}
	fclose(f);
	return painted;
}

/* likelihood.c:41-94 */
void oracle_count_ReadPairs(const oracle_split_row *rows, int64_t n_rows, oracle_sv *dels, int64_t del_count,
		oracle_sv *dups, int64_t dup_count)
{
	int64_t r, i;

	for (r = 0; r < n_rows; r++) {
		const oracle_split_row *row = &rows[r];
		if (row->svType == ORACLE_DUPLICATION) {
			for (i = 0; i < dup_count; i++) {
				int lo = dups[i].start - WRONGMAP_WINDOW_DEL, hi = dups[i].end + WRONGMAP_WINDOW_DEL;
				if (row->locMapLeftEnd >= lo && row->locMapLeftEnd <= hi && row->locMapRightStart <= hi
						&& row->locMapRightStart >= lo)
					dups[i].rp++;
			}
		} else if (row->svType == ORACLE_DELETION) {
			for (i = 0; i < del_count; i++) {
				int s = dels[i].start, e = dels[i].end;
				if (row->locMapLeftEnd <= s + WRONGMAP_WINDOW && row->locMapLeftEnd >= s - WRONGMAP_WINDOW_DEL
						&& row->locMapRightStart >= e - WRONGMAP_WINDOW
						&& row->locMapRightStart <= e + WRONGMAP_WINDOW_DEL)
					dels[i].border_rp++;
			}
		}
	}
}

/* bam_data.c:235,242,249 */
void oracle_write_headers(FILE *fp_svs, FILE *fp_del, FILE *fp_dup)
{
	static const char *cols = "#CHR\tSTART_SV\tEND_SV\tCOPY_NUMBER\tLIKELIHOOD\tREAD_PAIR\tMAPPABILITY\tOBSERVED_READS\tEXPECTED_READS\n";
	if (fp_svs)
		fprintf(fp_svs, "#CHR\tSTART_SV\tEND_SV\tSV_TYPE\tCOPY_NUMBER\tLIKELIHOOD\tREAD_PAIR\tMAPPABILITY\n");
	if (fp_del)
		fprintf(fp_del, "%s", cols);
	if (fp_dup)
		fprintf(fp_dup, "%s", cols);
}

/* likelihood.c:172-288 */
void oracle_output_SVs(const char *chr_name, const oracle_sv *dels, int64_t del_count, int have_dels,
		const oracle_sv *dups, int64_t dup_count, int have_dups, int have_mappability, int no_sr,
		int rp_support, float c_score, FILE *fp_svs, FILE *fp_del, FILE *fp_dup, int *sv_cnt_del_out,
		int *sv_cnt_dup_out)
{
	int64_t c;
	int sv_cnt_del = 0, sv_cnt_dup = 0;

	if (have_dels) {
		for (c = 0; c < del_count; c++) {
			const oracle_sv *v = &dels[c];
			const char *called = (v->copy_number == 2) ? "1/1" : "0/1";
			const char *gt;

			/* :184-195 / :207-220 -- c_score is a float; 1 / c_score is a float division */
			if (v->likelihood_score < c_score)
				gt = called;
			else if (v->likelihood_score <= (1 / c_score))
				gt = "N/A";
			else
				gt = "0/0";

			if (have_mappability) {
				fprintf(fp_del, "%s\t%d\t%d\t%s\t%.2f\t%d\t%.2lf\t%d\t%.1f\n", chr_name, v->start, v->end, gt,
						v->likelihood_score, v->border_rp, v->mappability, v->observed_rd_sv,
						v->expected_rd_sv);
				if (v->likelihood_score < c_score && v->mappability > 0.5) { /* :198-202 */
					fprintf(fp_svs, "%s\t%d\t%d\tDEL\t%s\t%.2f\t%d\t%.2lf\n", chr_name, v->start, v->end,
							called, v->likelihood_score, v->border_rp, v->mappability);
					sv_cnt_del++;
				}
			} else {
				fprintf(fp_del, "%s\t%d\t%d\t%s\t%.2f\t%d\tN/A\t%d\t%.1f\n", chr_name, v->start, v->end, gt,
						v->likelihood_score, v->border_rp, v->observed_rd_sv, v->expected_rd_sv);
				if (v->likelihood_score < c_score) { /* :223-227, seven columns */
					fprintf(fp_svs, "%s\t%d\t%d\tDEL\t%s\t%.2f\t%d\n", chr_name, v->start, v->end, called,
							v->likelihood_score, v->border_rp);
					sv_cnt_del++;
				}
			}
		}
	}

	if (have_dups) {
		for (c = 0; c < dup_count; c++) {
			const oracle_sv *v = &dups[c];
			const char *called = (v->copy_number == 2) ? "1/1" : "0/1";
			int keep;

			/* :237-239 */
			if (have_mappability)
				fprintf(fp_dup, "%s\t%d\t%d\t%s\t%.2lf\t%d\t%.2lf\t%d\t%.1f\n", chr_name, v->start, v->end,
						called, v->likelihood_score, v->rp, v->mappability, v->observed_rd_sv,
						v->expected_rd_sv);
			else
				fprintf(fp_dup, "%s\t%d\t%d\t%s\t%.2lf\t%d\tN/A\t%d\t%.1f\n", chr_name, v->start, v->end,
						called, v->likelihood_score, v->rp, v->observed_rd_sv, v->expected_rd_sv);

			/* :241-279 */
			if (!no_sr)
				keep = (v->rp > rp_support) || (v->likelihood_score < c_score);
			else
				keep = (v->likelihood_score < c_score);
			if (have_mappability)
				keep = keep && (v->mappability > 0.5);
			if (keep) {
				if (have_mappability)
					fprintf(fp_svs, "%s\t%d\t%d\tDUP\t%s\t%.2lf\t%d\t%.2lf\n", chr_name, v->start, v->end,
							called, v->likelihood_score, v->rp, v->mappability);
				else
					fprintf(fp_svs, "%s\t%d\t%d\tDUP\t%s\t%.2lf\t%d\tN/A\n", chr_name, v->start, v->end,
							called, v->likelihood_score, v->rp);
				sv_cnt_dup++;
			}
		}
	}

	if (sv_cnt_del_out)
		*sv_cnt_del_out = sv_cnt_del;
	if (sv_cnt_dup_out)
		*sv_cnt_dup_out = sv_cnt_dup;
}

int oracle_output_SVs_paths(const char *chr_name, const oracle_sv *dels, int64_t del_count, int have_dels,
		const oracle_sv *dups, int64_t dup_count, int have_dups, int have_mappability, int no_sr,
		int rp_support, float c_score, const char *path_svs, const char *path_del, const char *path_dup,
		int write_headers, int *sv_cnt_del, int *sv_cnt_dup)
{
	const char *mode = write_headers ? "w" : "a";
	FILE *fs = path_svs ? fopen(path_svs, mode) : NULL;
	FILE *fd = (path_del && have_dels) ? fopen(path_del, mode) : NULL;
	FILE *fu = (path_dup && have_dups) ? fopen(path_dup, mode) : NULL;

	if (!fs || (have_dels && !fd) || (have_dups && !fu)) {
		if (fs)
			fclose(fs);
		if (fd)
			fclose(fd);
		if (fu)
			fclose(fu);
		return -1;
	}
	if (write_headers)
		oracle_write_headers(fs, fd, fu);
	oracle_output_SVs(chr_name, dels, del_count, have_dels, dups, dup_count, have_dups, have_mappability,
			no_sr, rp_support, c_score, fs, fd, fu, sv_cnt_del, sv_cnt_dup);
	fclose(fs);
	if (fd)
		fclose(fd);
	if (fu)
		fclose(fu);
	return 0;
}
