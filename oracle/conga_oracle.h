/*
 * conga_oracle.h -- CPU restatement of CONGA's read-depth / likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under conga_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (asylvz/CONGA) ships no tests, fixtures or golden
 * outputs for this path, and it cannot be built here (every translation unit
 * includes htslib and sonic headers; both are empty, un-vendored submodules:
 * /root/reference/.gitmodules:1-6, Makefile:5-6).  The only external anchors are
 * the known answers recorded in SURVEY.md App. D (checked in tests/test_oracle.py).
 * Each function cites the reference file:line it restates.
 *
 * Third-party semantics that are not in /root/reference:
 *   calkan/sonic (unpinned submodule) sonic_get_gc_content(chr, a, b): assumed to
 *   return the GC% (0..100) of the 100-bp window containing `a`, window index
 *   clamped to the last window of the chromosome.  It is isolated in oracle_gc().
 */
#ifndef CONGA_ORACLE_H_
#define CONGA_ORACLE_H_

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_DELETION 'D'    /* common.h:12 */
#define ORACLE_DUPLICATION 'E' /* common.h:13 */

/* Fields of the reference's `svs` record that the hot path reads or writes (svs.h:10-28). */
typedef struct oracle_sv {
	int32_t start;
	int32_t end;
	int32_t observed_rd_sv;
	float expected_rd_sv;
	double lhomo;
	double lhetero;
	double lnone;
	double likelihood_score;
	int32_t copy_number;
	int32_t rp;
	int32_t border_rp;
	int32_t pad_;
	double mappability;
} oracle_sv;

/* A split-read row after pairing (common.h:106-119), only the fields count_ReadPairs reads. */
typedef struct oracle_split_row {
	int32_t locMapLeftEnd;
	int32_t locMapRightStart;
	char svType;
} oracle_split_row;

/* GC% byte of the window holding base i (assumed sonic rule; see header comment). */
int oracle_gc(const uint8_t *gc_w, int64_t n_win, int step, int64_t i);

/* read_distribution.c:12-18 + bam_data.c:201-216.  Returns the number of counted reads. */
int64_t oracle_count_reads(int16_t *read_depth, int64_t L, const int32_t *pos, const uint8_t *mapq,
		int64_t n, int mq_threshold);

/* read_distribution.c:27-46 */
float oracle_calc_mu_per_chr(const int16_t *read_depth, int64_t L, int64_t *rd_cnt_out);

/* read_distribution.c:49-84.  rd_per_gc / window_per_gc may be NULL. */
void oracle_calc_mean_per_chr(const int16_t *read_depth, int64_t L, const uint8_t *gc_hist_w,
		int64_t n_win, int step, float expected_read_depth[101], int64_t rd_per_gc[101],
		int32_t window_per_gc[101]);

/* read_distribution.c:20-24 + svs.c:363-371 (paint loop, inclusive end, file order). */
void oracle_paint_mappability(float *mappability, int64_t L, const int32_t *start, const int32_t *end,
		const float *val, int64_t m);

/* likelihood.c:96-105 */
double oracle_lpoisson(int observed, double lambda);

/* likelihood.c:131-168: scoring given the two reduced depths. */
void oracle_score(int observed_rd, float expected_rd, char type, oracle_sv *sv);

/* likelihood.c:108-169 for one interval; mappability may be NULL (no --mappability). */
void oracle_calculate_likelihood_CNV(const int16_t *read_depth, const float *mappability, int64_t L,
		const uint8_t *gc_like_w, int64_t n_win, int step, const float expected_read_depth[101],
		char type, oracle_sv *sv);

/* likelihood.c:290-308 for an array of intervals of one type. */
void oracle_find_depths(const int16_t *read_depth, const float *mappability, int64_t L,
		const uint8_t *gc_like_w, int64_t n_win, int step, const float expected_read_depth[101],
		char type, oracle_sv *svs, int64_t count);

/* common.c:199-215 + likelihood.c:324-328 */
void oracle_sort_svs(oracle_sv *svs, int64_t count);

/* svs.c:7-240 for ONE of the two files: rows with chr == `chr` and end-start >= min_sv_size,
 * in file order.  Returns a malloc'ed array in *out (caller frees) and the count, or -1. */
int64_t oracle_load_known_SVs(const char *bed_path, const char *chr, int min_sv_size, oracle_sv **out);

/* svs.c:317-377: parse the whole mappability BED and paint rows of `chr`.  Returns rows painted or -1. */
int64_t oracle_load_mappability_regions(const char *bed_path, const char *chr, float *mappability, int64_t L);

/* split_read.c:31-466 + bam_data.c:29-154,205-210 for one chromosome (conga_oracle_sr.c).  seq: one 4-bit BAM
 * base code per byte, qual: Phred bytes, both at data_off[i].  counts: {elements, mappings, DEL rows, DUP rows}. */
int64_t oracle_split_read_rows(const char *ref, int64_t L, const int32_t *sat_start, const int32_t *sat_end,
		int64_t n_sat, int64_t n_reads, const int32_t *pos, const uint8_t *mapq, const uint16_t *flag,
		const int32_t *l_qseq, const uint64_t *data_off, const uint8_t *seq, const uint8_t *qual, int mq_threshold,
		int min_read_length, oracle_split_row **rows_out, int64_t counts[4]);

/* likelihood.c:41-94 */
void oracle_count_ReadPairs(const oracle_split_row *rows, int64_t n_rows, oracle_sv *dels, int64_t del_count,
		oracle_sv *dups, int64_t dup_count);

/* likelihood.c:172-288.  Any FILE* may be NULL when the matching BED was not given
 * (fp_svs is always written by the reference).  Counts of rows written to fp_svs are returned
 * through sv_cnt_del / sv_cnt_dup. */
void oracle_output_SVs(const char *chr_name, const oracle_sv *dels, int64_t del_count, int have_dels,
		const oracle_sv *dups, int64_t dup_count, int have_dups, int have_mappability, int no_sr,
		int rp_support, float c_score, FILE *fp_svs, FILE *fp_del, FILE *fp_dup, int *sv_cnt_del,
		int *sv_cnt_dup);

/* bam_data.c:235,242,249 */
void oracle_write_headers(FILE *fp_svs, FILE *fp_del, FILE *fp_dup);

/* Path-based wrapper for Python: appends one chromosome's rows ("a" mode; with write_headers != 0
 * the files are truncated and the headers written first).  Paths may be NULL. Returns 0 or -1. */
int oracle_output_SVs_paths(const char *chr_name, const oracle_sv *dels, int64_t del_count, int have_dels,
		const oracle_sv *dups, int64_t dup_count, int have_dups, int have_mappability, int no_sr,
		int rp_support, float c_score, const char *path_svs, const char *path_del, const char *path_dup,
		int write_headers, int *sv_cnt_del, int *sv_cnt_dup);

#ifdef __cplusplus
}
#endif
#endif /* CONGA_ORACLE_H_ */
