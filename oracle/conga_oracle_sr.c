/*
 * conga_oracle_sr.c -- serial CPU restatement of CONGA's split-read evidence path (--rp with --dups).
 *
 * TEST INFRASTRUCTURE ONLY (see conga_oracle.h).  PARITY UNPINNED: the reference has no tests or fixtures for
 * this path and cannot be built in this image (htslib + sonic absent).  Restated from SURVEY.md App. A.8:
 *   k-mer index            split_read.c:31-73,357-466
 *   half-read mapping      split_read.c:75-354
 *   pairing                bam_data.c:29-154
 *   support counting       likelihood.c:41-94  (oracle_count_ReadPairs in conga_oracle.c)
 *
 * Documented deviations (undefined behaviour in the reference):
 *   - reference bases at or beyond the chromosome end compare as mismatches (the reference reads past its buffer);
 *   - base codes other than 1/2/4/8/15 decode to 'N' (the reference leaves the char uninitialised);
 *   - a half read and its mapping at the same position yield no row (uninitialised pos1_/pos2_ in the reference);
 *   - reads longer than 1022 bases are skipped (the reference overflows char str[512]);
 *   - the k-mer index holds every position whose 10-mer is ACGT-only (the reference's scan can run past the
 *     sequence end while recovering from a non-ACGT base).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "conga_oracle.h"

#define HASHKMERLEN 10     /* split_read.h:17 */
#define MAX_SR_HIT 50000   /* split_read.h:12 */
#define MAX_MAPPING 100    /* split_read.h:13 */
#define SR_LOOKAHEAD 100000 /* split_read.c:6 */
#define SOFTCLIP_WRONGMAP_WINDOW 50 /* bam_data.h:14 */
#define HASH_SIZE (1 << (2 * HASHKMERLEN))

/* BAM flag bits tested by is_proper (common.c:317-323) */
#define F_SECONDARY 0x100
#define F_QCFAIL 0x200
#define F_DUP 0x400
#define F_SUPPLEMENTARY 0x800

typedef struct kmer_index {
	int *count;   /* hash_table_count */
	int **bucket; /* hash_table_array */
} kmer_index;

static int is_dna(char c)
{
	return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}

/* split_read.c:37-49: two bits per base, (c & 6) >> 1  (A 0, C 1, T 2, G 3) */
static unsigned hash_kmer(const char *s)
{
	unsigned v = 0;
	int i;
	for (i = 0; i < HASHKMERLEN; i++)
		v = (v << 2) | ((unsigned) (s[i] & 0x6) >> 1);
	return v;
}

static int kmer_valid(const char *s, int64_t avail)
{
	int i;
	if (avail < HASHKMERLEN)
		return 0;
	for (i = 0; i < HASHKMERLEN; i++)
		if (!is_dna(s[i]))
			return 0;
	return 1;
}

/* split_read.c:357-466: count pass, drop empty / >= MAX_SR_HIT buckets, fill pass in increasing position */
static kmer_index *build_index(const char *ref, int64_t len)
{
	kmer_index *ix = (kmer_index *) calloc(1, sizeof *ix);
	int64_t i;
	int *iter;
	ix->count = (int *) calloc(HASH_SIZE, sizeof(int));
	ix->bucket = (int **) calloc(HASH_SIZE, sizeof(int *));
	iter = (int *) calloc(HASH_SIZE, sizeof(int));
	for (i = 0; i + HASHKMERLEN <= len; i++)
		if (kmer_valid(ref + i, len - i))
			ix->count[hash_kmer(ref + i)]++;
	for (i = 0; i < HASH_SIZE; i++) {
		if (ix->count[i] != 0 && ix->count[i] < MAX_SR_HIT)
			ix->bucket[i] = (int *) malloc((size_t) ix->count[i] * sizeof(int));
		else
			ix->count[i] = 0;
	}
	for (i = 0; i + HASHKMERLEN <= len; i++)
		if (kmer_valid(ref + i, len - i)) {
			const unsigned h = hash_kmer(ref + i);
			if (ix->count[h] != 0)
				ix->bucket[h][iter[h]++] = (int) i;
		}
	free(iter);
	return ix;
}

static void free_index(kmer_index *ix)
{
	int i;
	for (i = 0; i < HASH_SIZE; i++)
		free(ix->bucket[i]);
	free(ix->bucket);
	free(ix->count);
	free(ix);
}

/* common.c:278-287, with positions >= len counted as mismatches */
static int hamming(const char *ref, int64_t len, int64_t at, const char *s, int n)
{
	int d = 0, i;
	for (i = 0; i < n; i++)
		if (at + i >= len || ref[at + i] != s[i])
			d++;
	return d;
}

/* split_read.c:75-204.  Returns the number of mappings written (0 when there are none or >= MAX_MAPPING). */
static int almost_perfect_match(const kmer_index *ix, const char *ref, int64_t len, const char *str, int pos,
		int *posMap, char *orient, int *mapq)
{
	const int n = (int) strlen(str);
	const int dist_max = (int) (0.05 * n);
	int size = 0, h, c;
	char rev[1024];

	if (n < HASHKMERLEN)
		return 0;
	if (kmer_valid(str, n)) {
		const unsigned idx = hash_kmer(str);
		for (h = 0; h < ix->count[idx]; h++) {
			const int p = ix->bucket[idx][h];
			if (abs(p - pos) < SR_LOOKAHEAD && hamming(ref, len, p, str, n) <= dist_max) {
				if (size < MAX_MAPPING) {
					posMap[size] = p;
					orient[size] = 'F';
				}
				size++;
			}
		}
	}
	if (size < MAX_MAPPING) {
		for (c = 0; c < n; c++) {
			const char b = str[c];
			rev[n - c - 1] = b == 'A' ? 'T' : b == 'T' ? 'A' : b == 'G' ? 'C' : b == 'C' ? 'G' : 'N';
		}
		rev[n] = '\0';
		if (kmer_valid(rev, n)) {
			const unsigned idx = hash_kmer(rev);
			for (h = 0; h < ix->count[idx]; h++) {
				const int p = ix->bucket[idx][h];
				if (abs(p - pos) < SR_LOOKAHEAD && hamming(ref, len, p, rev, n) <= dist_max) {
					if (size < MAX_MAPPING) {
						posMap[size] = p;
						orient[size] = 'R';
					}
					size++;
				}
				if (size > MAX_MAPPING)
					break;
			}
		}
	}
	if (size > 0 && size < MAX_MAPPING) {
		*mapq = 60 / size;
		return size;
	}
	return 0;
}

static int is_satellite(const int32_t *ss, const int32_t *se, int64_t n, int64_t a, int64_t b)
{
	int64_t i;
	for (i = 0; i < n; i++)
		if ((int64_t) ss[i] < b && (int64_t) se[i] > a)
			return 1;
	return 0;
}

static char decode_base(int code)
{
	switch (code) {
	case 1: return 'A';
	case 2: return 'C';
	case 4: return 'G';
	case 8: return 'T';
	default: return 'N';
	}
}

typedef struct row_vec {
	oracle_split_row *v;
	int64_t n, cap;
} row_vec;

static void push_row(row_vec *r, int left_end, int right_start, char type)
{
	if (r->n == r->cap) {
		r->cap = r->cap ? r->cap * 2 : 1024;
		r->v = (oracle_split_row *) realloc(r->v, (size_t) r->cap * sizeof(oracle_split_row));
	}
	memset(&r->v[r->n], 0, sizeof(oracle_split_row));
	r->v[r->n].locMapLeftEnd = left_end;
	r->v[r->n].locMapRightStart = right_start;
	r->v[r->n].svType = type;
	r->n++;
}

/* read_SplitReads + determine_SvType for one (element, mapping) pair (bam_data.c:29-154) */
static void pair_one(row_vec *rows, int64_t L, const int32_t *ss, const int32_t *se, int64_t n_sat, int pos, int qual,
		int read_length, int is_read2, int posMap, char orient, int mapq, int mq_threshold)
{
	int pos1_2, pos2_1, lengthSplit, lengthRead;
	char type;

	if (is_satellite(ss, se, n_sat, pos, (int64_t) pos + 1) + is_satellite(ss, se, n_sat, posMap, (int64_t) posMap + 1) != 0)
		return;
	if (!(qual > mq_threshold && mapq > mq_threshold && pos > 0 && posMap > 0 && pos < L && posMap < L))
		return;
	lengthSplit = read_length / 2;
	lengthRead = read_length - lengthSplit;
	if (pos < posMap) {
		pos1_2 = pos + lengthRead;
		pos2_1 = posMap;
	} else if (posMap < pos) {
		pos1_2 = posMap + lengthSplit;
		pos2_1 = pos;
	} else
		return;
	if (pos1_2 >= pos2_1)
		return;
	if (orient != 'F')
		return; /* the element itself is always FORWARD (split_read.c:232,301) */
	if ((pos < posMap && !is_read2) || (pos > posMap && is_read2))
		type = ORACLE_DELETION;
	else
		type = ORACLE_DUPLICATION;
	push_row(rows, pos1_2 - SOFTCLIP_WRONGMAP_WINDOW, pos2_1 + SOFTCLIP_WRONGMAP_WINDOW, type);
}

/*
 * The whole split-read evidence pass for one chromosome: gate of bam_data.c:205-210, find_split_reads
 * (split_read.c:206-354), mapping, pairing.  Reads are given as the fields of bam1_t the path touches;
 * seq holds one 4-bit BAM base code per byte, qual the Phred bytes, both at data_off[i].
 * Returns the number of rows (malloc'ed array in *rows_out, caller frees) and, through counts[4]:
 * {elements created (split_read_count), mappings emitted, DEL rows, DUP rows}.
 */
int64_t oracle_split_read_rows(const char *ref, int64_t L, const int32_t *sat_start, const int32_t *sat_end,
		int64_t n_sat, int64_t n_reads, const int32_t *pos, const uint8_t *mapq, const uint16_t *flag,
		const int32_t *l_qseq, const uint64_t *data_off, const uint8_t *seq, const uint8_t *qual, int mq_threshold,
		int min_read_length, oracle_split_row **rows_out, int64_t counts[4])
{
	kmer_index *ix = build_index(ref, L);
	row_vec rows = {0, 0, 0};
	int64_t r, n_elem = 0, n_map = 0;
	int posMap[MAX_MAPPING];
	char orient[MAX_MAPPING], str[1024];

	for (r = 0; r < n_reads; r++) {
		const int l = l_qseq[r], p = pos[r], q = mapq[r], fl = flag[r];
		const uint8_t *sq = seq + data_off[r], *ql = qual + data_off[r];
		float avg = 0;
		int i, k, n, mq = 0, avg_floor, half = l / 2;

		/* bam_data.c:205-207 */
		if (!(q > mq_threshold))
			continue;
		if (!(l > min_read_length) || (fl & (F_SECONDARY | F_SUPPLEMENTARY | F_DUP | F_QCFAIL)) != 0)
			continue;
		if (is_satellite(sat_start, sat_end, n_sat, p, (int64_t) p + 20))
			continue;
		if (p == 0 || l > 1022)
			continue; /* split_read.c:216 */

		/* element 1: anchor pos, maps the second half (split_read.c:230-283) */
		for (i = half; i < l; i++)
			avg = avg + ql[i];
		avg = (float) avg / (float) (l - half);
		avg_floor = (int) floorf(avg);
		if (avg_floor < mq_threshold)
			continue; /* element 2 is never created either (split_read.c:245-251) */
		k = 0;
		for (i = half; i < l; i++)
			str[k++] = decode_base(sq[i]);
		str[k] = '\0';
		n = almost_perfect_match(ix, ref, L, str, p, posMap, orient, &mq);
		n_elem++;
		n_map += n;
		for (i = 0; i < n; i++)
			pair_one(&rows, L, sat_start, sat_end, n_sat, p, q, l, 0, posMap[i], orient[i], mq, mq_threshold);

		/* element 2: anchor pos + l/2, maps the first half; the quality mean CONTINUES from element 1's
		 * mean (the accumulator is not reset, split_read.c:307-310) */
		for (i = 0; i < half; i++)
			avg = avg + ql[i];
		avg = (float) avg / (float) half;
		avg_floor = (int) floorf(avg);
		if (avg_floor < mq_threshold)
			continue;
		k = 0;
		for (i = 0; i < half; i++)
			str[k++] = decode_base(sq[i]);
		str[k] = '\0';
		n = almost_perfect_match(ix, ref, L, str, p + half, posMap, orient, &mq);
		n_elem++;
		n_map += n;
		for (i = 0; i < n; i++)
			pair_one(&rows, L, sat_start, sat_end, n_sat, p + half, q, l, 1, posMap[i], orient[i], mq, mq_threshold);
	}
	free_index(ix);
	*rows_out = rows.v;
	if (counts) {
		int64_t i;
		counts[0] = n_elem;
		counts[1] = n_map;
		counts[2] = counts[3] = 0;
		for (i = 0; i < rows.n; i++)
			counts[rows.v[i].svType == ORACLE_DELETION ? 2 : 3]++;
	}
	return rows.n;
}
