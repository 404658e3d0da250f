/*
 * conga_hip.h -- C-ABI of the MI355X (gfx950) read-depth / likelihood engine.
 *
 * This is the drop-in seam for CONGA's per-chromosome hot path.  The reference is one
 * executable with no FFI; the seam sits where its BAM loop hands over to the depth model:
 *
 *   producer  count_reads_bam            /root/reference/bam_data.c:192-221
 *   consumers init_rd_per_chr            /root/reference/read_distribution.c:12-18
 *             calc_mean_per_chr          /root/reference/read_distribution.c:49-84
 *             init_mappability_per_chr   /root/reference/read_distribution.c:20-24
 *             load_mappability_regions   /root/reference/svs.c:317-377 (paint loop :363-371)
 *             find_depths / calculate_likelihood_CNV / lpoisson
 *                                        /root/reference/likelihood.c:96-169,290-308
 *   state     bam_info.read_depth / mappability / expected_read_depth
 *                                        /root/reference/common.h:88-104
 *             svs                        /root/reference/svs.h:10-28
 *
 * Conventions: plain pointers and sizes only; every function returns 0 (CONGA_OK) or a negative
 * conga_status and never calls exit(); the caller owns all host buffers it passes in; the library
 * owns device memory and the pinned staging buffers.  A context belongs to one GPU and must be
 * driven by one host thread at a time (the reference is single-threaded and keeps this state in
 * globals: likelihood.c:10-15).  There is no CPU fallback: without a usable HIP device
 * conga_create() fails.
 *
 * Two ways to drive it:
 *   sequential (default)  begin / reads / intervals / finish per chromosome, exactly the order of
 *                         the reference's loop (bam_data.c:269-339); each begin drops the previous
 *                         chromosome.
 *   batch (CONGA_FLAG_BATCH)  begin / reads / intervals for every chromosome of the sample, then ONE
 *                         conga_chrom_compute() -- each kernel is launched once over the whole batch --
 *                         then conga_chrom_select(i) + conga_chrom_fetch() per chromosome.  This is
 *                         the form that fills an MI355X; results are identical.
 *   cohort (batch + conga_sample_*)  the chromosomes, GC arrays, intervals and tracks are handed over once; every further
 *                         sample only replaces the read tuples (conga_sample_reads, or conga_sample_begin + the staging
 *                         ring) and is computed and fetched as before.  The reference runs one process per sample and
 *                         re-reads the annotation and the call set each time (svdepth.c:47-66); genotyping a cohort
 *                         against one call set is its use case (README.md:1-20).
 */
#ifndef CONGA_HIP_H_
#define CONGA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONGA_ABI_VERSION 9

typedef struct conga_ctx conga_ctx;

typedef enum conga_status {
	CONGA_OK = 0,
	CONGA_ERR_INVALID = -1,     /* bad argument or call order */
	CONGA_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range */
	CONGA_ERR_HIP = -3,         /* a HIP runtime call failed; see conga_last_error() */
	CONGA_ERR_NOMEM = -4,       /* host or device allocation failed */
	CONGA_ERR_UNSORTED = -5,    /* committed reads are not sorted by position (and CONGA_FLAG_READS_UNSORTED not set) */
	CONGA_ERR_RANGE = -6,       /* an interval or mappability row lies outside what the engine can address */
	CONGA_ERR_DATA = -7         /* conga_reads_bgzf: a block does not inflate to its size / CRC32, or the start points do not
	                               line up with the records; nothing was appended -- decode on the host instead */
} conga_status;

/* conga_opts.flags */
#define CONGA_FLAG_READS_UNSORTED 0x1u /* reads may arrive in any order: depth uses global atomics */
#define CONGA_FLAG_PROFILE 0x2u        /* bracket every kernel with HIP events; conga_chrom_stats.kernel_ms is filled */
#define CONGA_FLAG_BATCH 0x4u          /* conga_chrom_begin() ADDS a chromosome instead of replacing the previous one;
                                          conga_chrom_compute() then covers all of them in one launch per kernel */
#define CONGA_FLAG_MATERIALIZE_DEPTH 0x8u /* always build bam_info.read_depth[] in HBM, the way the reference does
                                          (the "dense" formulation: 4+ bytes of traffic per BASE).  Without this flag
                                          the engine works in tuple space whenever that is provably identical --
                                          read_depth[i] is a count of read starts, so the GC sums are a histogram over
                                          the kept READS and an interval's observed depth is the number of kept reads
                                          that start inside it (5 bytes of traffic per READ) -- and builds read_depth[]
                                          only when conga_copy_read_depth() asks for it.  It switches to the dense
                                          kernels by itself when the reads may be unsorted or when one base may hold
                                          more than 32767 read starts (the `short` of common.h:91 would wrap).
                                          The same holds for bam_info.mappability[]: a track whose rows are sorted
                                          and at most abutting is summed per interval straight from its rows; with
                                          this flag (or rows in any other order) it is painted base by base. */

#define CONGA_FLAG_RESULTS_ON_DEVICE 0x10u /* leave the result records in HBM: a compute does not send them over PCIe, and
                                          conga_chrom_fetch() copies them on demand.  For callers that consume the
                                          records on the device (conga_results_device / conga_results_copy: the
                                          multi-GPU gather over xGMI). */

#define CONGA_FLAG_EXPECT_BGZF 0x20u   /* conga_reads_bgzf() will be called: conga_create() gets its pinned staging ring (96 MB,
                                          ~50 ms) instead of the first such call */

#define CONGA_FLAG_EXPECT_COHORT 0x40u /* conga_reads_bgzf_next_fd() will be called for further inputs of about the size of the first
                                          (a tenth more fits): the second device buffer for compressed bytes and the spare output set
                                          (~3.6 bytes of HBM per byte of file) are allocated by the first conga_reads_bgzf* call before
                                          its first byte goes up -- an allocation is half a millisecond on an idle device and waits for
                                          whatever runs on a busy one --, the first input's own output set gets the spare's size (the two
                                          change places input by input), per-input buffers an eighth of room.  A context that holds
                                          reference text (conga_reference: split reads are mapped on the inflated stream in place) brings
                                          named bytes up ahead without inflating them ahead: only the second buffer for compressed bytes */

/* SV types, as the reference's DELETION / DUPLICATION (common.h:12-13) */
#define CONGA_DELETION 'D'
#define CONGA_DUPLICATION 'E'

typedef struct conga_opts {
	uint32_t struct_size;    /* sizeof(conga_opts), for forward compatibility */
	int32_t mq_threshold;    /* params->mq_threshold (cmdline.c:188-194): a read counts iff mapq > this; -1 = all */
	int32_t gc_step;         /* WINDOWSLIDE (read_distribution.h:8): bases per GC byte; 0 -> 100 */
	uint32_t flags;          /* CONGA_FLAG_* */
	int32_t min_read_length; /* params->min_read_length (cmdline.c:162-166), split reads only; <= 0 -> 60 */
	int32_t reserved[3];
} conga_opts;

/* Pinned host buffers the BAM loop fills (structure of arrays).  Replaces the fields of
 * bam1_core_t that count_reads_bam reads (bam_data.c:203-213): core.pos and core.qual. */
typedef struct conga_read_staging {
	int32_t *pos;    /* 0-based leftmost coordinate (bam1_core_t.pos) */
	uint8_t *mapq;   /* bam1_core_t.qual */
	size_t capacity; /* records that fit before the next conga_reads_commit() */
} conga_read_staging;

/* Output fields of one `svs` record (svs.h:10-28).  64 bytes, no padding holes. */
typedef struct conga_result {
	int32_t observed;    /* observed_rd_sv */
	float expected;      /* expected_rd_sv (serial float32 sum, bit-exact) */
	double lhomo;
	double lhetero;
	double lnone;
	double score;        /* likelihood_score */
	int32_t copy_number; /* 2 -> "1/1", else "0/1" */
	int32_t rp;          /* split-read support (dups); copied through from conga_split_support() */
	int32_t border_rp;   /* split-read support (dels) */
	int32_t reserved;
	double mappability;  /* mean mappability over [start, end); 0 when no track was given */
} conga_result;

enum {
	CONGA_K_INGEST = 0,   /* sortedness / range check + tile index over the read tuples; tuple-space formulation: also
	                         the read filter and the GC histogram over reads (K1 + K2 without read_depth[]) */
	CONGA_K_DEPTH,        /* dense formulation: LDS-tiled depth build + GC histogram (K1 + K2) */
	CONGA_K_EXPECTED,     /* expected_read_depth[101] */
	CONGA_K_PAINT,        /* mappability paint (K3) */
	CONGA_K_REDUCE,       /* per-interval integer depth sum + mappability sum (K4, memory side) */
	CONGA_K_SCORE,        /* serial-float expected chain (short intervals) + likelihoods + CN (K4 chain + K5) */
	CONGA_K_CHAIN,        /* serial-float expected chain of long intervals, one wave per interval */
	CONGA_K_COUNT_READS,  /* tuple-space formulation: per-interval count of kept reads (replaces the depth side of K_REDUCE) */
	CONGA_K_SPLIT,        /* split-read evidence: half-read mapping, pairing, support counts (split_map_kernel) */
	CONGA_K_EXPAND,       /* conga_sample_reads_packed: the differences back into int32 positions (delta16.hip.h: four launches);
	                         0 for a compute whose reads came as positions */
	CONGA_K_COUNT
};

typedef struct conga_chrom_stats {
	int64_t reads_committed;    /* tuples handed over for this chromosome */
	int64_t reads_counted;      /* bam_info.total_read_count_unfiltered (bam_data.c:214) */
	int64_t reads_out_of_range; /* pos outside [0, L): skipped (undefined behaviour in the reference) */
	int64_t rd_sum;             /* rd_cnt of calc_mu_per_chr (read_distribution.c:35) */
	float mean;                 /* bam_info.mean (read_distribution.c:39) */
	int32_t n_kernels;          /* CONGA_K_COUNT */
	int64_t rd_per_gc[101];     /* rd_per_gc_unfiltered (read_distribution.c:52) */
	int64_t window_per_gc[101]; /* window_per_gc (read_distribution.c:51) */
	double kernel_ms[12];       /* per-kernel device time of the last compute (CONGA_FLAG_PROFILE), else 0 */
	/* split-read evidence (only when split reads and a reference sequence were given) */
	int64_t split_elements;     /* split_read_count (split_read.c:14): half-read elements created */
	int64_t split_mappings;     /* mappings emitted by almostPerfect_match_seq_ref (split_read.c:185-201) */
	int64_t split_del_rows;     /* SplitRow records of type DELETION (bam_data.c:104-146) */
	int64_t split_dup_rows;     /* ... of type DUPLICATION */
	int32_t depth_materialized; /* 1: the last compute built read_depth[] (dense formulation), 0: tuple space */
	int32_t reserved;
} conga_chrom_stats;

/* ---- lifetime -------------------------------------------------------------------------- */

/* Creates a context on HIP device `device`.  Returns NULL on failure; *status (may be NULL)
 * receives the reason.  opts may be NULL (defaults: mq_threshold -1, gc_step 100, flags 0). */
conga_ctx *conga_create(int device, const conga_opts *opts, int *status);
void conga_destroy(conga_ctx *ctx);
const char *conga_strerror(int status);
/* Text of the most recent failure on this context ("" if none). */
const char *conga_last_error(const conga_ctx *ctx);
int conga_abi_version(void);
int conga_device_count(void);

/* ---- one chromosome --------------------------------------------------------------------- */

/* Batch mode: drop every chromosome held by the context (sequential mode does this in begin). */
int conga_reset(conga_ctx *ctx);
/* Number of chromosomes held, and which of them conga_mappability / conga_intervals /
 * conga_split_support / conga_chrom_fetch / conga_copy_* refer to (default: the one begun last).
 * Reads always stream into the chromosome begun last. */
int conga_chrom_count(const conga_ctx *ctx);
int conga_chrom_select(conga_ctx *ctx, int index);

/* init_rd_per_chr + the GC side of calc_mean_per_chr / calculate_likelihood_CNV.
 * chrom_len = sonic->chromosome_lengths[chr_index].  gc_hist_w[w] is the rounded GC% (0..100) that
 * `sonic_get_gc_content(chr, i, min(i + step, L))` yields for bases i in window w (loop of
 * read_distribution.c:63-73); gc_like_w[w] the one `sonic_get_gc_content(chr, i, i + step)` yields
 * (likelihood.c:117).  They may be the same pointer.  n_win must be ceil(chrom_len / gc_step); a value above
 * 100 is refused with CONGA_ERR_RANGE (the reference would index past its 101-entry tables).
 * The arrays are copied.  Sequential mode: resets all per-chromosome state (reads, intervals,
 * mappability, split support).  Batch mode: opens one more chromosome and selects it. */
int conga_chrom_begin(conga_ctx *ctx, int64_t chrom_len, const uint8_t *gc_hist_w, const uint8_t *gc_like_w,
		int64_t n_win);

/* count_reads_bam, producer side: get a pinned buffer, fill pos[0..n) / mapq[0..n) in BAM order,
 * commit.  Commit starts an asynchronous copy into HBM and returns; the next conga_reads_staging()
 * hands out the other buffer of the ring.  May be repeated any number of times per chromosome. */
int conga_reads_staging(conga_ctx *ctx, conga_read_staging *out);
int conga_reads_commit(conga_ctx *ctx, size_t n);

/* load_mappability_regions: rows of this chromosome in FILE ORDER (later rows overwrite earlier
 * ones, end inclusive: svs.c:368).  Copies the arrays; call at most once per chromosome.
 * Not calling it means "no --mappability" (conga_result.mappability = 0).  Rows that are sorted by start with
 * start[k + 1] >= end[k] (the bedGraph layout of the reference's README) take the fast paths. */
int conga_mappability(conga_ctx *ctx, const int32_t *start, const int32_t *end, const float *val, size_t m);

/* load_known_SVs + qsort output for this chromosome: rows already filtered
 * (end - start >= min_sv_size) and sorted by (start, end) (likelihood.c:324-328), type
 * CONGA_DELETION or CONGA_DUPLICATION.  Copies the arrays; once per type per chromosome. */
int conga_intervals(conga_ctx *ctx, char type, const int32_t *start, const int32_t *end, size_t n);

/* ---- cohort mode: the next sample's reads behind the same layout ----------------------------------------------
 * The consumer side of count_reads_bam (bam_data.c:192-221) for the second and every further sample of a cohort: the
 * reads of ALL chromosomes the context holds are replaced, everything else (chromosomes, GC arrays, intervals, tracks,
 * the device layout that conga_chrom_compute() prepared) stays.  With split reads the records go with the reads (they are the
 * sample's); the reference sequences, the satellites and the 10-mer indexes built from them are the layout's and stay.
 *
 * conga_sample_reads: pos / mapq hold the sample's (bam1_core_t.pos, bam1_core_t.qual) tuples, chromosome 0's first,
 * then chromosome 1's, ... each in BAM order; chromosome c owns [chrom_off[c], chrom_off[c + 1]) (n_chrom + 1 entries,
 * n_chrom == conga_chrom_count()).  With mq_threshold < 0 (the reference's default: every read counts) mapq is never read and
 * may be NULL -- 4 bytes per read cross the PCIe link instead of 5.  The copies into HBM are enqueued and the call returns: the
 * arrays must stay unchanged until a conga_chrom_fetch / conga_sample_fetch / conga_sync that FOLLOWS the next
 * conga_chrom_compute of this context has returned (or a conga_sync right away).  From pinned memory (conga_host_alloc) the
 * copy runs at the PCIe link's rate with no staging copy; pageable memory works, slower.
 * The hand-over is double-buffered: called behind a conga_chrom_compute, it leaves that compute's inputs and results alone --
 * the tuples go into a second pair of buffers on a stream of their own -- so that ONE context pipelines a cohort:
 *     conga_sample_reads(k + 1);  conga_sample_fetch(k);  conga_chrom_compute();      (the copy beside kernels and fetch)
 * The tuple-space formulation's guard against a wrapping `short` depth counter runs on the
 * device for such reads (see CONGA_FLAG_MATERIALIZE_DEPTH): a sample that needs the dense kernels is recomputed with
 * them inside the fetch (or conga_sync) that follows -- a caller that reads the records on the device
 * (CONGA_FLAG_RESULTS_ON_DEVICE) calls conga_sync() first.
 *
 * conga_sample_begin / conga_sample_chrom: the same through the staging ring, for a decoder that produces one
 * chromosome after the other: begin drops every chromosome's reads and makes chromosome 0 the target of
 * conga_reads_commit(); conga_sample_chrom(i) moves the target on (ascending only). */
void *conga_host_alloc(conga_ctx *ctx, size_t bytes); /* pinned host memory (hipHostMalloc); NULL on failure */
void conga_host_free(conga_ctx *ctx, void *p);
int conga_sample_reads(conga_ctx *ctx, const int32_t *pos, const uint8_t *mapq, const uint64_t *chrom_off, int n_chrom);
/* The same hand-over with the positions as 16-bit differences: the step of a cohort is the copy of the sample's tuples over PCIe,
 * and the positions of a position-sorted sample (bam1_core_t.pos in the order sam_itr_next yields them, bam_data.c:201-213) are
 * ~100 apart at 1x.  delta[i] = pos[i] - pos[i - 1] where that lies in [0, 0xFFFE]; otherwise 0xFFFF and an entry of the exception
 * list -- (esc_index[k] = i, esc_pos[k] = pos[i]), sorted by index.  The first read of every chromosome that has reads IS an
 * exception (nothing is carried over a chromosome's border), and so is a position in front of its predecessor (the engine's
 * order check then sees it as it is).  The engine turns the differences back into the int32 array all its kernels read (a
 * segmented scan, three small launches on the stream the copy runs on): 2 bytes per read over the link instead of 4.
 * Everything said of conga_sample_reads above holds (mapq, lifetime of the arrays, double-buffering). */
int conga_sample_reads_d16(conga_ctx *ctx, const uint16_t *delta, const uint32_t *esc_index, const int32_t *esc_pos, size_t n_esc,
		const uint8_t *mapq, const uint64_t *chrom_off, int n_chrom);
/* ... and as differences of `width` = 4 .. 16 bits, packed little-endian: difference i occupies bits [i * width,
 * (i + 1) * width) of `bits` (bit b of the stream is bit b & 7 of bits[b >> 3]; width 16 is conga_sample_reads_d16's array).  All ones
 * = see the exception list.  The producer picks the narrowest width that keeps exceptions rare (one in a thousand reads or fewer: an
 * exception costs the expansion a search of the list): at 1x two neighbours are ~100 bases apart and 10 bits hold all but one
 * difference in ten thousand -- 1.25 bytes per read over the link --, at 5x eight bits do, at 30x six.
 * EVERY all-ones difference must have its entry: the engine checks the list (sorted, inside the reads, the first read of every
 * chromosome) but not the stream against it -- an all-ones value without an entry takes the next entry's position (memory-safe,
 * positions wrong; conga_packer_* below keeps the contract by construction).
 * esc_index == esc_pos == NULL with n_esc > 0: the exceptions lie in `bits` behind the differences -- at the next multiple of 16
 * bytes behind ceil(n / 8) * width, esc_index[n_esc] then esc_pos[n_esc] -- and the sample goes up as ONE copy. */
int conga_sample_reads_packed(conga_ctx *ctx, const uint8_t *bits, int width, const uint32_t *esc_index, const int32_t *esc_pos, size_t n_esc,
		const uint8_t *mapq, const uint64_t *chrom_off, int n_chrom);
/* The PRODUCER of that format, for positions that lie in an array (host code only: no device, no context -- the caller's side of the
 * seam, count_reads_bam's loop, bam_data.c:201-213, for a producer that has not subtracted while it decoded).  A packer owns a pool of
 * host threads (n_threads <= 0: half of the cores the process may use).  conga_packer_start() begins to encode one sample -- pos[] /
 * chrom_off[] as in conga_sample_reads; width 4 .. 16, or 0: the narrowest width that keeps exceptions at or below one read in a
 * thousand, judged on a sample of the differences -- into `out` in the ONE-COPY layout (differences, exceptions behind them at the next
 * multiple of 16 bytes) and returns at once; conga_packer_finish() waits for it and says which width was used, how many exceptions
 * there are and how many bytes of `out` go over the link: hand (out, *width, NULL, NULL, *n_esc) to conga_sample_reads_packed.
 * `out` holds conga_pack_bound(n_reads, max_esc) bytes for up to max_esc exceptions at any width (CONGA_ERR_NOMEM from start / finish:
 * it does not); pos[] and out[] must stay as they are between start and finish.  One sample at a time per packer; every all-ones
 * difference gets its entry (the contract conga_sample_reads_packed relies on).  On Linux the packer's threads keep to the CPUs of the
 * memory node the positions lie on (a two-socket host: reading across the sockets' link costs the encode a third of its rate). */
typedef struct conga_packer conga_packer;
conga_packer *conga_packer_create(int n_threads);
void conga_packer_destroy(conga_packer *p);
int conga_packer_threads(const conga_packer *p); /* host threads of its pool */
size_t conga_pack_bound(uint64_t n_reads, size_t max_esc);
int conga_packer_start(conga_packer *p, const int32_t *pos, const uint64_t *chrom_off, int n_chrom, int width, uint8_t *out, size_t out_cap);
/* ... for a decoder that leaves one array per chromosome (the loop of read_bam, bam_data.c:269-339, calls count_reads_bam once per
 * chromosome): chrom_pos[c] holds chromosome c's chrom_off[c + 1] - chrom_off[c] positions (not looked at where that is 0).  The bytes
 * written are those conga_packer_start writes for the same reads in one array. */
int conga_packer_start_v(conga_packer *p, const int32_t *const *chrom_pos, const uint64_t *chrom_off, int n_chrom, int width, uint8_t *out,
		size_t out_cap);
int conga_packer_finish(conga_packer *p, int *width, size_t *n_esc, size_t *out_bytes);
int conga_sample_begin(conga_ctx *ctx);
int conga_sample_chrom(conga_ctx *ctx, int index);
/* conga_chrom_fetch for every chromosome in one call: records[] receives conga_chrom_count() groups one behind the
 * other (chromosome order; a chromosome's deletions, then its duplications -- the order of conga_results_device()),
 * n_records must be the total; expected_rd (may be NULL) receives 101 floats per chromosome, stats (may be NULL) one
 * entry per chromosome. */
int conga_sample_fetch(conga_ctx *ctx, conga_result *records, size_t n_records, float *expected_rd, conga_chrom_stats *stats);

/* ---- two computes in flight (ABI v9) -------------------------------------------------------------------------
 * The reference handles one sample per process, one chromosome after the other (bam_data.c:269-339): nothing of sample k + 1
 * depends on sample k.  With the loop above the GPU still waits for the host between two samples -- for the wait on sample k
 * to return, for its records to be taken out, for the launches of sample k + 1 to arrive: 0.27 ms a step where the kernels take
 * 0.12 (one chromosome of a 1x genome, the share of a rank on eight GPUs).  conga_chrom_compute_ahead() enqueues a compute
 * BEHIND the last one and keeps that one's results (pinned read-back blocks, records in HBM, statistics) where they are:
 *     conga_sample_reads(k + 1);  conga_chrom_compute_ahead();  conga_sample_fetch_previous(k);
 * The `_previous` calls are their namesakes for the compute before the latest one; the plain calls keep addressing the latest
 * (the last sample of a loop is fetched with conga_sample_fetch).  The wrap guard of the older compute is settled inside them --
 * out of the pair of tuple buffers that no copy is writing: the next conga_sample_reads* / conga_sample_begin settles it first if
 * the caller has not (and then waits for that compute) --, a second compute in the dense formulation is enqueued behind the
 * latest one's launches.  conga_chrom_compute() gives up whatever an older compute left; conga_chrom_compute_ahead() is
 * conga_chrom_compute() when there is nothing to keep (no compute yet, a changed layout, split reads, CONGA_FLAG_PROFILE). */
int conga_chrom_compute_ahead(conga_ctx *ctx);
int conga_sample_fetch_previous(conga_ctx *ctx, conga_result *records, size_t n_records, float *expected_rd, conga_chrom_stats *stats);
int conga_sync_previous(conga_ctx *ctx);   /* waits for that compute alone (the latest one may still run) and settles its guard */
int conga_results_copy_previous(conga_ctx *ctx, void *dst_device, size_t dst_bytes); /* conga_results_copy of its records, on the context's stream */

/* ---- count_reads_bam with the BAM decode on the device -------------------------------------------------------
 * Instead of decoded tuples the caller hands over a stretch of the BAM file exactly as it is on disk, the table of
 * its BGZF blocks, and start points taken from the .bai's linear index (16 kb windows).  The engine inflates the blocks
 * (one BGZF block per wave, conga_amd/csrc/inflate_wave.hip.h, CRC32 checked) and walks the records (one lane
 * per start point), and gives every chromosome named by the start points its reads: (pos, mapq) of the records with
 * refID == ref_id and 0 <= pos < pos_hi of its last segment -- the records `sam_itr_queryi(idx, tid, 0, L)` +
 * `sam_itr_next` hand to count_reads_bam (bam_data.c:192-221, 293) -- in file order, without the tuples ever being on
 * the host.  One call may serve several chromosomes of a batch context (a whole low-coverage genome is one launch).
 *   blocks[i]   : one BGZF block with a non-empty payload, in file order
 *   segments[k] : `start` = offset, in the concatenation of the inflated blocks, of a record boundary at or in front
 *                 of the first record of target `ref_id` whose position is >= pos_lo; the segment owns the records with
 *                 pos_lo <= pos < pos_hi.  `chrom` = index of the chromosome (conga_chrom_select) the records go to.
 *                 Segments are grouped by ascending `chrom`; those of one chromosome are in order and tile [0, L).
 * No chromosome from the first named one on may have reads yet (the tuples of a context are laid out in chromosome
 * order); chromosomes in between that are not named get none.  The engine verifies that every segment stops exactly
 * where the next one of its chromosome found its first record; if not, or if a block fails its checks, it returns
 * CONGA_ERR_DATA and has changed nothing.
 * Split reads (--rp with --dups): a chromosome named here that has a reference sequence (conga_reference) gets its split-read
 * records from the same call -- the engine notes where each kept record starts in the inflated stream, keeps the stream, and
 * find_split_reads' fields (core.pos, core.qual, core.flag, core.l_qseq, bam_get_seq, bam_get_qual: split_read.c:206-354) are
 * read where they lie in HBM; conga_split_reads_staging / _commit are not called for such a chromosome.  A kept record whose
 * name, CIGAR, sequence and qualities do not fit its block_size is CONGA_ERR_DATA (the host reader's "corrupt BAM record"). */
typedef struct conga_bgzf_block {
	uint64_t data_off;     /* of the raw deflate data inside `bytes` */
	uint32_t data_len;
	uint32_t inflated_len; /* ISIZE */
	uint32_t crc32;        /* of the inflated bytes */
	uint32_t reserved;
} conga_bgzf_block;
typedef struct conga_bam_segment {
	uint64_t start;
	int32_t pos_lo, pos_hi;
	int32_t ref_id;        /* refID of the target in the BAM header */
	int32_t chrom;         /* chromosome of the context */
} conga_bam_segment;
int conga_reads_bgzf(conga_ctx *ctx, const uint8_t *bytes, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		const conga_bam_segment *segments, size_t n_segments, uint64_t *reads_per_chrom /* [conga_chrom_count()] or NULL */);

/* The same with the bytes still in the file: [file_off, file_off + n_bytes) of the open descriptor `fd` is read with pread
 * straight into the pinned pieces that go up (no mapping of the file, no page faults, no copy in between); blocks[].data_off
 * are relative to file_off, as they are to `bytes` above. */
int conga_reads_bgzf_fd(conga_ctx *ctx, int fd, uint64_t file_off, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		const conga_bam_segment *segments, size_t n_segments, uint64_t *reads_per_chrom /* [conga_chrom_count()] or NULL */);

/* A cohort's pipeline (conga --cohort: read_bam_cohort).  Names the bytes that a LATER conga_reads_bgzf_fd call of this context
 * will bring -- same descriptor, offset and length --: the context's upload thread starts on them as soon as the bytes of the
 * call in progress are up (bytes named between two calls go up when the next call begins, behind its own), into the other of
 * two device buffers, so that sample k + 1 crosses the link while sample k is inflated, walked, computed and written out.
 * The reference reads its samples one process at a time (bam_data.c:253-339); nothing of its results depends on when a
 * sample's bytes were copied.  At most three stretches are held named; *ticket is 0 when nothing was started (a fourth one, a
 * stretch too small for the overlapped route, no pinned ring yet).
 *
 * known_starts (n_known > 0): offsets inside the stretch at which the caller KNOWS a BGZF block to begin -- the index's linear
 * offsets, ascending, the first one 0.  The engine then reads the block table off the bytes while they pass through its pinned
 * ring (every copying thread follows the chain of headers inside its piece from the first known start on; what straddles two
 * pieces is read from the file), publishes it batch by batch, and -- once a call of this context has shown how much such a
 * file inflates to -- inflates the batches ahead as well, into a spare output buffer.  stop_at (0: none; otherwise ONE MORE than an
 * offset inside the stretch -- offset 0, the stretch's first block, is an offset like any other): the table ends with the first block
 * that begins at or behind that offset.  conga_reads_bgzf_next_table() waits for the table (n_blocks 0: none
 * -- a header of an unusual form, a chain that does not arrive at a known start: the caller walks the file itself); the
 * conga_reads_bgzf_fd call that brings the SAME table finds the stream inflated and goes straight to its record walks, any
 * other table makes it inflate as usual.  The pointer stays valid until that call has returned or the ticket is forgotten.
 *
 * A caller that has read the table by itself hands it over with conga_reads_bgzf_next_blocks() (copied) to the same effect.
 *
 * conga_reads_bgzf_next_go(ticket) (ABI v8): "the call in front of these bytes' own has begun (or returned): nothing else will be
 * brought before them" -- bytes that were named between two calls then start at once instead of with the next call.  A driver whose
 * thread of the calls waits for conga_reads_bgzf_next_table() of the NEXT sample before it begins that sample's call needs this:
 * the table waits for the bytes, the bytes for the call, the call for the table (the engine breaks that circle after 0.4 s by
 * returning no table -- it cost `conga --cohort` 0.44 s in one run out of three until the executable said "go").
 *
 * All of these may be called from another thread than the one inside a call of this context: they touch the upload thread's
 * queue only.  The descriptor must stay open until the conga_reads_bgzf_fd call that takes the bytes up has returned, or until
 * conga_reads_bgzf_forget(ticket) has: that one gives an upload up that no call will ask for (the caller decided to decode on
 * the host) and is a no-op for a ticket already taken up. */
int conga_reads_bgzf_next_fd(conga_ctx *ctx, int fd, uint64_t file_off, size_t n_bytes, const uint64_t *known_starts, size_t n_known,
		uint64_t stop_at, uint64_t *ticket);
int conga_reads_bgzf_next_table(conga_ctx *ctx, uint64_t ticket, const conga_bgzf_block **blocks, size_t *n_blocks);
int conga_reads_bgzf_next_blocks(conga_ctx *ctx, uint64_t ticket, const conga_bgzf_block *blocks, size_t n_blocks);
int conga_reads_bgzf_next_go(conga_ctx *ctx, uint64_t ticket);
int conga_reads_bgzf_forget(conga_ctx *ctx, uint64_t ticket);

/* Gives the pinned staging of conga_reads_bgzf* (96 MB) back to the system; the next call makes it again.  A caller that is
 * done reading -- the conga executable after its last BAM -- calls this, from a thread of its own if it likes, while the
 * context computes: pinned memory that is still there when the process ends is taken down by the driver at four times the
 * cost.  Must not run beside a conga_reads_bgzf* call of the same context. */
int conga_release_staging(conga_ctx *ctx);

/* Test / tool hook: the first stage of conga_reads_bgzf alone.  Inflates the blocks on the device and copies their payloads
 * (one behind the other) to `out` (may be NULL); status[b]: 0 inflated and CRC32 right, 1 refused (not a valid deflate stream
 * of the recorded size), 2 CRC32 mismatch.  kernel_ms (may be NULL): device time of the inflate launch.  Any bytes will do:
 * no BAM structure is assumed. */
int conga_inflate_blocks(conga_ctx *ctx, const uint8_t *bytes, size_t n_bytes, const conga_bgzf_block *blocks, size_t n_blocks,
		uint8_t *out, size_t out_bytes, uint8_t *status, double *kernel_ms);

/* ---- split-read evidence: find_split_reads / read_SplitReads / count_ReadPairs on the device --------------
 * Used when the reference would run its split-read path (`--rp` given AND `--dups` given: svdepth.c:57,
 * bam_data.c:207,306,331, likelihood.c:344).  For the chromosome begun last, hand over
 *   - its reference sequence (readReferenceSeq, common.c:423-463; upper-cased by the callee),
 *   - its satellite intervals (sonic_is_satellite, bam_data.c:96-97,207; any order, may overlap),
 *   - every record of the BAM loop with the fields find_split_reads touches (split_read.c:206-354):
 *     core.pos, core.qual, core.flag, core.l_qseq, the packed 4-bit sequence and the base qualities.
 * The gate of bam_data.c:205-207 (mapq, length, flags, satellite) is applied by the engine.
 * conga_chrom_compute() then builds the 10-mer index (once per reference sequence: it stays resident), maps both halves of
 * every read, pairs them and adds the support counts into conga_result.rp (dups) / border_rp (dels).
 * Records go to the chromosome begun last, or to the one conga_sample_chrom() named; a chromosome's records are committed
 * without another's in between.  A chromosome without a reference sequence takes no part (its records are carried, no more). */
typedef struct conga_split_staging {
	int32_t *pos;       /* bam1_core_t.pos */
	uint8_t *mapq;      /* bam1_core_t.qual */
	uint16_t *flag;     /* bam1_core_t.flag */
	int32_t *l_qseq;    /* bam1_core_t.l_qseq */
	uint64_t *data_off; /* offset of record i's block inside `data` */
	uint8_t *data;      /* per record: (l_qseq + 1) / 2 bytes bam_get_seq() then l_qseq bytes bam_get_qual() */
	size_t capacity_reads;
	size_t capacity_bytes;
} conga_split_staging;
int conga_reference(conga_ctx *ctx, const char *seq, int64_t len);
int conga_satellites(conga_ctx *ctx, const int32_t *start, const int32_t *end, size_t n);
int conga_split_reads_staging(conga_ctx *ctx, conga_split_staging *out);
int conga_split_reads_commit(conga_ctx *ctx, size_t n_reads, size_t n_bytes);

/* count_ReadPairs result for this chromosome (likelihood.c:41-94): per-interval split-read support
 * in the order of conga_intervals(); copied through to conga_result.rp (dups) / border_rp (dels). */
int conga_split_support(conga_ctx *ctx, char type, const int32_t *support, size_t n);

/* calc_mean_per_chr + find_depths: enqueue every kernel for the chromosome (batch mode: for all
 * chromosomes held) on the context's stream and return without waiting.  May be called again on
 * the same resident inputs (it recomputes everything from the committed tuples). */
int conga_chrom_compute(conga_ctx *ctx);

/* Waits for the last conga_chrom_compute() and copies the selected chromosome's results out.  dels / dups receive one record
 * per interval in the order given to conga_intervals() (NULL allowed when that type has no
 * intervals); expected_rd receives bam_info.expected_read_depth; stats may be NULL. */
int conga_chrom_fetch(conga_ctx *ctx, conga_result *dels, conga_result *dups, float expected_rd[101],
		conga_chrom_stats *stats);

/* conga_chrom_compute() followed by conga_chrom_fetch(). */
int conga_chrom_finish(conga_ctx *ctx, conga_result *dels, conga_result *dups, float expected_rd[101],
		conga_chrom_stats *stats);

/* ---- device-side access (multi-GPU gather, profiling, tests) ------------------------------ */

/* Device pointer to the result records of the last compute: chromosomes in begin order, each with
 * its deletions then its duplications; valid until the next conga_chrom_begin() / conga_reset().
 * Used to gather results over RCCL without a host hop. */
int conga_results_device(conga_ctx *ctx, void **dev_ptr, size_t *n_records);
/* Enqueue a device-to-device copy of those records into dst_device (at least
 * n_records * sizeof(conga_result) bytes, e.g. a torch tensor's data_ptr) on the context's stream. */
int conga_results_copy(conga_ctx *ctx, void *dst_device, size_t dst_bytes);
/* Turn per-kernel HIP-event timing (CONGA_FLAG_PROFILE) on or off for later computes. */
int conga_set_profile(conga_ctx *ctx, int on);
/* hipStream_t of this context (as void*), e.g. to record HIP events around conga_chrom_compute(). */
void *conga_stream(conga_ctx *ctx);
/* Blocks until the context's stream is idle (and settles the wrap guard of conga_sample_reads, see there). */
int conga_sync(conga_ctx *ctx);
/* Test hooks: copy bam_info.read_depth (int16[chrom_len]) / bam_info.mappability (float[chrom_len])
 * of the last compute back to the host.  After a tuple-space compute conga_copy_read_depth() first builds
 * read_depth[] from the resident tuples (K0 + K1). */
int conga_copy_read_depth(conga_ctx *ctx, int16_t *out, int64_t n);
int conga_copy_mappability(conga_ctx *ctx, float *out, int64_t n);

/* Host-callable build of the device routine that advances the serial float32 accumulator of
 * likelihood.c:119 by k equal addends in O(1) (used by CPU tests to check it against k real adds). */
float conga_host_repeat_add_f32(float s, float c, uint32_t k);
/* Host build of the per-window variant the chain kernels call (at most one binade crossing handled without a
 * loop or an integer divide, everything else passed on to the routine above); same contract. */
float conga_host_window_add_f32(float s, float c, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif /* CONGA_HIP_H_ */
