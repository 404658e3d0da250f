// kernels_bam.hip.h -- BAM decode on the device (conga_reads_bgzf): BGZF inflate and the record walk.
//
// The producer side of count_reads_bam (bam_data.c:192-221) is htslib's BGZF + BAM iterator in the reference.  Here the
// compressed blocks go to HBM as they are and
//   bgzf_inflate_kernel  one LANE per BGZF block runs the host reader's block decoder (conga_amd/host/inflate_core.h,
//                        the same source, tables in a per-lane scratch in HBM) and checks the block's CRC32.  No wave
//                        cooperation: a genome's worth of independent blocks keeps every lane of the chip busy, and the
//                        measured rate (17 GB/s inflated with 45 000 blocks in flight, tools/gpu_inflate.hip) is already
//                        several times a 16-core host's.
//   bam_walk_kernel      one lane per start point of the .bai's linear index follows the block_size chain of the
//                        records, skips what lies in front of its window, stops at the first record of the next
//                        segment; a counting pass, an exclusive scan on the host, then the same walk writes (pos, mapq)
//                        into the context's tuple arrays.
//   equal_run_kernel     the tuple-space formulation's guard (a `short` depth counter wraps after 32767 read starts on
//                        one base): looks for a run of kWrapRun equal positions in what was appended.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/conga_hip.h"
#include "../host/inflate_core.h"
#include "inflate_wave.hip.h"

namespace conga {

using InflateScratch = conga_host::inflate_core::Decoder;

enum { kBgzfOk = 0, kBgzfRefused = 1, kBgzfCrc = 2 };

__device__ __forceinline__ uint32_t crc32_bytes(const uint32_t *table, const uint8_t *p, uint32_t n)
{
	uint32_t c = 0xFFFFFFFFu;
	for (uint32_t i = 0; i < n; i++)
		c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
	return c ^ 0xFFFFFFFFu;
}

// status[b]: kBgzfOk / kBgzfRefused / kBgzfCrc.  Lanes loop over blocks with the launch's lane count as stride, so the
// scratch is one decoder per launched lane.
// (Register budget for 4 waves per SIMD: left alone the compiler takes 256 registers, one wave per SIMD, and a genome's
// 79 000 blocks then need two rounds of 65 536 resident lanes.)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void bgzf_inflate_kernel(uint32_t n_blocks, const uint8_t *__restrict__ bytes,
		const conga_bgzf_block *__restrict__ blocks, const uint64_t *__restrict__ out_off, uint8_t *__restrict__ out,
		InflateScratch *scratch, const uint32_t *__restrict__ crc_table, uint8_t *__restrict__ status)
{
	__shared__ uint32_t s_crc[256]; // (the byte-wise CRC looks one entry up per byte)
	for (int i = threadIdx.x; i < 256; i += blockDim.x)
		s_crc[i] = crc_table[i];
	__syncthreads();
	const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x, lanes = gridDim.x * blockDim.x;
	// the scratch is raw hipMalloc memory (a member initialiser never runs on the device, and the allocator recycles
	// blocks): the fixed-code tables count as built only once THIS launch has built them
	scratch[lane].fixed_ready = false;
	for (uint32_t b = lane; b < n_blocks; b += lanes) {
		const conga_bgzf_block bl = blocks[b];
		uint8_t *dst = out + out_off[b];
		uint8_t st = kBgzfOk;
		if (!conga_host::inflate_core::inflate_block_stream(scratch[lane], bytes + bl.data_off, bl.data_len, dst, bl.inflated_len))
			st = kBgzfRefused;
		else if (crc32_bytes(s_crc, dst, bl.inflated_len) != bl.crc32)
			st = kBgzfCrc;
		status[b] = st;
	}
}

struct BamWalkArgs {
	const uint8_t *stream;  // the inflated blocks, concatenated
	uint64_t stream_len;
	const conga_bam_segment *segments;
	uint32_t n_segments;
	// per segment
	uint32_t *count;        // records it owns (pass 1)
	uint64_t *v_first;      // where it found its first own record (or where it stopped, if it owns none)
	uint64_t *v_stop;       // the record that ended it (first record of the next segment / of another target); ~0: end of stream
	uint8_t *bad;           // 1: a record that cannot be one (block_size < 32, running past the stream, fields that do not fit it);
	                        // 2: a record in front of its predecessor (the file is not sorted by position)
	const uint64_t *write_at; // pass 2: first tuple of each segment
	int32_t *pos;
	uint8_t *mapq;
	uint64_t *rec_off;      // pass 2, or nullptr: where each kept record starts (rec_base + its offset in `stream`), for the
	uint64_t rec_base;      // split-read path, which reads the records where they lie (split_map.hip.h)
	uint32_t check_body;    // pass 1: a kept record's name, CIGAR, sequence and qualities must fit its block_size (the host
	                        // reader calls anything else a corrupt BAM record; the split-read path is about to read them)
};

// a little-endian int32 at ANY address: global memory takes unaligned dword accesses (one load instead of four byte loads
// put together -- a record's fields are what the walk below reads, one lane per segment, every load a cache line of its own)
__device__ __forceinline__ int32_t load_i32(const uint8_t *p)
{
	uint32_t v;
	__builtin_memcpy(&v, p, 4); // (one global_load_dword: the alignment is the compiler's to know)
	return (int32_t) v;
}

template <bool WRITE> __global__ __launch_bounds__(64) void bam_walk_kernel(BamWalkArgs a)
{
	const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= a.n_segments)
		return;
	const conga_bam_segment sg = a.segments[k];
	constexpr uint64_t kNone = ~0ull;
	uint64_t at = sg.start, first = kNone, stop = kNone;
	uint32_t n = 0;
	uint64_t w = WRITE ? a.write_at[k] : 0;
	int32_t hp0 = 0, hp1 = 0, hp2 = 0; // WRITE: the tuples held back for the next 16-byte store (named: nothing indexed by a variable)
	uint32_t hm = 0;
	int held = 0;
	uint8_t bad = 0;
	for (;;) {
		if (at + 12 > a.stream_len)
			break; // end of the stream (stop stays kNone)
		const uint8_t *r = a.stream + at;
		const int32_t block_size = load_i32(r);
		if (block_size < 32) {
			bad = 1;
			break;
		}
		const int32_t ref = load_i32(r + 4), p = load_i32(r + 8);
		const uint64_t here = at;
		if (!(ref >= 0 && ref < sg.ref_id) && (ref != sg.ref_id || p >= sg.pos_hi)) {
			// the record that ends the segment; only its first fields are needed (the piece of the file may end inside it).
			// It has to look like one: of this target behind the segment, of a later target, or of the unplaced tail -- a
			// damaged header in the middle of a target must not pass for the target's end (tools/bam_fuzz.py found that it did:
			// the reads behind it were left out without a word)
			const bool plausible = block_size <= (1 << 26)
					&& (ref == sg.ref_id || (ref > sg.ref_id && ref < (1 << 24) && p >= -1) || (ref == -1 && p >= -1));
			if (!plausible) {
				bad = 1;
				break;
			}
			stop = here;
			if (first == kNone)
				first = here;
			break;
		}
		if (at + 4 + (uint64_t) block_size > a.stream_len) {
			bad = 1; // a record of this target that is not all there
			break;
		}
		at += 4 + (uint64_t) block_size;
		if (ref >= 0 && ref < sg.ref_id)
			continue; // (the tail of the previous target in front of this one's first record)
		if (p < sg.pos_lo) {
			if (first != kNone) { // behind a record of this segment: the file is not sorted by position -- not this decoder's to judge
				bad = 2;
				break;
			}
			continue; // starts in front of this segment: the previous one's
		}
		if (first == kNone)
			first = here;
		if (p < 0)
			continue;
		if (!WRITE && a.check_body) {
			const uint32_t names = (uint32_t) load_i32(r + 12), cigars = (uint32_t) load_i32(r + 16);
			const int32_t l_seq = load_i32(r + 20);
			const uint64_t l = l_seq > 0 ? (uint64_t) l_seq : 0;
			if (l_seq < 0 || 32u + (uint64_t) (names & 0xFFu) + 4u * (uint64_t) (cigars & 0xFFFFu) + (l + 1) / 2 + l > (uint64_t) block_size) {
				bad = 1;
				break;
			}
		}
		if (WRITE) {
			// A lane's tuples go out four at a time -- a 16-byte store of positions and a 4-byte store of MAPQ bytes once the
			// lane's place is a multiple of four: sixty-four lanes store to sixty-four different lines, and a 4-byte piece of a
			// line is a read-modify-write of all of it for the memory (the writing walk took 1.6 ms of a 1x genome where the
			// counting one takes 0.67).
			const uint32_t mq = ((uint32_t) load_i32(r + 12) >> 8) & 0xFFu; // l_read_name, MAPQ, bin: the byte at 13
			if (a.rec_off)
				a.rec_off[w + held] = a.rec_base + here;
			if (held == 0 && (w & 3u) != 0) { // (up to three in front of the first whole group)
				a.pos[w] = p;
				a.mapq[w] = (uint8_t) mq;
				w++;
			} else {
				hm |= mq << (8 * held);
				if (held == 3) {
					*reinterpret_cast<int4 *>(a.pos + w) = make_int4(hp0, hp1, hp2, p);
					*reinterpret_cast<uint32_t *>(a.mapq + w) = hm;
					w += 4;
					held = 0;
					hm = 0;
				} else {
					hp2 = held == 2 ? p : hp2;
					hp1 = held == 1 ? p : hp1;
					hp0 = held == 0 ? p : hp0;
					held++;
				}
			}
		}
		n++;
	}
	if (WRITE)
		for (int j = 0; j < held; j++) { // (what is left of the last group)
			a.pos[w + j] = j == 0 ? hp0 : j == 1 ? hp1 : hp2;
			a.mapq[w + j] = (uint8_t) (hm >> (8 * j));
		}
	if (!WRITE) {
		a.count[k] = n;
		a.v_first[k] = first;
		a.v_stop[k] = stop;
		a.bad[k] = bad;
	}
}

// flag |= 1 when pos[i] == pos[i + run - 1] for some i (positions are sorted: a run of `run` equal values)
__global__ __launch_bounds__(256) void equal_run_kernel(const int32_t *__restrict__ pos, uint64_t n, uint32_t run, uint32_t *flag)
{
	const uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i + run - 1 < n && pos[i] == pos[i + run - 1])
		atomicOr(flag, 1u);
}

} // namespace conga
