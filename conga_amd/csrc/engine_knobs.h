// engine_knobs.h -- every switch the engine takes from the environment, read ONCE per context by read_knobs() (conga_create).
//
// No other function of the library calls getenv(): a hot call path (conga_chrom_compute, conga_reads_bgzf*, conga_sample_reads*)
// looks at ctx->knobs.  Two kinds (INTEGRATION.md lists them):
//   * test knobs, always read: they make files of test size take the routes that whole genomes take (the overlapped upload in
//     pieces of a few KB, a few decoder lanes, the chain kernel's class borders), turn a stage of the cohort pipeline off so that
//     the others are tested alone, or print where the time goes (CONGA_TIMING);
//   * measurement switches, read only with CONGA_DEBUG=1 in the environment: alternative kernels kept for comparison, ring and
//     stream geometry, a call that fails its checks on purpose (CONGA_BGZF_UPLOAD_ONLY) -- the lab bench, out of a production
//     process's way.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace conga {

struct Knobs {
	// ---- test knobs
	bool timing = false;              // CONGA_TIMING: stage timings on stderr
	bool graph = false;               // CONGA_GRAPH=1: the step as a captured hipGraph from the third compute of a layout on
	int chain_long_windows = -1;      // CONGA_CHAIN_LONG_WINDOWS / _SERIAL_ / _BLOCK_: class borders of the chain kernel (-1: default)
	int chain_serial_windows = -1;
	int chain_block_windows = -1;
	int bgzf_overlap = -1;            // CONGA_BGZF_OVERLAP: 0 / 1 forces the plain / the overlapped upload (-1: by size)
	long bgzf_piece_kb = 0;           // CONGA_BGZF_PIECE_KB: pieces of the overlapped upload (0: a slot of the ring)
	int bgzf_lanes = 0;               // CONGA_BGZF_LANES: decoder lanes of the lane kernel (0: by the number of blocks)
	bool bgzf_no_inflate_ahead = false; // CONGA_BGZF_NO_INFLATE_AHEAD: named bytes are brought up, not inflated ahead
	bool bgzf_no_table = false;       // CONGA_BGZF_NO_TABLE: the engine does not read block tables off named bytes
	bool bgzf_no_ahead = false;       // CONGA_BGZF_NO_AHEAD: conga_reads_bgzf_next_fd starts nothing
	// ---- measurement switches (CONGA_DEBUG=1)
	bool bgzf_lane_kernel = false;    // CONGA_BGZF_KERNEL=lane: round 1's one-block-per-lane inflate
	bool bgzf_one_phase = false;      // CONGA_BGZF_KERNEL=wave1: round 2's symbol loop
	bool bgzf_other_kernel = false;   // CONGA_BGZF_KERNEL set to anything but "wave": no inflating ahead (it uses the usual kernel)
	size_t bgzf_slot_bytes = (size_t) 8 << 20; // CONGA_BGZF_SLOT_MB
	int bgzf_slots = 12;              // CONGA_BGZF_SLOTS (2 .. 12)
	int bgzf_streams = 3;             // CONGA_BGZF_STREAMS (1 .. 3)
	bool bgzf_no_priority = false;    // CONGA_BGZF_NO_PRIORITY: copy and launch streams at one priority
	bool bgzf_plain_pread = false;    // CONGA_BGZF_PLAIN_PREAD: pread() straight into the pinned slot
	int bgzf_copy_threads = 0;        // CONGA_BGZF_COPY_THREADS
	int bgzf_launch_mb = 0;           // CONGA_BGZF_LAUNCH_MB: bytes of file per inflate launch
	bool bgzf_round_robin = false;    // CONGA_BGZF_ROUND_ROBIN: inflate launches deal their blocks round robin (no ticket counter)
	bool bgzf_upload_only = false;    // CONGA_BGZF_UPLOAD_ONLY: no inflate launches (the call then fails its checks)
	bool streams_normal = false;      // CONGA_STREAMS_NORMAL: the context's streams at the default priority, launch streams of their own
	int tuple_blocks_per_cu = 0;      // CONGA_TUPLE_BLOCKS_PER_CU
	int depth_tiles_per_block = 0;    // CONGA_DEPTH_TILES_PER_BLOCK
	int bgzf_groups_per_cu = 8;       // CONGA_BGZF_GROUPS_PER_CU: workgroups (of four waves) an inflate launch puts on a CU, 1 .. 8
	int bgzf_ahead_follow = 0;        // CONGA_BGZF_AHEAD_FOLLOW: launches ahead follow the batches whatever the upload's rate (bz::Config::ahead_wait_factor = 0)
	bool bgzf_ahead_one_stream = false; // CONGA_BGZF_AHEAD_ONE_STREAM: every launch ahead on the first of the two streams
	int bgzf_copy_streams = 1;        // CONGA_BGZF_COPY_STREAMS: 2 = the ring's odd slots go up on a second copy stream
	bool bgzf_slot_spin = false;      // CONGA_BGZF_SLOT_SPIN: a copying thread polls its slot's event (hipEventQuery) instead of hipEventSynchronize
	bool bgzf_trace = false;          // CONGA_BGZF_TRACE: the upload pipeline's events with a clock, on stderr (bz::trace)
	int split_flags = 0;              // CONGA_SPLIT_FLAGS: 1 = split_map_kernel does not ask the presence bitmaps, 2 = plain unit order
};

inline Knobs read_knobs()
{
	Knobs k;
	auto num = [](const char *name, int unset) {
		const char *e = getenv(name);
		return e ? atoi(e) : unset;
	};
	k.timing = getenv("CONGA_TIMING") != nullptr;
	k.graph = getenv("CONGA_GRAPH") != nullptr;
	k.chain_long_windows = num("CONGA_CHAIN_LONG_WINDOWS", -1);
	k.chain_serial_windows = num("CONGA_CHAIN_SERIAL_WINDOWS", -1);
	k.chain_block_windows = num("CONGA_CHAIN_BLOCK_WINDOWS", -1);
	k.bgzf_overlap = getenv("CONGA_BGZF_OVERLAP") ? (atoi(getenv("CONGA_BGZF_OVERLAP")) != 0 ? 1 : 0) : -1;
	k.bgzf_piece_kb = getenv("CONGA_BGZF_PIECE_KB") ? atol(getenv("CONGA_BGZF_PIECE_KB")) : 0;
	k.bgzf_lanes = num("CONGA_BGZF_LANES", 0);
	k.bgzf_no_inflate_ahead = getenv("CONGA_BGZF_NO_INFLATE_AHEAD") != nullptr;
	k.bgzf_no_table = getenv("CONGA_BGZF_NO_TABLE") != nullptr;
	k.bgzf_no_ahead = getenv("CONGA_BGZF_NO_AHEAD") != nullptr;
	const char *dbg = getenv("CONGA_DEBUG");
	if (!dbg || atoi(dbg) == 0)
		return k;
	if (const char *which = getenv("CONGA_BGZF_KERNEL")) {
		k.bgzf_lane_kernel = strcmp(which, "lane") == 0;
		k.bgzf_one_phase = strcmp(which, "wave1") == 0;
		k.bgzf_other_kernel = strcmp(which, "wave") != 0;
	}
	if (getenv("CONGA_BGZF_SLOT_MB"))
		k.bgzf_slot_bytes = (size_t) std::max(1, std::min(num("CONGA_BGZF_SLOT_MB", 8), 64)) << 20;
	k.bgzf_slots = std::max(2, std::min(num("CONGA_BGZF_SLOTS", 12), 12));
	k.bgzf_streams = std::max(1, std::min(num("CONGA_BGZF_STREAMS", 3), 3));
	k.bgzf_no_priority = getenv("CONGA_BGZF_NO_PRIORITY") != nullptr;
	k.bgzf_plain_pread = getenv("CONGA_BGZF_PLAIN_PREAD") != nullptr;
	k.bgzf_copy_threads = num("CONGA_BGZF_COPY_THREADS", 0);
	k.bgzf_launch_mb = num("CONGA_BGZF_LAUNCH_MB", 0);
	k.bgzf_upload_only = getenv("CONGA_BGZF_UPLOAD_ONLY") != nullptr;
	k.bgzf_round_robin = getenv("CONGA_BGZF_ROUND_ROBIN") != nullptr;
	k.streams_normal = getenv("CONGA_STREAMS_NORMAL") != nullptr;
	k.tuple_blocks_per_cu = num("CONGA_TUPLE_BLOCKS_PER_CU", 0);
	k.depth_tiles_per_block = num("CONGA_DEPTH_TILES_PER_BLOCK", 0);
	k.bgzf_trace = getenv("CONGA_BGZF_TRACE") != nullptr;
	k.bgzf_slot_spin = getenv("CONGA_BGZF_SLOT_SPIN") != nullptr;
	k.bgzf_copy_streams = std::max(1, std::min(num("CONGA_BGZF_COPY_STREAMS", 1), 2));
	k.bgzf_ahead_follow = num("CONGA_BGZF_AHEAD_FOLLOW", 0);
	k.bgzf_ahead_one_stream = getenv("CONGA_BGZF_AHEAD_ONE_STREAM") != nullptr;
	k.bgzf_groups_per_cu = std::max(1, std::min(num("CONGA_BGZF_GROUPS_PER_CU", 8), 8));
	k.split_flags = num("CONGA_SPLIT_FLAGS", 0);
	return k;
}

} // namespace conga
