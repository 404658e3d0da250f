// annotation.h -- stand-in for the SONIC annotation the reference loads with sonic_load()
// (svdepth.c:47).  calkan/sonic is an un-vendored submodule, so `--sonic` takes this implementation's own
// container (conga_amd/formats.py: write_annotation).  Field and function names follow the call sites:
//   sonic->number_of_chromosomes / chromosome_lengths / chromosome_names   (bam_data.c:269-293)
//   sonic_get_gc_content(sonic, chr, start, end)                            (read_distribution.c:70, likelihood.c:117)
//   sonic_is_satellite(sonic, chr, start, end)                              (bam_data.c:96-97,207)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace conga_host {

struct sonic {
	int gc_step = 100;
	int number_of_chromosomes = 0;
	std::vector<std::string> chromosome_names;
	std::vector<int64_t> chromosome_lengths;
	std::vector<std::vector<uint8_t>> gc;                       // rounded GC% per window
	std::vector<std::vector<int32_t>> sat_start, sat_end;       // satellite intervals, sorted by start
};

// Returns nullptr (and a message in *err) on failure.
sonic *sonic_load(const std::string &path, std::string *err);
int sonic_refind_chromosome_index(const sonic *s, const std::string &chr);
// ASSUMED rule (parity unpinned): GC% of the gc_step-bp window that holds `start`,
// window index clamped to the chromosome's last window.
float sonic_get_gc_content(const sonic *s, int chr_index, int64_t start, int64_t end);
int sonic_is_satellite(const sonic *s, int chr_index, int64_t start, int64_t end);

// The two per-window arrays the C-ABI takes (conga_chrom_begin): loop B's lookup
// (i, min(i + step, L)) and loop C's lookup (i, i + step).  Under the assumed rule they coincide.
void gc_window_arrays(const sonic *s, int chr_index, std::vector<uint8_t> *gc_hist_w, std::vector<uint8_t> *gc_like_w);

} // namespace conga_host
