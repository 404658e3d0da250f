// svs.h -- known-SV and mappability BED loading; mirrors svs.h:10-32 of the reference.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/conga_hip.h"
#include "params.h"

namespace conga_host {

// Input half of the reference's `svs` record (svs.h:10-28); the output half is conga_result.
struct sv_row {
	int32_t start;
	int32_t end;
};

// One parsed BED, bucketed by chromosome name, rows in FILE ORDER inside a bucket.  The reference re-reads
// every file for every chromosome (svs.c:7-240, 317-377); one pass yields the same rows in the same order.
struct bed_index {
	std::map<std::string, std::vector<sv_row>> rows;
	std::map<std::string, std::vector<float>> values; // 4th column, mappability only
};

// Tokenisation exactly as the reference: 512-byte fgets chunks, delimiters " \t\r\n", atoi / atof,
// blank chunks skipped.  Returns false when the file cannot be opened.
bool load_bed(const std::string &path, bool with_value, bed_index *out);

// load_known_SVs for one chromosome (svs.c:55: strcmp(chr) == 0 && end - start >= min_sv_size), followed by
// the qsort of find_SVs (likelihood.c:324-328, comparator common.c:199-215).
std::vector<sv_row> known_SVs_for(const bed_index &bed, const std::string &chr, int min_sv_size);

} // namespace conga_host
