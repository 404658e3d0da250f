// likelihood.h -- per-chromosome SV genotyping on top of the C-ABI; mirrors likelihood.h:17-23.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/conga_hip.h"
#include "params.h"
#include "svs.h"

namespace conga_host {

extern int total_dels; // svdepth.c:13
extern int total_dups; // svdepth.c:14

// Everything find_SVs (likelihood.c:311-371) holds for one chromosome between loading and output.
struct chrom_svs {
	std::string chr_name;
	std::vector<sv_row> dels, dups;           // all_svs_del / all_svs_dup, sorted
	std::vector<conga_result> del_res, dup_res; // filled by the engine
};

// output_SVs (likelihood.c:172-288): same rows, same format strings, same filters.
// progress: where the "Found n DELs - m DUPs" line goes (NULL: stderr)
void output_SVs(const parameters *params, const chrom_svs &svs, FILE *fpSVs, FILE *fp_del, FILE *fp_dup, FILE *progress = nullptr);

} // namespace conga_host
