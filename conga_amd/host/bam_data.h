// bam_data.h -- the per-chromosome driver; mirrors bam_data.h:17-19.
#pragma once
#include <cstdio>

#include "annotation.h"
#include "params.h"

namespace conga_host {

extern FILE *logFile; // svdepth.c:12

// read_bam (bam_data.c:224-359): opens the three output files, walks the annotation's chromosomes,
// streams each one's reads into the engine, loads its SVs and writes the genotypes.
// Returns a process exit code (0 on success).
int read_bam(parameters *params, sonic *this_sonic);

} // namespace conga_host
