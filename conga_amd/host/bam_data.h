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

// The same for every BAM listed in params->cohort_file (an extension: the reference takes one sample per process), each
// with its own three output files, in ONE process: the engine context, the HIP runtime and -- when the samples' headers
// select the same chromosomes -- the whole device layout are kept from sample to sample (conga_sample_begin).
int read_bam_cohort(parameters *params, sonic *this_sonic);

} // namespace conga_host
