// cmdline.h -- mirror of the reference's cmdline.h:6-7
#pragma once
#include "params.h"

namespace conga_host {

// Same contract as parse_cmd_line (cmdline.c:12-208): RETURN_SUCCESS (1) to go on,
// CONGA_EXIT_PARAM_ERROR (3) on a missing required option, CONGA_EXIT_SUCCESS (0) after --help / --version.
int parse_cmd_line(int argc, char **argv, parameters *params);
void print_help(void);
void get_working_directory(parameters *params);

} // namespace conga_host
