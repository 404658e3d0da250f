#!/bin/bash
# builds tools/inflate_prof.hip in several experimental flavours ON the GPU box and runs each over one BAM
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
B=/tmp/iwexp.bam
[ -f $B ] || python3 $R/tools/mkbam.py $B 1,2,3 1
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -o /tmp/iwexp $R/tools/inflate_prof.hip 2>/dev/null
  echo "== $flags"
  case "$flags" in
  *IW_PROF*) timeout -k 10 120 /tmp/iwexp $B 1 | tail -10; timeout -k 10 120 /tmp/iwexp $B 8000 | tail -10 ;;
  *) timeout -k 10 120 /tmp/iwexp $B | grep -v "^  " | tail -1; timeout -k 10 120 /tmp/iwexp $B 8000 | grep -v "^  " | tail -1 ;;
  esac
done
