#!/bin/bash
# BAM stage and context creation of the CLI under measurement switches (after tools/e2e_quick.sh made the inputs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/e2e_quick.sh > /dev/null 2>&1
cd /tmp/e2e_wg
for mode in "$@"; do
 for i in 1 2 3; do
  t0=$(date +%s%N)
  env CONGA_TIMING=1 CONGA_T0_NS=$t0 $mode /root/repo/conga_amd/host/conga -i r.bam --ref r.fa --sonic a.cga --dels dels.bed --out x > x.log 2>&1
  t1=$(date +%s%N)
  echo "[$mode] wall $(( (t1 - t0) / 1000000 )) ms; $(grep -a 'conga_create' x.log | sed 's/.*streams + events/streams + events/'); $(grep -a 'conga_reads_bgzf:' x.log | sed 's/.*upload + inflate/upload + inflate/'); $(grep -a 'overlapped upload' x.log | sed 's/.*pieces by/pieces by/')"
 done
done
