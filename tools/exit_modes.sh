#!/bin/bash
# wall time of the CLI on the whole-genome BAM with and without the orderly teardown (after tools/e2e_quick.sh made the inputs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/e2e_quick.sh > /dev/null 2>&1
cd /tmp/e2e_wg
for mode in default CONGA_CLEAN_EXIT=1 default CONGA_CLEAN_EXIT=1 default CONGA_CLEAN_EXIT=1; do
  if [ "$mode" = default ]; then e="X=1"; else e="$mode"; fi
  t0=$(date +%s%N)
  env CONGA_TIMING=1 CONGA_T0_NS=$t0 $e /root/repo/conga_amd/host/conga -i r.bam --ref r.fa --sonic a.cga --dels dels.bed --out x > x.log 2>&1
  t1=$(date +%s%N)
  echo "[$mode] wall $(( (t1 - t0) / 1000000 )) ms; $(grep -a 'leaving' x.log | sed 's/.*leaving: //')"
done
