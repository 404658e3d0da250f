#!/bin/bash
# ten whole genomes (the same BAM ten times) through one `conga --cohort` process: the steady rate per genome
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/e2e_quick.sh > /dev/null 2>&1
cd /tmp/e2e_wg
for i in 0 1 2 3 4 5 6 7 8 9; do printf "r.bam\tk$i\n"; done > list10.txt
t0=$(date +%s%N)
env CONGA_TIMING=1 "$@" /root/repo/conga_amd/host/conga --cohort list10.txt --ref r.fa --sonic a.cga --dels dels.bed --out co > cohort10.log 2>&1
t1=$(date +%s%N)
grep -a "timing\] open\|waited" cohort10.log | sed 's/.*fetch + output/fetch + output/'
echo "cohort of 10: wall $(( (t1 - t0) / 1000000 )) ms"
md5sum k0_dels.bed k9_dels.bed o1_dels.bed | cut -c1-32 | sort -u | wc -l
