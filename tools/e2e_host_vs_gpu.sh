#!/bin/bash
# whole-genome CLI run with the BAM decoded by the host decoders against the run that decodes on the GPU: same files
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/e2e_quick.sh > /dev/null 2>&1
cd /tmp/e2e_wg
t0=$(date +%s%N)
env CONGA_GPU_BAM=0 /root/repo/conga_amd/host/conga -i r.bam --ref r.fa --sonic a.cga --dels dels.bed --out host > host.log 2>&1
t1=$(date +%s%N)
echo "host decoders: wall $(( (t1 - t0) / 1000000 )) ms"
md5sum host_dels.bed o1_dels.bed c3_dels.bed host_svs.bed o1_svs.bed c3_svs.bed
cmp host_dels.bed o1_dels.bed && cmp host_svs.bed o1_svs.bed && cmp host_dels.bed c3_dels.bed && echo "identical"
