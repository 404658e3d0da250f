cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/e2e_quick.sh > /dev/null 2>&1   # builds the inputs, warms the page cache
cd /tmp/e2e_wg
for mode in default "HSA_ENABLE_SDMA=0" "CONGA_BGZF_UPLOAD_ONLY=1" "HSA_ENABLE_SDMA=0 CONGA_BGZF_UPLOAD_ONLY=1" "CONGA_BGZF_NO_PRIORITY=1" ; do
  for i in 1 2; do
    if [ "$mode" = default ]; then e=""; else e="$mode"; fi
    env CONGA_DEBUG=1 CONGA_TIMING=1 $e /root/repo/conga_amd/host/conga -i r.bam --ref r.fa --sonic a.cga --dels dels.bed --out x > x.log 2>&1
    echo "[$mode] $(grep -a 'overlapped upload' x.log | sed 's/.*enqueued after/enqueued after/') | $(grep -a 'conga_reads_bgzf' x.log | sed 's/.*buffers/buffers/')"
  done
done
