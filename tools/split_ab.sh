#!/bin/bash
# split_map_kernel with and without the presence bitmaps / the XCD-aware unit order: kernel time by HIP events and FETCH_SIZE +
# L2 hits / misses per launch (one rocprofv3 --pmc pass each).  tools/split_ab.sh TAG [sr_quick args]  ->  gpurun_out/TAG_split_ab.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export CONGA_DEBUG=1
: > $O/${TAG}_split_ab.txt
for F in 0 1 2 3; do
  export CONGA_SPLIT_FLAGS=$F
  python3 $R/tools/sr_quick.py --no-cli --steps 5 "$@" > $O/${TAG}_ab_$F.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/${TAG}_ab_$F.json'))
print('flags $F: kernel %.4f ms (HIP events), %d algorithmic bytes' % (d['roofline']['avg_launch_ms'], d['roofline']['algorithmic_bytes_per_launch']))
" >> $O/${TAG}_split_ab.txt
  for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf $O/${TAG}_abp
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_abp -- python3 $R/tools/sr_quick.py --no-cli --steps 3 "$@" > /dev/null 2>&1
    f=$(find $O/${TAG}_abp -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY' >> $O/${TAG}_split_ab.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "split_map_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("   %-16s %16.0f per launch (avg of %d)" % (k, v / max(n, 1), n))
PY
    rm -rf $O/${TAG}_abp
  done
  rm -f $O/${TAG}_ab_$F.json
done
cat $O/${TAG}_split_ab.txt
