#!/bin/bash
# split_map_kernel on the WHOLE 5x genome (131.5 M records, conga_amd/rp_bench.py with --rp-chroms all; ~60 GB of host memory, a few
# minutes): its duration by HIP events, and FETCH_SIZE / WRITE_SIZE per launch (one rocprofv3 --pmc pass each)
#   tools/split_wg_pmc.sh TAG  ->  gpurun_out/TAG_split_wg.txt + TAG_split_wg.json
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/sr_quick.py --no-cli --steps 3 --rp-chroms all > $O/${TAG}_split_wg.json 2>/dev/null
python3 -c "
import json
d=json.load(open('$O/${TAG}_split_wg.json'))
r=d['roofline']
print('whole genome: split_map_kernel %.3f ms (HIP events), %d records, %d elements, %d algorithmic bytes' % (r['avg_launch_ms'], d['records'], d['split_elements'], r['algorithmic_bytes_per_launch']))
" > $O/${TAG}_split_wg.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${TAG}_wgp
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_wgp -- python3 $R/tools/sr_quick.py --no-cli --steps 2 --rp-chroms all > /dev/null 2>&1
  f=$(find $O/${TAG}_wgp -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $O/${TAG}_split_wg.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "split_map_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("   %-12s %16.0f KB per launch (avg of %d launches)" % (k, v / max(n, 1), n))
PY
  rm -rf $O/${TAG}_wgp
done
cat $O/${TAG}_split_wg.txt
