#!/bin/bash
# Round-2 measurement campaign on the GPU box: tools/evidence2.sh TAG  ->  gpurun_out/TAG_* (what is to be judged is copied into profiles/)
set -e
set +o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --cpu-seconds 0 --no-config-legs --steps 20 --warmup 5 > $O/${TAG}_bench_under_rocprof.json
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_$C -- python3 $R/bench.py --cpu-seconds 0 --no-config-legs --steps 6 --warmup 3 > /dev/null
done
python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_fetch_write.json $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE > $O/${TAG}_pmc_summary.txt
echo "pmc done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetchcal -- $R/tools/fetchcal > $O/${TAG}_fetchcal_bytes.json
python3 $R/tools/pmc_summary.py $O/${TAG}_fetchcal.json $O/${TAG}_fetchcal > $O/${TAG}_fetchcal_summary.txt
echo "fetchcal done"
