#!/bin/bash
# One measurement campaign on the GPU box: tools/evidence.sh TAG  ->  gpurun_out/TAG_* (copy what is to be judged into profiles/)
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --cpu-seconds 0 --steps 20 > $O/${TAG}_bench_under_rocprof.json
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $R/bench.py --cpu-seconds 0 --steps 3 --warmup 1 > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/bench.py --cpu-seconds 0 --steps 3 --warmup 1 > /dev/null
python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_fetch_write.json $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write > $O/${TAG}_pmc_summary.txt
echo "pmc done"
python3 $R/bench.py --config "dels+dups+map" --cpu-seconds 0 > $O/${TAG}_bench_configs2.json
python3 $R/bench.py --results-on-device --cpu-seconds 0 --no-dense-leg > $O/${TAG}_bench_results_on_device.json
python3 $R/bench.py --dist-selftest --cpu-seconds 0 --no-dense-leg > $O/${TAG}_bench_dist_selftest.json
python3 $R/bench.py --cov 5 --cpu-seconds 0 > $O/${TAG}_bench_cov5.json
echo "cov5 done"
python3 $R/bench.py --cov 30 --cpu-seconds 0 --steps 10 > $O/${TAG}_bench_cov30.json
echo "cov30 done"
