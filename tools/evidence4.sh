#!/bin/bash
# Round-4 measurement campaign on the GPU box: tools/evidence4.sh TAG  ->  gpurun_out/TAG_* (what is to be judged is copied into profiles/)
#   the bench line (N = 1) and, on the same box, the multi-rank code path with the one rank there is (--dist-selftest);
#   rocprofv3 kernel-trace stats of the bench, of the split-read leg and of the inflate leg; FETCH_SIZE / WRITE_SIZE of the bench's
#   kernels (one --pmc pass per counter, never with a trace domain besides --kernel-trace)
set -e
set +o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "bench done"
python3 $R/bench.py --steps 20 --warmup 5 --dist-selftest --no-e2e-leg --no-config-legs --no-dense-leg --cpu-seconds 0 > $O/${TAG}_bench_dist_selftest.json 2> /dev/null
echo "selftest done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --cpu-seconds 0 --no-config-legs --no-e2e-leg --steps 20 --warmup 5 > $O/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_sr -- python3 $R/tools/sr_quick.py --no-cli --steps 5 > $O/${TAG}_sr_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_inflate -- python3 $R/tools/inflate_bench.py --kernels wave --sizes 0 --reps 4 > $O/${TAG}_inflate_under_rocprof.json 2> /dev/null
for d in stats stats_sr stats_inflate; do
  f=$(find $O/${TAG}_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_${d}.csv
  rm -rf $O/${TAG}_$d
done
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_$C -- python3 $R/bench.py --cpu-seconds 0 --no-config-legs --no-e2e-leg --steps 6 --warmup 3 > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_fetch_write.json $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE > $O/${TAG}_pmc_summary.txt
rm -rf $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
echo "pmc done"
