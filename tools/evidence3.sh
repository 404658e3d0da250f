#!/bin/bash
# Round-3 measurement campaign on the GPU box: tools/evidence3.sh TAG  ->  gpurun_out/TAG_* (what is to be judged is copied into profiles/)
#   bench line; rocprofv3 kernel-trace stats of the bench, of the split-read leg and of the inflate leg; FETCH_SIZE / WRITE_SIZE of the
#   bench's kernels; SQ counters of split_map_kernel and of bgzf_inflate_wave_kernel (one --pmc pass per counter set, never with a trace domain)
set -e
set +o pipefail
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --cpu-seconds 0 --no-config-legs --no-e2e-leg --steps 20 --warmup 5 > $O/${TAG}_bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_sr -- python3 $R/tools/sr_quick.py --no-cli --steps 5 > $O/${TAG}_sr_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_inflate -- python3 $R/tools/inflate_bench.py --kernels wave --sizes 0 --reps 4 > $O/${TAG}_inflate_under_rocprof.json
for d in stats stats_sr stats_inflate; do
  f=$(find $O/${TAG}_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_${d}.csv
  rm -rf $O/${TAG}_$d
done
echo "stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_$C -- python3 $R/bench.py --cpu-seconds 0 --no-config-legs --no-e2e-leg --steps 6 --warmup 3 > /dev/null
done
python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_fetch_write.json $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE > $O/${TAG}_pmc_summary.txt
rm -rf $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_srpmc_$C -- python3 $R/tools/sr_quick.py --no-cli --steps 3 > /dev/null
done
python3 $R/tools/pmc_summary.py $O/${TAG}_sr_pmc_fetch_write.json $O/${TAG}_srpmc_FETCH_SIZE $O/${TAG}_srpmc_WRITE_SIZE > $O/${TAG}_sr_pmc_summary.txt
rm -rf $O/${TAG}_srpmc_FETCH_SIZE $O/${TAG}_srpmc_WRITE_SIZE
echo "pmc done"
bash $R/tools/sq_pmc.sh ${TAG}_split_map split_map_kernel python3 $R/tools/sr_quick.py --no-cli --steps 3
bash $R/tools/inflate_pmc.sh ${TAG}
echo "sq done"
