set -e
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
  v=""; [ $((i % 2)) = 0 ] && v="--env CONGA_BGZF_SLOT_SPIN=1"
  python tools/cohort_trace.py --samples 10 --from-sample 1 --to-sample 10 $v > gpurun_out/trace_j$i.log 2>&1
  echo "== run $i $v"
  grep -E "^wall" gpurun_out/trace_j$i.log
  grep -E "every piece is enqueued" gpurun_out/trace_j$i.log | sed -n '5,7p' | cut -c1-420
done
