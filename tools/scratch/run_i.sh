set -e
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
  python tools/cohort_trace.py --samples 10 --from-sample 1 --to-sample 10 > gpurun_out/trace_i$i.log 2>&1
  grep -E "^wall" gpurun_out/trace_i$i.log
  grep -E "every piece is enqueued" gpurun_out/trace_i$i.log | cut -c1-200
done
