set -e
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  python tools/cohort_trace.py --samples 6 --from-sample 1 --to-sample 3 > gpurun_out/trace_f$i.log 2>&1
  grep -E "^wall" gpurun_out/trace_f$i.log
done
