set -e
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10; do
  python tools/cohort_trace.py --samples 6 --from-sample 1 --to-sample 6 > gpurun_out/trace_l$i.log 2>&1
  echo "== run $i $(grep -E '^wall' gpurun_out/trace_l$i.log)"
done
