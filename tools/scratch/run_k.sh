set -e
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18; do
  case $((i % 3)) in
    0) v="--env CONGA_BGZF_SLOT_MB=16"; n="slot16";;
    1) v=""; n="base";;
    2) v="--env GPU_MAX_HW_QUEUES=8"; n="hwq8";;
  esac
  python tools/cohort_trace.py --samples 10 --from-sample 1 --to-sample 10 $v > gpurun_out/trace_k$i.log 2>&1
  echo "== run $i $n $(grep -E '^wall' gpurun_out/trace_k$i.log)"
  grep -E "every piece is enqueued" gpurun_out/trace_k$i.log | sed -n '6,6p' | cut -c60-420
done
