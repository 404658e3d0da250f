set -e
mkdir -p gpurun_out
for i in $(seq 1 24); do
  case $((i % 3)) in
    0) v="--env CONGA_BGZF_SLOT_MB=16"; n="slot16";;
    1) v=""; n="base";;
    2) v="--env CONGA_BGZF_COPY_STREAMS=2"; n="two-streams";;
  esac
  python tools/cohort_trace.py --samples 10 --from-sample 9 --to-sample 10 $v > gpurun_out/trace_n$i.log 2>&1
  echo "== run $i $n $(grep -E '^wall' gpurun_out/trace_n$i.log)"
done
