#!/bin/bash
# `conga --cohort` on whole-genome 1x BAMs under alternating sets of measurement switches, N runs each, on ONE box (the steady state has
# modes that differ from run to run: profiles/r04j_cohort_two_modes.log):  tools/cohort_ab.sh RUNS "K=V,K=V" "K=V" ...   ("" = no switch)
# prints every run's sample ends; the traces (tools/cohort_trace.py) stay under gpurun_out/ab_<variant>_<run>.log
set -e
mkdir -p gpurun_out
runs=$1; shift
for i in $(seq 1 $runs); do
  k=0
  for v in "$@"; do
    k=$((k + 1))
    if [ -n "$v" ]; then e="--env $v"; else e=""; fi
    python tools/cohort_trace.py --samples 12 --from-sample 11 --to-sample 12 $e > gpurun_out/ab_${k}_$i.log 2>&1
    echo "run $i [${v:-no switch}] $(grep -E '^wall' gpurun_out/ab_${k}_$i.log)"
  done
done
