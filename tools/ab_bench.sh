#!/bin/bash
# A/B of two builds of the library on the same box: tools/ab_bench.sh <base.so> <new.so> [bench args]
# Alternates the two, three rounds each, device-resident records and host-delivered records.
base=$1; new=$2; shift 2
for round in 1 2 3; do
  for lib in "$base" "$new"; do
    for mode in "--results-on-device" ""; do
      out=$(CONGA_LIB_PATH=$lib python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-dense-leg $mode "$@" 2>/dev/null)
      echo "$(basename $lib) ${mode:-host-records} $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["chain_ms"], d["roofline"]["avg_launch_ms"])')"
    done
  done
done
