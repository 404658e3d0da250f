#!/bin/bash
# Sweep of the chain-class thresholds (env knobs of conga_api.hip: prepare) on the bench workload, records on the device.
run() {
  out=$(env "$@" python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-dense-leg --results-on-device 2>/dev/null)
  echo "$* $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["chain_ms"])')"
}
run A=0
for s in ${SERIALS:-40 48 52 60 64 72}; do run CONGA_CHAIN_SERIAL_WINDOWS=$s; done
for l in ${LONGS:-256 384 768 1024}; do run CONGA_CHAIN_LONG_WINDOWS=$l; done
for b in ${BLOCKS:-1024 1536 3072 4096}; do run CONGA_CHAIN_BLOCK_WINDOWS=$b; done
run A=0
