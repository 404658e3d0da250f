#!/bin/bash
# packprobe_bench.sh -- bench.py's main leg (producer timed inside the step) with the producer's switches, alternating on one box.
cd "$(dirname "$0")/.." || exit 1
export CONGA_DEBUG=1
one() { printf '%-40s' "$1"; shift; env "$@" python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --no-dense-leg --no-config-legs --no-e2e-leg 2>/dev/null |
	python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["hand_over"]; print("encode %.3f int32 %.3f pre %.3f" % (h["packed_encode_timed"]["ms_per_step"], h["int32"]["ms_per_step"], h["packed_preencoded"]["ms_per_step"]))'; }
for round in 1 2 3; do
	one "default" X=1
	one "round 4's producer" CONGA_PACK_NO_STREAM=1 CONGA_PACK_PREFETCH=0 CONGA_PACK_BATCH=1
	one "plain stores" CONGA_PACK_NO_STREAM=1
	one "no requests ahead" CONGA_PACK_PREFETCH=0
done
