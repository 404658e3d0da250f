#!/bin/bash
# packprobe_threads.sh -- bench.py's main leg by the producer's thread count, with the host thread's phases (stderr of bench.py)
cd "$(dirname "$0")/.." || exit 1
for round in 1 2; do
for t in 12 14 13 12 14 13; do
	printf 'threads %-3s' "$t"
	CONGA_BENCH_PACK_THREADS=$t CONGA_BENCH_PHASES=1 python bench.py --steps 40 --warmup 8 --cpu-seconds 0 --no-dense-leg --no-config-legs --no-e2e-leg 2>/tmp/pp.err |
		python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["hand_over"]; print("encode %.3f int32 %.3f pre %.3f" % (h["packed_encode_timed"]["ms_per_step"], h["int32"]["ms_per_step"], h["packed_preencoded"]["ms_per_step"]), end="  ")'
	grep -h "^\[phases\]" /tmp/pp.err | tail -1
done
done
