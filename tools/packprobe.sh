#!/bin/bash
# packprobe.sh -- the producer's switches against each other on one box, alternating (tools/packbench --rotate 3, fourteen threads):
# non-temporal stores, the requests made ahead, the runs a worker takes at a time.   tools/packprobe.sh > gpurun_out/packprobe.log
cd "$(dirname "$0")/.." || exit 1
export CONGA_DEBUG=1
T=${1:-14}
run() { printf '%-44s' "$1"; shift; env "$@" tools/packbench --rotate 3 "$T" | sed 's/.*): //; s/ (width.*//'; }
for round in 1 2 3; do
	run "default (stream, ahead 2048, batch 8)" X=1
	run "round 4's producer (plain, none, 1)" CONGA_PACK_NO_STREAM=1 CONGA_PACK_PREFETCH=0 CONGA_PACK_BATCH=1
	run "plain stores" CONGA_PACK_NO_STREAM=1
	run "no requests ahead" CONGA_PACK_PREFETCH=0
	run "batch 1" CONGA_PACK_BATCH=1
	run "ahead 256" CONGA_PACK_PREFETCH=256
	run "ahead 512" CONGA_PACK_PREFETCH=512
	run "ahead 1024" CONGA_PACK_PREFETCH=1024
	run "ahead 4096" CONGA_PACK_PREFETCH=4096
	run "batch 32" CONGA_PACK_BATCH=32
done
