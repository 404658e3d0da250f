#!/bin/bash
# soak_all.sh SEED [SECONDS_EACH] -- every mode of tests/soak.py one after the other on the GPU box, each for a while; -> gpurun_out/soak_all_<seed>.log
# (every mode's whole output is kept beside it, gpurun_out/soak_all_<seed>_<mode>.log: a failure's text is the last thing a run prints, and
#  a `tail -1` once threw the only copy of one away)
cd "$(dirname "$0")/.." || exit 1
S=${1:-1000}; T=${2:-150}
{
timeout -k 10 $((T + 120)) python tests/soak.py --ahead --cases 100000 --seed $((S + 1)) --seconds $T > gpurun_out/soak_all_${S}_ahead.log 2>&1; tail -1 gpurun_out/soak_all_${S}_ahead.log
timeout -k 10 $((T + 120)) python tests/soak.py --bam --cases 100000 --seed $((S + 2)) --seconds $T > gpurun_out/soak_all_${S}_bam.log 2>&1; tail -1 gpurun_out/soak_all_${S}_bam.log
timeout -k 10 $((T + 120)) python tests/soak.py --bam-rp --cases 100000 --seed $((S + 3)) --seconds $T > gpurun_out/soak_all_${S}_bam_rp.log 2>&1; tail -1 gpurun_out/soak_all_${S}_bam_rp.log
timeout -k 10 $((T + 120)) python tests/soak.py --cases 100000 --seed $((S + 4)) --seconds $T > gpurun_out/soak_all_${S}_cases.log 2>&1; tail -1 gpurun_out/soak_all_${S}_cases.log
timeout -k 10 $((T + 120)) python tests/soak.py --batch --cases 100000 --seed $((S + 5)) --seconds $T > gpurun_out/soak_all_${S}_batch.log 2>&1; tail -1 gpurun_out/soak_all_${S}_batch.log
timeout -k 10 $((T + 120)) python tests/soak.py --packed --cases 100000 --seed $((S + 6)) --seconds $T > gpurun_out/soak_all_${S}_packed.log 2>&1; tail -1 gpurun_out/soak_all_${S}_packed.log
timeout -k 10 $((T + 120)) python tests/soak.py --split-reads --cases 100000 --seed $((S + 7)) --seconds $T > gpurun_out/soak_all_${S}_split_reads.log 2>&1; tail -1 gpurun_out/soak_all_${S}_split_reads.log
} > gpurun_out/soak_all_$S.log 2>&1
cat gpurun_out/soak_all_$S.log
