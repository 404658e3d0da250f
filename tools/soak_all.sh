#!/bin/bash
# soak_all.sh SEED [SECONDS_EACH] -- every mode of tests/soak.py one after the other on the GPU box, each for a while; -> gpurun_out/soak_all_<seed>.log
cd "$(dirname "$0")/.." || exit 1
S=${1:-1000}; T=${2:-150}
{
timeout -k 10 $((T + 120)) python tests/soak.py --ahead --cases 100000 --seed $((S + 1)) --seconds $T 2>&1 | tail -1
timeout -k 10 $((T + 120)) python tests/soak.py --bam --cases 100000 --seed $((S + 2)) --seconds $T 2>&1 | tail -1
timeout -k 10 $((T + 120)) python tests/soak.py --bam-rp --cases 100000 --seed $((S + 3)) --seconds $T 2>&1 | tail -1
timeout -k 10 $((T + 120)) python tests/soak.py --cases 100000 --seed $((S + 4)) --seconds $T 2>&1 | tail -1
timeout -k 10 $((T + 120)) python tests/soak.py --batch --cases 100000 --seed $((S + 5)) --seconds $T 2>&1 | tail -1
timeout -k 10 $((T + 120)) python tests/soak.py --packed --cases 100000 --seed $((S + 6)) --seconds $T 2>&1 | tail -1
timeout -k 10 $((T + 120)) python tests/soak.py --split-reads --cases 100000 --seed $((S + 7)) --seconds $T 2>&1 | tail -1
} > gpurun_out/soak_all_$S.log 2>&1
cat gpurun_out/soak_all_$S.log
