set -e
mkdir -p gpurun_out
export CONGA_DEBUG=1 CONGA_BGZF_TRACE=1 CONGA_BENCH_STDERR_DIR=$PWD/gpurun_out
python bench.py --steps 5 --warmup 2 --no-config-legs --no-dense-leg --cpu-seconds 0 > gpurun_out/p_bench.json 2> gpurun_out/p_bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/p_bench.json").read().strip().splitlines()[-1])
print(j["end_to_end"]["end_to_end"])
PY
grep -E "is done|spare output set grows|told to go|has grown" gpurun_out/conga_cohort_gpu.err | cut -c1-160 | head -60
