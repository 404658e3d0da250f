set -e
mkdir -p gpurun_out
python -m pytest tests/test_host_cli.py tests/test_gpu_cohort.py -x -q -m gpu -k "cohort or named or pipeline" > gpurun_out/r_tests.log 2>&1 || { tail -40 gpurun_out/r_tests.log; exit 1; }
tail -2 gpurun_out/r_tests.log
export CONGA_DEBUG=1 CONGA_BGZF_TRACE=1 CONGA_BENCH_STDERR_DIR=$PWD/gpurun_out
python bench.py --steps 5 --warmup 2 --no-config-legs --no-dense-leg --cpu-seconds 0 > gpurun_out/r_bench.json 2> gpurun_out/r_bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r_bench.json").read().strip().splitlines()[-1])
print(j["end_to_end"]["end_to_end"])
PY
grep -E "spare output set grows|has grown" gpurun_out/conga_cohort_gpu.err | cut -c1-160 | head
grep -E "is done" gpurun_out/conga_cohort_gpu.err | awk '{print $9}' | tr '\n' ' '
