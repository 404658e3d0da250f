set -e
mkdir -p gpurun_out
python -m pytest tests/test_host_cli.py tests/test_gpu_cohort.py -x -q -m gpu -k "cohort or named" > gpurun_out/g_tests.log 2>&1 || { tail -40 gpurun_out/g_tests.log; exit 1; }
tail -2 gpurun_out/g_tests.log
for i in 1 2 3 4 5 6 7 8; do
  python tools/cohort_trace.py --samples 8 --from-sample 1 --to-sample 3 > gpurun_out/trace_h$i.log 2>&1
  grep -E "^wall" gpurun_out/trace_h$i.log
done
