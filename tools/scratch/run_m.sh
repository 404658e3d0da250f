set -e
mkdir -p gpurun_out
python -m pytest tests/test_host_cli.py tests/test_gpu_cohort.py -x -q -m gpu -k "cohort or named" > gpurun_out/m_tests.log 2>&1 || { tail -40 gpurun_out/m_tests.log; exit 1; }
tail -2 gpurun_out/m_tests.log
for i in 1 2 3 4 5 6 7 8 9 10; do
  python tools/cohort_trace.py --samples 6 --from-sample 1 --to-sample 6 > gpurun_out/trace_m$i.log 2>&1
  echo "== run $i $(grep -E '^wall' gpurun_out/trace_m$i.log) $(grep -c 'spare output set grows' gpurun_out/trace_m$i.log) growths"
done
