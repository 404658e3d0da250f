set -e
mkdir -p gpurun_out
python -m pytest tests/test_host_cli.py -x -q -m gpu -k "cohort" > gpurun_out/a_tests.log 2>&1 || { tail -40 gpurun_out/a_tests.log; exit 1; }
tail -3 gpurun_out/a_tests.log
python tools/ctp_cohort.py --samples 8 > gpurun_out/ctp1.log 2>&1
cat gpurun_out/ctp1.log | cut -c1-420
