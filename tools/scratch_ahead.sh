cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_cohort.py tests/test_gpu_parity.py -x -q > gpurun_out/ahead_tests.log 2>&1 || { tail -30 gpurun_out/ahead_tests.log; exit 1; }
tail -2 gpurun_out/ahead_tests.log
one() { printf '%-28s' "$1"; shift; env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 8 --cpu-seconds 0 --no-dense-leg --no-config-legs --no-e2e-leg $ARGS 2>/tmp/e.err |
	python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d["hand_over"]; print("encode %.3f int32 %.3f pre %.3f  chain %.4f" % (h["packed_encode_timed"]["ms_per_step"], h["int32"]["ms_per_step"], h["packed_preencoded"]["ms_per_step"], d["roofline"]["kernel_ms_per_step"]["interval_chain"]))'; grep -h "phases" /tmp/e.err | sed -n '1p;3p' | cut -c1-200; }
export CONGA_BENCH_PHASES=1
for ARGS in "--chroms 21" "--chroms 1 --dist-selftest" ""; do
  echo "== bench $ARGS"
  for r in 1 2; do
    one "two computes in flight" X=1
    one "round 4's order" CONGA_BENCH_NO_AHEAD=1
  done
done
