cd /root/repo
CONGA_DEBUG=1 CONGA_TIMING=1 CONGA_BENCH_PHASES=1 python bench.py --steps 200 --warmup 10 --cpu-seconds 0 --no-dense-leg --no-config-legs --no-e2e-leg --chroms 21 2>&1 >/dev/null | grep "phases\|timing" | head -12
