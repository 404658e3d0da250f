mkdir -p gpurun_out
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for rep in 1 2 3 4; do
for v in "14 20000" "14 2000" "14 0" "12 20000" "12 2000"; do
  set -- $v
  a=$(thr)
  CONGA_DEBUG=1 CONGA_PACK_SPIN=$2 CONGA_BENCH_PACK_THREADS=$1 python bench.py --steps 20 --warmup 5 --no-e2e-leg --no-config-legs --no-dense-leg --cpu-seconds 0 > gpurun_out/u.json 2>/dev/null
  b=$(thr)
  python3 - "$v" "$a" "$b" <<'PY'
import json,sys
j=json.loads(open("gpurun_out/u.json").read().strip().splitlines()[-1])
h=j["hand_over"]
a=sys.argv[2].split(); b=sys.argv[3].split()
print("threads/spin %-9s: encode %.3f int32 %.3f pre %.3f | throttled +%d periods" % (sys.argv[1], h["packed_encode_timed"]["ms_per_step"], h["int32"]["ms_per_step"], h["packed_preencoded"]["ms_per_step"], int(b[1])-int(a[1])))
PY
done
done
