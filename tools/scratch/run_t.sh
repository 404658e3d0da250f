mkdir -p gpurun_out
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for rep in 1 2 3; do
for t in 14 12 10 8; do
  a=$(thr)
  CONGA_BENCH_PACK_THREADS=$t python bench.py --steps 20 --warmup 5 --no-e2e-leg --no-config-legs --no-dense-leg --cpu-seconds 0 > gpurun_out/t_$t.json 2>/dev/null
  b=$(thr)
  python3 - $t "$a" "$b" <<'PY'
import json,sys
j=json.loads(open("gpurun_out/t_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
h=j["hand_over"]
a=sys.argv[2].split(); b=sys.argv[3].split()
print("threads %s: encode %.3f int32 %.3f pre %.3f chosen %s | throttled +%d periods, +%.1f ms" % (sys.argv[1], h["packed_encode_timed"]["ms_per_step"], h["int32"]["ms_per_step"], h["packed_preencoded"]["ms_per_step"], h["chosen"], int(b[1])-int(a[1]), (int(b[3])-int(a[3]))/1e3))
PY
done
done
