mkdir -p gpurun_out
f() { "$@" ./tools/packbench 14 2>&1 | head -1 | awk '{print $5}'; }
{
echo "# tools/packbench 14 (fourteen threads, best of 12 samples of 25.6 M positions, ms) on a GPU box: 2 x EPYC 9575F, 256 CPUs in two memory nodes, the"
echo "# process may run on all of them for 16 CPUs' worth of time.  'follows' = the packer's workers keep to the memory node the positions lie on"
echo "# (pack_host.h: node_of / want_node_); 'anywhere' = CONGA_DEBUG=1 CONGA_PACK_NO_NUMA=1; node0 / node1 = the whole process under taskset"
for rep in 1 2 3 4 5 6 7 8; do
  a=$(f env CONGA_DEBUG=1 CONGA_PACK_NO_NUMA=1); b=$(f env); c=$(f taskset -c 0-63,128-191); d=$(f taskset -c 64-127,192-255)
  echo "anywhere $a   follows $b   node0 $c   node1 $d"
done
} > gpurun_out/r04k_packbench_numa.log 2>&1
cat gpurun_out/r04k_packbench_numa.log
for rep in 1 2 3 4 5 6; do
  for v in 0 1; do
    if [ $v = 1 ]; then export CONGA_DEBUG=1 CONGA_PACK_NO_NUMA=1; else unset CONGA_DEBUG CONGA_PACK_NO_NUMA; fi
    python bench.py --steps 20 --warmup 5 --no-e2e-leg --no-config-legs --no-dense-leg --cpu-seconds 0 > gpurun_out/w.json 2>/dev/null
    python3 - $v <<'PY'
import json,sys
j=json.loads(open("gpurun_out/w.json").read().strip().splitlines()[-1])
h=j["hand_over"]
print("bench, workers %s: encode %.3f int32 %.3f pre %.3f" % ("anywhere" if sys.argv[1]=="1" else "follow  ", h["packed_encode_timed"]["ms_per_step"], h["int32"]["ms_per_step"], h["packed_preencoded"]["ms_per_step"]))
PY
  done
done
