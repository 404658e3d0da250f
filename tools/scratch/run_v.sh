mkdir -p gpurun_out
f() { "$@" ./tools/packbench 14 2>&1 | head -1 | awk '{print $5}'; }
for rep in 1 2 3 4 5 6; do
  a=$(f env); b=$(f taskset -c 0-63,128-191); c=$(f taskset -c 64-127,192-255); d=$(f taskset -c 0-31); e=$(f taskset -c 0-13)
  echo "14 threads, best of 12 (ms): anywhere $a   node0 $b   node1 $c   cpus 0-31 $d   cpus 0-13 $e"
done
