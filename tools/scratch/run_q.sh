set -e
mkdir -p gpurun_out
slow=0
for i in 1 2 3; do
  python tools/cohort_trace.py --samples 12 --from-sample 11 --to-sample 12 > gpurun_out/trace_q$i.log 2>&1
  echo "== probe $i base $(grep -E '^wall' gpurun_out/trace_q$i.log)"
  python3 - gpurun_out/trace_q$i.log <<'PY' || slow=$((slow+1))
import re,sys
t=open(sys.argv[1]).read()
e=[float(x) for x in re.search(r"sample ends \(ms\): (.*)", t).group(1).split()]
d=sorted(b-a for a,b in zip(e[3:-1],e[4:]))
sys.exit(1 if d[len(d)//2] > 44 else 0)
PY
done
echo "slow probes: $slow"
[ $slow -ge 1 ] || exit 3
for i in $(seq 4 21); do
  case $((i % 3)) in
    0) v="--env CONGA_BGZF_SLOT_MB=32"; n="slot32";;
    1) v=""; n="base";;
    2) v="--env CONGA_BGZF_SLOT_MB=16"; n="slot16";;
  esac
  python tools/cohort_trace.py --samples 12 --from-sample 11 --to-sample 12 $v > gpurun_out/trace_q$i.log 2>&1
  echo "== run $i $n $(grep -E '^wall' gpurun_out/trace_q$i.log)"
done
