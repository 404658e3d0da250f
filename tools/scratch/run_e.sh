set -e
mkdir -p gpurun_out
python tools/cohort_trace.py --samples 14 --from-sample 9 --to-sample 10 --env CONGA_BGZF_AHEAD_FOLLOW=1 > gpurun_out/trace_e0.log 2>&1
grep -E "^wall" gpurun_out/trace_e0.log
python - <<'PY'
import re,sys
t=open("gpurun_out/trace_e0.log").read()
ends=[float(x) for x in re.search(r"sample ends \(ms\): (.*)", t).group(1).split()]
d=sorted(b-a for a,b in zip(ends[4:-1],ends[5:]))
med=d[len(d)//2]
print("median period", med)
sys.exit(0)
PY
i=1
for v in "CONGA_BGZF_AHEAD_FOLLOW=1" "CONGA_BGZF_AHEAD_FOLLOW=0" "CONGA_BGZF_AHEAD_FOLLOW=0" "CONGA_BGZF_AHEAD_FOLLOW=1" "CONGA_BGZF_AHEAD_FOLLOW=0" "CONGA_BGZF_AHEAD_FOLLOW=1"; do
  python tools/cohort_trace.py --samples 14 --from-sample 9 --to-sample 10 --env $v > gpurun_out/trace_e$i.log 2>&1
  echo "== $v"; grep -E "^wall|overlapped upload" gpurun_out/trace_e$i.log | cut -c1-260 | head -3
  i=$((i+1))
done
