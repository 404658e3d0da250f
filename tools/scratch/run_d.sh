set -e
mkdir -p gpurun_out
i=0
for v in "CONGA_BGZF_AHEAD_POLICY=0" "CONGA_BGZF_AHEAD_POLICY=0,CONGA_BGZF_AHEAD_ONE_STREAM=1" "CONGA_BGZF_AHEAD_POLICY=1" "CONGA_BGZF_AHEAD_POLICY=2" "CONGA_BGZF_AHEAD_POLICY=1,CONGA_BGZF_AHEAD_ONE_STREAM=1" "CONGA_BGZF_AHEAD_POLICY=0"; do
  python tools/cohort_trace.py --samples 14 --from-sample 9 --to-sample 10 --env $v > gpurun_out/trace_v$i.log 2>&1
  echo "== $v"; grep -E "^wall|overlapped upload" gpurun_out/trace_v$i.log | cut -c1-260 | head -3
  i=$((i+1))
done
