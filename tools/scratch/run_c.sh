set -e
mkdir -p gpurun_out
for g in 8 7 6 5; do
  python tools/cohort_trace.py --samples 12 --from-sample 8 --to-sample 9 --env CONGA_BGZF_GROUPS_PER_CU=$g > gpurun_out/trace_g$g.log 2>&1
  echo "== groups per CU $g"; grep -E "wall|inflate launch|launches are through|walks are through" gpurun_out/trace_g$g.log | head -12
done
