# the whole-genome --rp cohort (twelve samples): tools/rp_ahead.sh [variants...]   a variant is "default" (the CLI's own choice: one sample
# named ahead, its bytes brought up but not inflated ahead) or a value of CONGA_COHORT_AHEAD (0: none named; 1, 2)
set +o pipefail
cd ${GRAFT_REPO_ROOT:-/root/repo}
for V in ${@:-0 default}; do
  X=""
  [ "$V" != "default" ] && X="CONGA_COHORT_AHEAD=$V"
  mkdir -p gpurun_out/rp_ahead_$V
  env $X CONGA_BENCH_RP_K=12 CONGA_BENCH_STDERR_DIR=$PWD/gpurun_out/rp_ahead_$V timeout -k 10 500 python3 tools/sr_quick.py --rp-chroms all --steps 3 > gpurun_out/rp_ahead_$V.json 2> gpurun_out/rp_ahead_$V.err
  echo "$V: $(python3 -c "import json; d=json.load(open('gpurun_out/rp_ahead_$V.json')); print('per further sample', d.get('ms_per_step'), 'ms; first sample', d.get('end_to_end',{}).get('first_sample_s'), 's')")"
  grep -a "cohort: sample" gpurun_out/rp_ahead_$V/conga_cohort_gpu.err | sed 's/.*cohort: //' | tr '\n' ';'
  echo
done
