set +o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rp_ahead0 gpurun_out/rp_ahead1
for A in 0 1; do
  CONGA_BENCH_RP_K=12 CONGA_COHORT_AHEAD=$A CONGA_BENCH_STDERR_DIR=$GRAFT_REPO_ROOT/gpurun_out/rp_ahead$A timeout -k 10 500 python3 tools/sr_quick.py --rp-chroms all --steps 3 > gpurun_out/rp_ahead$A.json 2> gpurun_out/rp_ahead$A.err
  echo "AHEAD=$A: $(python3 -c "import json; d=json.load(open('gpurun_out/rp_ahead$A.json')); print(d.get('ms_per_step'), d.get('end_to_end',{}).get('first_sample_s'))")"
  grep -a "cohort: sample" gpurun_out/rp_ahead$A/conga_cohort_gpu.err | sed 's/.*cohort: //' | tr '\n' ';'
  echo
done
