set -e
mkdir -p gpurun_out
R=$PWD
python tools/cohort_trace.py --samples 12 --keep /tmp/ck > gpurun_out/ck.log 2>&1
cd /tmp/ck && export TMPDIR=/tmp
export CONGA_GPU_BAM=1 CONGA_CLEAN_EXIT=1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ckprof -- $R/conga_amd/host/conga --cohort list.txt --out x --ref r.fa --sonic a.cga --dels dels.bed > /tmp/ck/run.log 2>&1
cp /tmp/ck/run.log $R/gpurun_out/ck_run.log; f=$(find /tmp/ckprof -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/ck_kernel_stats.csv
t=$(find /tmp/ckprof -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $R/gpurun_out/ck_trace_tail.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last third of the run: one line per kernel
n = len(rows)
for r in rows[int(n * 0.6):int(n * 0.6) + 260]:
    print("%10.3f %9.3f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:90]))
PY
head -30 $R/gpurun_out/ck_kernel_stats.csv | cut -c1-200
