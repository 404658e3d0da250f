#!/bin/bash
# The upload of a cohort on rocprofv3's time line: `conga --cohort` over twelve whole-genome 1x BAMs under --kernel-trace --memory-copy-trace
# (no counters), then per 25 ms window the host-to-device copies of 0.08 ms and more (the ring's 8 MB pieces take 0.153 ms each): how many,
# their median duration, how long the copy engine was busy.  tools/cohort_copy_trace.sh [RUNS]  ->  gpurun_out/copies_<run>.txt
set -e
mkdir -p gpurun_out
R=$PWD
python tools/cohort_trace.py --samples 12 --keep /tmp/ck > gpurun_out/ck.log 2>&1
cd /tmp/ck && export TMPDIR=/tmp
export CONGA_GPU_BAM=1 CONGA_CLEAN_EXIT=1 CONGA_TIMING=1
for i in $(seq 1 ${1:-3}); do
  rm -rf /tmp/ckprof
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ckprof -- $R/conga_amd/host/conga --cohort list.txt --out x --ref r.fa --sonic a.cga --dels dels.bed > /tmp/ck/run$i.log 2>&1
  k=$(find /tmp/ckprof -name "*kernel_trace.csv" | head -1)
  m=$(find /tmp/ckprof -name "*memory_copy_trace.csv" | head -1)
  python3 - "$k" "$m" /tmp/ck/run$i.log > $R/gpurun_out/copies_$i.txt <<'PY'
import csv, sys, re
k = list(csv.DictReader(open(sys.argv[1])))
m = list(csv.DictReader(open(sys.argv[2])))
err = open(sys.argv[3]).read()
done = [float(x) for x in re.findall(r"cohort: sample \d+ of \d+ is done ([0-9.]+) ms", err)]
print("sample ends:", " ".join("%.0f" % x for x in done))
infl = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in k if "bgzf_inflate_wave" in r["Kernel_Name"])
print("inflate launches:", len(infl), "durations ms:", " ".join("%.1f" % ((b - a) / 1e6) for a, b in infl[-8:]))
h2d = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in m if "HOST_TO_DEVICE" in r["Direction"].upper() and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 80000), key=lambda x: x[0])
print("H2D copies of 0.08 ms and more:", len(h2d))
if h2d:
    t0, w = h2d[0][0], 25e6
    for i in range(int((h2d[-1][1] - t0) / w) + 1):
        grp = [(a, b) for a, b in h2d if t0 + i * w <= a < t0 + (i + 1) * w]
        if grp:
            dur = sorted((b - a) / 1e6 for a, b in grp)
            print("window %3d (%4.0f ms): %3d copies, median %.3f ms, p90 %.3f, busy %.1f of 25 ms" % (i, i * 25, len(grp), dur[len(dur) // 2], dur[int(len(dur) * 0.9)], sum(dur)))
PY
  head -3 $R/gpurun_out/copies_$i.txt | cut -c1-200
done
