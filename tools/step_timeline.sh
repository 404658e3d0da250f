#!/bin/bash
# step_timeline.sh -- one small step on rocprofv3's time line (kernels + copies, no counters): where a step's time goes when the copy is
# small (a rank's share on eight GPUs, or one chromosome).  tools/step_timeline.sh [bench args]  ->  gpurun_out/step_timeline.txt
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stprof
rocprofv3 --kernel-trace --memory-copy-trace ${HIPTRACE:+--hip-runtime-trace} --output-format csv -d /tmp/stprof -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-dense-leg --no-config-legs --no-e2e-leg "$@" > /tmp/st.json 2> /tmp/st.err
k=$(find /tmp/stprof -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/stprof -name "*memory_copy_trace.csv" | head -1)
h=$(find /tmp/stprof -name "*hip_api_trace.csv" | head -1)
python3 - "$k" "$m" $h > $R/gpurun_out/step_timeline${TAG}.txt <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], "q" + r.get("Queue_Id", "?")))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"], ""))
if len(sys.argv) > 3:   # the host's calls (HIPTRACE=1): which thread, how long
    for r in csv.DictReader(open(sys.argv[3])):
        if r["Function"] in ("__hipPushCallConfiguration", "__hipPopCallConfiguration"):
            continue
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "    host t%s %s" % (r["Thread_Id"][-3:], r["Function"]), ""))
ev.sort()
# the first timed leg: find the 12th..16th chain launch and print everything from there for 5 steps
chains = [i for i, e in enumerate(ev) if "interval_chain_kernel" in e[2]]
print("events", len(ev), "chain launches", len(chains))
for lo_c, hi_c in ((12, 14),):
    if len(chains) > hi_c:
        lo, hi = chains[lo_c], chains[hi_c]
        t0 = ev[lo][0]
        print("--- from chain launch", lo_c)
        prev_end = t0
        for a, b, name, q in ev[lo:hi + 1]:
            print("%9.1f us  +%7.1f us  (gap %7.1f)  %s %s" % ((a - t0) / 1e3, (b - a) / 1e3, (a - prev_end) / 1e3, name, q))
            prev_end = max(prev_end, b)
PY
tail -3 /tmp/st.err >> $R/gpurun_out/step_timeline${TAG}.txt
