#!/bin/bash
# The step's copies on a time line: rocprofv3 --kernel-trace --memory-copy-trace of a short bench run; the large host-to-device
# copies' durations and the gaps between them -> gpurun_out/TAG_copy_gaps.txt
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_copytrace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/p -- python3 $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-dense-leg --no-config-legs --no-e2e-leg > $O/bench.json 2> $O/bench.err || { echo "trace failed"; tail -3 $O/bench.err; }
f=$(find $O/p -name "*memory_copy_trace.csv" | head -1)
k=$(find $O/p -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$k" <<'PY' | tee $R/gpurun_out/${TAG}_copy_gaps.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", list(rows[0].keys()))
big = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    size = int(r.get("Bytes", r.get("Size", 0)) or 0) if ("Bytes" in r or "Size" in r) else 0
    big.append((s, e, r.get("Direction", r.get("Kind", "")), size))
big.sort()
# the steps' copies: host-to-device, tens of MB (or, without sizes, ~0.5 ms long)
sel = [b for b in big if ("HOST_TO_DEVICE" in b[2].upper() or "H2D" in b[2].upper()) and ((b[3] > 20e6) if b[3] else (b[1] - b[0] > 300e3))]
print(len(rows), "copies,", len(sel), "large host-to-device ones")
prev = None
out = []
for s, e, d, size in sel:
    gap = (s - prev) / 1e3 if prev is not None else None
    out.append((round((e - s) / 1e3, 1), None if gap is None else round(gap, 1), size))
    prev = e
for o in out[-40:]:
    print("copy %.1f us  gap before it %s us  bytes %d" % (o[0], o[1], o[2]))
PY
