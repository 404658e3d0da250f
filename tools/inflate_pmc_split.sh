#!/bin/bash
# SQ counters of bgzf_inflate_wave_kernel per DISPATCH of tools/inflate_ab.py (its first four launches inflate the random-quality
# file, the last four the run-structured one): tools/inflate_pmc_split.sh TAG  ->  gpurun_out/TAG_inflate_pmc_split.txt
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_inflate_pmc_split
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p -- python3 $R/tools/inflate_ab.py > $O/p.log 2>&1 || { echo "pass failed"; tail -3 $O/p.log; }
f=$(find $O/p -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/${TAG}_inflate_pmc_split.txt
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "inflate_wave" in r["Kernel_Name"]:
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (d, c) in enumerate(rows.items()):
    busy = c.get("SQ_BUSY_CYCLES", 0) / 32.0   # summed over the 32 shader engines' sequencers
    valu = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024.0
    sca = c.get("SQ_ACTIVE_INST_SCA", 0) * 4 / 1024.0
    print("dispatch %s (%s): cycles %.1f M, vector busy %.2f, scalar busy %.2f, VALU insts %.2f G, SALU insts %.2f G, wave cycles waiting on an instruction %.2f"
          % (d, "random" if i < len(rows) // 2 else "run-structured", busy / 1e6, valu / max(busy, 1), sca / max(busy, 1),
             c.get("SQ_INSTS_VALU", 0) / 1e9, c.get("SQ_INSTS_SALU", 0) / 1e9, c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
rm -rf $O
