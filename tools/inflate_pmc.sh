#!/bin/bash
# SQ / memory-pipe counters of the BGZF inflate kernel (tools/inflate_bench.py, every block of a three-chromosome BAM);
# one rocprofv3 pass per counter set, summaries under gpurun_out/<tag>_inflate_pmc/
set -e
TAG=${1:-r02c}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_inflate_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P3="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3" "${@:2}"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -- python3 $R/tools/inflate_bench.py --kernels wave --sizes 0 --reps 2 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "inflate_wave" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:32s} {v / max(n, 1):16.0f}  (avg of {n} dispatches)")
PY
  rm -rf $O/p$i
done
