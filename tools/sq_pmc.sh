#!/bin/bash
# SQ counters of one kernel: tools/sq_pmc.sh TAG KERNEL_SUBSTRING program args...   (one rocprofv3 --pmc pass per counter set;
# the program itself follows, never a shell or env wrapper)  ->  gpurun_out/TAG_sq_pmc/summary.txt
set -e
TAG=$1; KERNEL=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_sq_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P3="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH"
i=0
: > $O/summary.txt
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -- "$@" > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$KERNEL" <<'PY' | tee -a $O/summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:32s} {v / max(n, 1):16.0f}  (avg of {n} dispatches)")
PY
  rm -rf $O/p$i
done
