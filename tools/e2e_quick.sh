#!/bin/bash
# whole-genome 1x BAM once (cached under /tmp for the life of the box), then timed CLI runs with CONGA_TIMING
set -e
set +o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=/tmp/e2e_wg
if [ ! -f $D/r.bam.bai ]; then
  mkdir -p $D
  python - <<PY
import os, sys
sys.path.insert(0, ".")
from conga_amd import formats, synth
d = "$D"
lens = dict(synth.GRCH37_AUTOSOMES)
names = [n for n, _ in synth.GRCH37_AUTOSOMES]
total = sum(lens.values())
cs = [synth.make_chrom(n, lens[n], cov=1.0, n_dels=int(round(42000 * lens[n] / total))) for n in names]
formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
formats.write_bam_fast(os.path.join(d, "r.bam"), "SYNTH", [(c.name, c.length, c.pos, c.mapq) for c in cs], realistic=True, index=True)
synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
PY
fi
cd $D
for i in 1 2 3; do
  t0=$(date +%s%N)
  env CONGA_TIMING=1 CONGA_T0_NS=$t0 "$@" ${GRAFT_REPO_ROOT:-/root/repo}/conga_amd/host/conga -i r.bam --ref r.fa --sonic a.cga --dels dels.bed --out o$i > run$i.log 2>&1 || { echo "conga failed"; tail -5 run$i.log; }
  t1=$(date +%s%N)
  grep -a "timing" run$i.log || true
  echo "wall $(( (t1 - t0) / 1000000 )) ms"
done
md5sum o1_dels.bed o2_dels.bed o3_dels.bed
# the same genome as a cohort of four (the same BAM four times): one process, the engine kept from sample to sample
printf "r.bam\tc0\nr.bam\tc1\nr.bam\tc2\nr.bam\tc3\n" > list.txt
t0=$(date +%s%N)
env CONGA_TIMING=1 CONGA_T0_NS=$t0 "$@" ${GRAFT_REPO_ROOT:-/root/repo}/conga_amd/host/conga --cohort list.txt --ref r.fa --sonic a.cga --dels dels.bed --out co > cohort.log 2>&1 || { echo "cohort failed"; tail -5 cohort.log; }
t1=$(date +%s%N)
grep -a "timing\] open\|caller" cohort.log || true
echo "cohort of 4: wall $(( (t1 - t0) / 1000000 )) ms"
md5sum c0_dels.bed c3_dels.bed
