#!/usr/bin/env python3
"""`conga --cohort` over read-tuple containers (.ctp) of whole-genome 1x samples: what a further sample costs when nothing has to be
decoded -- the hand-over of count_reads_bam's tuples through the executable.  tools/ctp_cohort.py [--samples K]"""
import argparse
import os
import re
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import e2e_bench, formats, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=12)
ap.add_argument("--chroms", default="")
a = ap.parse_args()
chroms = synth.GRCH37_AUTOSOMES
if a.chroms:
    chroms = tuple(c for c in chroms if c[0] in set(a.chroms.split(",")))
plan = synth.genome_plan(chroms, synth.N_DELS_GENOME, 0)
cs = [synth.make_chrom(n, L, cov=1.0, n_dels=nd) for n, L, nd, _nu in plan]
d = tempfile.mkdtemp(prefix="conga_ctp_cohort_")
try:
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    formats.write_tuples(os.path.join(d, "s.ctp"), "S", [(c.name, c.length, c.pos, c.mapq) for c in cs])
    with open(os.path.join(d, "list.txt"), "w") as f:
        for k in range(a.samples):
            f.write("s.ctp\tc%d\n" % k)
    for rep in range(2):
        dt, err = e2e_bench.run_conga(["--cohort", "list.txt", "--out", "x", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed"], d, dict(CONGA_TIMING="1"))
        done = [float(x) for x in re.findall(r"cohort: sample \d+ of \d+ is done ([0-9.]+) ms", err)]
        print("run %d: wall %.3f s; sample ends (ms): %s; per further sample %.1f ms" % (rep, dt, " ".join("%.0f" % x for x in done), (done[-1] - done[0]) / max(len(done) - 1, 1)))
        lines = [ln for ln in err.splitlines() if "[timing] packed hand-over" in ln]
        for ln in lines[2:3]:
            print("   " + ln[:300])
        lines = [ln for ln in err.splitlines() if "[timing] open + BED" in ln]
        for ln in lines[:1] + lines[3:4]:   # the first sample's and a later one's
            print("   " + ln[:300])
finally:
    shutil.rmtree(d, ignore_errors=True)
