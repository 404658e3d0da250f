#!/usr/bin/env python3
"""The timeline of a `conga --cohort` steady state: whole-genome 1x BAMs (tools/bamwrite), CONGA_DEBUG=1 CONGA_BGZF_TRACE=1, the trace
lines ([bz <ms>] ...: conga_amd/csrc/bz_sched.h) of a few samples in the middle of the run, times relative to the first of them.
tools/cohort_trace.py [--samples K] [--from-sample A] [--to-sample B] [--env K=V,K=V]"""
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
ap.add_argument("--from-sample", type=int, default=7)
ap.add_argument("--to-sample", type=int, default=9)
ap.add_argument("--chroms", default="")
ap.add_argument("--env", default="")
ap.add_argument("--keep", default="", help="make the inputs in this directory, leave them there and print the command line (for a profiler)")
a = ap.parse_args()
chroms = synth.GRCH37_AUTOSOMES
if a.chroms:
    chroms = tuple(c for c in chroms if c[0] in set(a.chroms.split(",")))
plan = synth.genome_plan(chroms, synth.N_DELS_GENOME, 0)
cs = [synth.make_chrom(n, L, cov=1.0, n_dels=nd) for n, L, nd, _nu in plan]
d = a.keep or tempfile.mkdtemp(prefix="conga_cohort_trace_")
os.makedirs(d, exist_ok=True)
try:
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    bam, t = e2e_bench.write_bam(d, "s0", [(c.name, c.length, c.pos, c.mapq) for c in cs])
    with open(os.path.join(d, "list.txt"), "w") as f:
        for k in range(a.samples):
            f.write("%s\ts%d\n" % (bam, k))
    if a.keep:
        print("cd %s && %s --cohort list.txt --out x --ref r.fa --sonic a.cga --dels dels.bed" % (d, os.path.join(ROOT, "conga_amd", "host", "conga")))
        sys.exit(0)
    env = dict(CONGA_GPU_BAM="1", CONGA_TIMING="1", CONGA_DEBUG="1", CONGA_BGZF_TRACE="1")
    if a.env:
        env.update(kv.split("=", 1) for kv in a.env.split(","))
    for rep in range(2):
        dt, err = e2e_bench.run_conga(["--cohort", "list.txt", "--out", "x", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed"], d, env)
    done = [float(x) for x in re.findall(r"cohort: sample \d+ of \d+ is done ([0-9.]+) ms", err)]
    print("wall %.3f s; sample ends (ms): %s" % (dt, " ".join("%.0f" % x for x in done)))
    for ln in [ln for ln in err.splitlines() if "overlapped upload: named ahead" in ln][-3:]:
        print("   " + ln[:330])
    lines = [ln for ln in err.splitlines() if ln.startswith("[bz ")]
    t0 = None
    show = False
    for ln in lines:
        m = re.match(r"\[bz\s+([0-9.]+)\] (.*)", ln)
        t, what = float(m.group(1)), m.group(2)
        b = re.match(r"cli: begins \(sample (\d+)\)", what)
        if b:
            show = a.from_sample <= int(b.group(1)) <= a.to_sample
            if show and t0 is None:
                t0 = t
        if show:
            print("%8.2f  %s" % (t - t0, what))
finally:
    if not a.keep:
        shutil.rmtree(d, ignore_errors=True)
