#!/usr/bin/env python3
"""Where a further sample of `conga --cohort` spends its time: whole-genome 1x BAMs written on the spot (tools/bamwrite), a
list of four through the executable with CONGA_TIMING=1, the [timing] lines of the run printed."""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import e2e_bench, formats, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chroms", default="")
ap.add_argument("--samples", type=int, default=4)
ap.add_argument("--decode", default="1")
ap.add_argument("--variants", default="", help="semicolon-separated sets of K=V,K=V environment settings: the cohort is run once per set "
                                               "and only the BAM stage's [timing] lines are printed (measurement switches of the engine)")
a = ap.parse_args()
chroms = synth.GRCH37_AUTOSOMES
if a.chroms:
    chroms = tuple(c for c in chroms if c[0] in set(a.chroms.split(",")))
plan = synth.genome_plan(chroms, synth.N_DELS_GENOME, 0)
cs = [synth.make_chrom(n, L, cov=1.0, n_dels=nd) for n, L, nd, _nu in plan]
d = tempfile.mkdtemp(prefix="conga_e2e_timing_")
try:
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    bam, t = e2e_bench.write_bam(d, "s0", [(c.name, c.length, c.pos, c.mapq) for c in cs])
    print("BAM %.2f GB written in %.1f s" % (os.path.getsize(bam) / 1e9, t))
    with open(os.path.join(d, "list.txt"), "w") as f:
        for k in range(a.samples):
            f.write("%s\ts%d\n" % (bam, k))
    for var in [v for v in a.variants.split(";") if v]:
        envx = dict(kv.split("=", 1) for kv in var.split(","))
        for rep in range(2):
            try:
                import time
                dt, err = e2e_bench.run_conga(["--cohort", "list.txt", "--out", "x", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed"], d,
                                              dict(envx, CONGA_GPU_BAM=a.decode, CONGA_TIMING="1", CONGA_T0_NS=str(time.time_ns())))
            except RuntimeError as e:
                print("[%s] failed: %s" % (var, str(e)[-200:]))
                continue
            print("[%s] run %d: wall %.3f s" % (var, rep, dt))
            for line in err.splitlines():   # (with CONGA_T0_NS in the variant: where the process's own clock stands against the caller's)
                if "after the caller's clock" in line or "conga_create:" in line:
                    print("   " + line[:300])
            for line in err.splitlines():
                if "overlapped upload" in line or "conga_reads_bgzf:" in line:
                    print("   " + line[:300])
            import re
            done = [float(x) for x in re.findall(r"cohort: sample \d+ of \d+ is done ([0-9.]+) ms", err)]
            if len(done) > 2:
                print("   sample ends (ms): " + " ".join("%.0f" % x for x in done) + "   per further sample: %.1f, median of the later ones: %.1f"
                      % ((done[-1] - done[0]) / (len(done) - 1), sorted(b - a for a, b in zip(done[2:-1], done[3:]))[max(0, (len(done) - 4) // 2)] if len(done) > 4 else float("nan")))
    for rep in range(0 if a.variants else 2):
        dt, err = e2e_bench.run_conga(["--cohort", "list.txt", "--out", "x", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed"], d,
                                      dict(CONGA_GPU_BAM=a.decode, CONGA_TIMING="1"))
        print("run %d: wall %.3f s for %d samples" % (rep, dt, a.samples))
    for line in ([] if a.variants else err.splitlines()):
        if "[timing]" in line or "[CONGA] sample" in line:
            print(line[:260])
finally:
    shutil.rmtree(d, ignore_errors=True)
