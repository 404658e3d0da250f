#!/usr/bin/env python3
"""Inputs of an end-to-end run, written once so that several builds can be timed on them: gen_e2e_inputs.py DIR CHROMS COV"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conga_amd import formats, synth  # noqa: E402

d, names, cov = sys.argv[1], sys.argv[2].split(","), float(sys.argv[3])
os.makedirs(d, exist_ok=True)
lens = dict(synth.GRCH37_AUTOSOMES)
total = sum(lens[n] for n in names)
cs = [synth.make_chrom(n, lens[n], cov=cov, n_dels=int(round(3000 * len(names) * lens[n] / total))) for n in names]
formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
formats.write_bam_fast(os.path.join(d, "r.bam"), "SYNTH", [(c.name, c.length, c.pos, c.mapq) for c in cs], realistic=True, index=True)
synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
print(d, sum(len(c.pos) for c in cs), "reads", os.path.getsize(os.path.join(d, "r.bam")) / 1e6, "MB")
