#!/usr/bin/env python3
"""Writes the bench's two inflate inputs (chromosome 21 at 1x: random qualities at level 1, run-structured binned qualities at
level 6) into a directory, for tools/inflate_prof:  python3 tools/inflate_prof_bamlike.py DIR  ->  DIR/random.bam, DIR/bamlike.bam"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import e2e_bench, formats, synth  # noqa: E402

d = sys.argv[1]
os.makedirs(d, exist_ok=True)
lens = dict(synth.GRCH37_AUTOSOMES)
c = synth.make_chrom("21", lens["21"], cov=1.0)
formats.write_bam_fast(os.path.join(d, "random.bam"), "S", [(c.name, c.length, c.pos, c.mapq)], realistic=True, level=1)
p2, _t = e2e_bench.write_bam(d, "q", [(c.name, c.length, c.pos, c.mapq)], level=6)
os.replace(p2, os.path.join(d, "bamlike.bam"))
print("written")
