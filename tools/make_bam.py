#!/usr/bin/env python3
"""A synthetic BAM (+ .bai) with random bases and qualities for the decode experiments: tools/make_bam.py OUT.bam [CHROM COV]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conga_amd import formats, synth  # noqa: E402

out = sys.argv[1]
name = sys.argv[2] if len(sys.argv) > 2 else "21"
cov = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
c = synth.make_chrom(name, dict(synth.GRCH37_AUTOSOMES)[name], cov=cov)
formats.write_bam_fast(out, "SYNTH", [(c.name, c.length, c.pos, c.mapq)], realistic=True, index=True, level=int(os.environ.get("LEVEL", "6")))
print(out, len(c.pos), "reads", os.path.getsize(out) / 1e6, "MB")
