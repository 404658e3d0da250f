"""Regenerates tests/golden/small_chr.npz and the three expected output files.

These are REGRESSION fixtures produced by the oracle restatement (oracle/conga_oracle.c), not outputs of
the reference: the reference cannot be built in this image (htslib + sonic are absent) and ships no
fixtures of its own.  They pin the oracle against accidental change and give the GPU tests a case that
does not depend on the generator.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from conga_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    c = synth.make_chrom("21", 200_000, cov=1.5, n_dels=40, n_dups=10, mappability=True, gaps=False, seed=7)
    # hand-made rows for the edge cases of SURVEY.md section 8c
    extra_s = np.array([0, 100_000, 100_000, 150_000, 150_000, 199_000, 120_050], np.int32)
    extra_e = np.array([1000, 101_000, 101_000, 151_000, 150_999, 200_000, 121_050], np.int32)
    ds, de = synth.kept_sorted(np.concatenate([c.del_start, extra_s]), np.concatenate([c.del_end, extra_e]))
    us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
    pos, mapq = c.pos.copy(), c.mapq.copy()
    hole = (pos >= 120_000) & (pos < 122_000)          # an interval with zero reads: score -0.00
    pos, mapq = pos[~hole], mapq[~hole]

    rd, counted = O.count_reads(c.length, pos, mapq, -1)
    E, S, W = O.calc_mean_per_chr(rd, c.gc)
    m = O.paint_mappability(c.length, c.map_start, c.map_end, c.map_val)
    dels = O.find_depths(rd, m, c.gc, E, "D", O.make_svs(ds, de))
    dups = O.find_depths(rd, m, c.gc, E, "E", O.make_svs(us, ue))
    np.savez_compressed(os.path.join(HERE, "small_chr.npz"), length=c.length, step=c.step, gc=c.gc, pos=pos,
                        mapq=mapq, map_start=c.map_start, map_end=c.map_end, map_val=c.map_val,
                        counted=counted, rd_nonzero_idx=np.flatnonzero(rd).astype(np.int32),
                        rd_nonzero_val=rd[rd != 0], E=E, S=S, W=W, dels=dels, dups=dups)
    for tag, have_map in (("map", True), ("nomap", False)):
        paths = [os.path.join(HERE, "small_chr_%s_%s.bed" % (tag, k)) for k in ("svs", "dels", "dups")]
        O.output_svs("21", dels, dups, *paths, have_mappability=have_map, write_headers=True)
    print("wrote", os.path.join(HERE, "small_chr.npz"), len(pos), "reads", len(ds), "dels", len(us), "dups")


if __name__ == "__main__":
    main()
