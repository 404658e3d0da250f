#!/usr/bin/env python3
"""BGZF inflate on the device (conga_inflate_blocks), wave-per-block against lane-per-block, on a synthetic BAM with
random bases and qualities (compresses about as poorly as real data).  Prints one JSON object."""
import argparse
import json
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import capi, formats, synth  # noqa: E402


def block_table(raw):
    blocks, at, n = [], 0, len(raw)
    while at + 18 <= n:
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", raw, at + bsize - 8)
        if isize:
            blocks.append((at + 18, bsize - 26, isize, crc))
        at += bsize
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chroms", default="1,2,3")
    ap.add_argument("--cov", type=float, default=1.0)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kernels", default="wave,lane")
    ap.add_argument("--sizes", default="64,1000,8000,0", help="block counts to time (0 = all)")
    a = ap.parse_args()
    lens = dict(synth.GRCH37_AUTOSOMES)
    cs = [synth.make_chrom(n, lens[n], cov=a.cov) for n in a.chroms.split(",")]
    d = tempfile.mkdtemp(prefix="conga_infl_")
    path = os.path.join(d, "r.bam")
    t0 = time.time()
    formats.write_bam_fast(path, "S", [(c.name, c.length, c.pos, c.mapq) for c in cs], realistic=True, level=a.level)
    raw = np.fromfile(path, np.uint8)
    blocks = block_table(raw.tobytes())
    out = dict(file_mb=round(len(raw) / 1e6, 1), blocks=len(blocks), inflated_mb=round(sum(b[2] for b in blocks) / 1e6, 1),
               level=a.level, write_s=round(time.time() - t0, 1), runs=[])
    with capi.Context(device=0) as ctx:
        for n in [int(x) for x in a.sizes.split(",")]:
            sel = blocks if n == 0 else blocks[:n]
            hi = sel[-1][0] + sel[-1][1] + 8
            inflated = sum(b[2] for b in sel)
            for kernel in a.kernels.split(","):
                os.environ["CONGA_DEBUG"] = "1"
                os.environ["CONGA_BGZF_KERNEL"] = kernel
                best = 1e30
                for _ in range(a.reps):
                    _o, status, ms = ctx.inflate_blocks(raw[:hi], sel, want_out=False)
                    assert not status.any(), (kernel, n, int(status.argmax()), int(status.max()))
                    best = min(best, ms)
                out["runs"].append(dict(kernel=kernel, blocks=len(sel), inflated_mb=round(inflated / 1e6, 1), ms=round(best, 3),
                                        inflated_gbs=round(inflated / best / 1e6, 2)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
