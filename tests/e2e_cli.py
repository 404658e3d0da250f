#!/usr/bin/env python3
"""End-to-end run of the `conga` executable on a synthetic BAM (BASELINE configs[0] shape by default: chr21 only,
~2k deletions, 0.5x) and a larger one; prints wall times.  The outputs are compared with the oracle's files.
Not a benchmark line (bench.py is): host BAM decoding and text parsing dominate here."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import formats, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chroms", default="21")
    ap.add_argument("--cov", type=float, default=0.5)
    ap.add_argument("--dels", type=int, default=2000)
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--input", choices=("bam", "tuples"), default="bam",
                    help="tuples: the read-tuple container (decoded pos / mapq per chromosome) instead of a BAM -- what "
                         "is left of the wall time is file read, PCIe staging, compute and the output writer")
    ap.add_argument("--repeat", type=int, default=1, help="runs per variant; the fastest is reported")
    ap.add_argument("--decode", choices=("default", "both"), default="default",
                    help="both: run the same input a second time with CONGA_GPU_BAM=0 (host decoders only)")
    ap.add_argument("--bai", type=int, default=1, help="write the .bai too (0: the reader then goes through the file front to back)")
    ap.add_argument("--gpus", default="1", help="comma list of `conga --gpus` values to run on the same input (contexts "
                    "share the device when there are fewer devices)")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="conga_e2e_")
    lens = dict(synth.GRCH37_AUTOSOMES)
    names = [n for n, _ in synth.GRCH37_AUTOSOMES] if a.chroms == "all" else a.chroms.split(",")
    total = sum(lens[n] for n in names)
    cs = [synth.make_chrom(n, lens[n], cov=a.cov, n_dels=int(round(a.dels * lens[n] / total))) for n in names]
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    t0 = time.time()
    reads_file = "r.bam" if a.input == "bam" else "r.ctp"
    if a.input == "bam":
        formats.write_bam_fast(os.path.join(d, reads_file), "SYNTH", [(c.name, c.length, c.pos, c.mapq) for c in cs], realistic=True,
                               index=bool(a.bai))
    else:
        formats.write_tuples(os.path.join(d, reads_file), "SYNTH", [(c.name, c.length, c.pos, c.mapq) for c in cs])
    t_bam = time.time() - t0
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    runs = {}
    first_bytes = None
    variants = [(int(x), {}) for x in a.gpus.split(",")]
    if a.decode == "both":       # the same input once more with the host decoders only
        variants.append((variants[0][0], {"CONGA_GPU_BAM": "0"}))
    for g, extra_env in variants:
        t_cli = None
        for _ in range(max(1, a.repeat)):   # best of `--repeat` runs: a 3 GB input makes single runs noisy
            t0 = time.time()
            r1 = subprocess.run([os.path.join(ROOT, "conga_amd", "host", "conga"), "-i", reads_file, "--out", "got", "--ref", "none.fa",
                                 "--sonic", "a.cga", "--dels", "dels.bed", "--gpus", str(g)], cwd=d, capture_output=True, text=True,
                                env=dict(os.environ, CONGA_TIMING="1", **extra_env))
            t1 = time.time() - t0
            assert r1.returncode == 0, r1.stderr[-2000:]
            if t_cli is None or t1 < t_cli:
                t_cli, r = t1, r1
        runs["%d%s" % (g, " host-decode" if extra_env else "")] = round(t_cli, 3)
        got = [open(os.path.join(d, "got_%s.bed" % k), "rb").read() for k in ("svs", "dels")]
        if first_bytes is None:
            first_bytes = got
        assert got == first_bytes, "--gpus %d changed the output files" % g
        for line in r.stderr.splitlines():
            if "[timing" in line:
                print("--gpus %d%s %s" % (g, " CONGA_GPU_BAM=0" if extra_env else "", line), file=sys.stderr)
    t_cli = runs[a.gpus.split(",")[0]]
    ok = None
    if a.check:
        paths = [os.path.join(d, "want_%s.bed" % k) for k in ("svs", "dels")]
        first = True
        for c in cs:
            ds = O.sort_svs(O.load_known_SVs(os.path.join(d, "dels.bed"), c.name, 1000))
            if len(ds) == 0:
                continue
            rd, _ = O.count_reads(c.length, c.pos, c.mapq, -1)
            E, _, _ = O.calc_mean_per_chr(rd, c.gc)
            O.find_depths(rd, None, c.gc, E, "D", ds)
            O.output_svs(c.name, ds, None, paths[0], paths[1], None, have_mappability=False, write_headers=first)
            first = False
        ok = all(open(os.path.join(d, "got_%s.bed" % k), "rb").read() == open(w, "rb").read()
                 for k, w in zip(("svs", "dels"), paths))
    n_iv = sum(1 for _ in open(os.path.join(d, "got_dels.bed"))) - 1
    print(json.dumps(dict(chroms=names, reads=int(sum(len(c.pos) for c in cs)), input=a.input, input_mb=round(os.path.getsize(os.path.join(d, reads_file)) / 1e6, 1),
                          intervals=n_iv, cli_wall_s=round(t_cli, 3), cli_wall_s_by_gpus=runs, intervals_per_s=round(n_iv / t_cli, 1),
                          input_write_s=round(t_bam, 1), outputs_match_oracle=ok, host_cpus=os.cpu_count())))


if __name__ == "__main__":
    main()
