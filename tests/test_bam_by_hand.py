"""A BAM + .bai that the repository's writers did not write (tests/bam_by_hand.py: assembled from the SAM/BAM specification byte by
byte -- multi-operation CIGARs, optional fields of every type, placed-unmapped reads, records across three and more BGZF blocks, an
empty block in mid-file, stored blocks, the EOF marker, the metadata pseudo-bin 37450) through every route the records can take:
the host decoders front to back and index-guided, the decode on the GPU, and --rp with the records read in place.  The seam is
count_reads_bam and its htslib iterator (bam_data.c:192-221, 199-201, 253-259, 293)."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bam_by_hand  # noqa: E402
from conga_amd import formats, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONGA = os.path.join(ROOT, "conga_amd", "host", "conga")


def lay_out(d):
    """the hand-made BAM, an annotation that knows "1", "3" (not in the BAM), "2" and "empty" -- not "GL000207.1" --, and a call set"""
    h = bam_by_hand.make(d)
    lens = dict(h["refs"])
    rng = np.random.default_rng(5)
    chroms = []
    for name in ("1", "3", "2", "empty"):
        L = lens.get(name, 80_000)
        gc = np.clip(rng.normal(41, 8, (L + 99) // 100).round(), 20, 75).astype(np.uint8)
        chroms.append(types.SimpleNamespace(name=name, length=L, gc=gc))
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [(1000, 3000)] if c.name == "1" else [], []) for c in chroms])
    dels, dups = [], []
    for c in chroms:
        for _ in range(40 if c.name != "empty" else 3):
            s = int(rng.integers(0, c.length - 25_000))
            dels.append((c.name, s, s + int(rng.integers(1000, 20_000))))
        for _ in range(12 if c.name != "empty" else 1):
            s = int(rng.integers(0, c.length - 45_000))
            dups.append((c.name, s, s + int(rng.integers(2000, 40_000))))
    dels.append(("1", 0, 1500))            # an interval on the reference's very first bases (three reads start on base 0)
    synth.write_bed(os.path.join(d, "dels.bed"), dels)
    synth.write_bed(os.path.join(d, "dups.bed"), dups)
    for c in chroms:
        e = h["expect"].get(c.name)
        c.pos = e[:, 0].astype(np.int32) if e is not None and len(e) else np.zeros(0, np.int32)
        c.mapq = e[:, 1].astype(np.uint8) if e is not None and len(e) else np.zeros(0, np.uint8)
    return h, chroms


def test_the_host_readers_find_what_was_written(tmp_path):
    """No GPU: --dump-reads prints, per chromosome, count and checksums of the records count_reads_bam would be handed -- front to
    back, index-guided in one piece and in eight segments, with every block through zlib; and the assembler's own claims hold (a
    record really lies across three blocks, the file really has an empty block inside and ends with the EOF marker)."""
    d = str(tmp_path)
    h, chroms = lay_out(d)
    raw = open(os.path.join(d, "hand.bam"), "rb").read()
    assert raw.endswith(bam_by_hand.EOF_MARKER)
    sizes, isizes, at = [], [], 0
    while at < len(raw):
        bsize = int.from_bytes(raw[at + 16:at + 18], "little") + 1
        sizes.append(bsize)
        isizes.append(int.from_bytes(raw[at + bsize - 4:at + bsize], "little"))
        at += bsize
    assert 0 in isizes[:-1] and isizes[-1] == 0                                 # an empty block inside, and the marker
    assert sum(1 for x in isizes if 0 < x <= 64) >= 3                           # the stretch of 64-byte blocks
    starts = np.cumsum([0] + isizes)
    a = h["three_block_record_at"]
    assert np.searchsorted(starts, a + 900, side="right") - np.searchsorted(starts, a, side="right") >= 3   # that record: many blocks
    want = {c.name: (len(c.pos), int(c.pos.astype(np.int64).sum()), int(c.mapq.astype(np.int64).sum())) for c in chroms}
    for env in ({}, {"CONGA_BAM_SEGMENTS": "0"}, {"CONGA_BAM_SEGMENTS": "8"}, {"CONGA_ZLIB_INFLATE": "1"}, {"CONGA_BAM_THREADS": "0"}):
        for bai in (True, False):
            if not bai:
                os.rename(os.path.join(d, "hand.bam.bai"), os.path.join(d, "hidden.bai"))
            r = subprocess.run([CONGA, "-i", "hand.bam", "--out", "o", "--ref", "ref.fa", "--sonic", "a.cga", "--dump-reads"], cwd=d,
                               capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
            if not bai:
                os.rename(os.path.join(d, "hidden.bai"), os.path.join(d, "hand.bam.bai"))
            assert r.returncode == 0, r.stderr[-2000:]
            lines = dict(ln.split("\t", 1) for ln in r.stdout.splitlines() if "\t" in ln)
            assert lines["sample"] == "HAND" and (lines["index"] == "none") == (not bai)
            assert lines["3"] == "missing"
            for name in ("1", "2", "empty"):
                assert tuple(int(x) for x in lines[name].split("\t")) == want[name], (env, bai, name, lines[name], want[name])
    assert want["1"][0] == 5200 and want["2"][0] == 2100 and want["empty"][0] == 0


@pytest.mark.gpu
def test_every_route_gives_the_oracles_files(tmp_path, oracle):
    """Host decoders, the decode on the GPU (one call, and chromosome by chromosome), --rp with the records read where the inflate
    left them: the depth columns are the oracle's on the tuples the assembler wrote, and the routes' files are alike byte for byte."""
    from test_host_cli import _oracle_files
    d = str(tmp_path)
    h, chroms = lay_out(d)
    common = ["-i", "hand.bam", "--ref", "ref.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed"]
    in_bam = [c for c in chroms if c.name != "3"]
    want = _oracle_files(oracle, d, chroms, in_bam, dels=True, dups=True, with_map=False, mq=-1)
    size = os.path.getsize(os.path.join(d, "hand.bam"))
    outs = {}
    for tag, env in (("host", dict(CONGA_GPU_BAM="0")), ("host8", dict(CONGA_GPU_BAM="0", CONGA_BAM_SEGMENTS="8")),
                     ("gpu", dict(CONGA_GPU_BAM="1", CONGA_TIMING="1")),
                     ("gpu_ovl", dict(CONGA_GPU_BAM="1", CONGA_TIMING="1", CONGA_BGZF_OVERLAP="1", CONGA_BGZF_PIECE_KB="8")),
                     ("gpu_each", dict(CONGA_GPU_BAM="1", CONGA_TIMING="1", CONGA_GPU_BAM_MAX_MB="%.4f" % (size * 0.7 / 1048576)))):
        r = subprocess.run([CONGA] + common + ["--out", tag], cwd=d, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, (tag, r.stderr[-2500:])
        if tag.startswith("gpu"):
            assert "decoding on the host" not in r.stderr and "conga_reads_bgzf:" in r.stderr, (tag, r.stderr[-2500:])
        outs[tag] = [open(os.path.join(d, "%s_%s.bed" % (tag, k)), "rb").read() for k in ("svs", "dels", "dups")]
        assert outs[tag] == [open(w, "rb").read() for w in want], tag
        assert "Cannot find chromosome name 3 in BAM/CRAM HAND" in r.stderr
    # --rp: the split-read path on records with clips, insertions, missing qualities, ambiguity codes, unmapped-but-placed reads
    rp = {}
    for tag, env in (("rp_host", dict(CONGA_GPU_BAM="0")), ("rp_gpu", dict(CONGA_GPU_BAM="1", CONGA_TIMING="1")),
                     ("rp_gpu_each", dict(CONGA_GPU_BAM="1", CONGA_TIMING="1", CONGA_GPU_BAM_MAX_MB="%.4f" % (size * 0.7 / 1048576)))):
        r = subprocess.run([CONGA] + common + ["--rp", "2", "--min-read-length", "50", "--out", tag], cwd=d, capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, (tag, r.stderr[-2500:])
        if "gpu" in tag:
            assert "decoding on the host" not in r.stderr, (tag, r.stderr[-2500:])
        import re
        rp[tag] = ([open(os.path.join(d, "%s_%s.bed" % (tag, k)), "rb").read() for k in ("svs", "dels", "dups")],
                   re.findall(r"\((\d+) reads, (\d+) split-reads\)", r.stderr), re.findall(r"CONGA paired (\d+)", r.stderr))
    assert rp["rp_host"] == rp["rp_gpu"] == rp["rp_gpu_each"]
    assert [int(a) for a, _b in rp["rp_host"][1]] == [5200, 2100, 0] and int(rp["rp_host"][1][0][1]) > 5000   # split-read elements were made
    # the depth columns do not depend on --rp (OBSERVED / EXPECTED of the dels file: columns 8 and 9)
    cols = lambda blob: [ln.split(b"\t")[7:9] for ln in blob.splitlines()[1:]]
    assert cols(rp["rp_host"][0][1]) == cols(outs["host"][1])
