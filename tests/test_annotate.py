"""`conga-annotate` (conga_amd/host/annotate.cpp): FASTA (+ satellite BED) -> the annotation container that stands in
for the reference's .sonic file (svdepth.c:47).  What CONGA reads from the annotation is
chromosome names / lengths (bam_data.c:269-293), GC% per 100-base window (read_distribution.c:70, likelihood.c:117)
and satellite membership (bam_data.c:96-97,207)."""
import os
import subprocess

import numpy as np
import pytest

from conga_amd import formats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "conga_amd", "host", "conga-annotate")
CONGA = os.path.join(ROOT, "conga_amd", "host", "conga")


def annotate(args, cwd):
    return subprocess.run([TOOL] + args, cwd=cwd, capture_output=True, text=True, timeout=120)


def gc_windows(seq, step):
    """Round-half-up of 100 * (#G + #C) / bases, per window; the last window is shorter."""
    b = np.frombuffer(seq, np.uint8)
    is_gc = np.isin(b, np.frombuffer(b"GCgc", np.uint8)).astype(np.int64)
    n_win = (len(b) + step - 1) // step
    edges = np.minimum(np.arange(n_win + 1) * step, len(b))
    cs = np.concatenate([[0], np.cumsum(is_gc)])
    gc = cs[edges[1:]] - cs[edges[:-1]]
    ln = edges[1:] - edges[:-1]
    return ((200 * gc + ln) // (2 * ln)).astype(np.uint8)


def random_seq(rng, n, p_gc):
    at = rng.choice(np.frombuffer(b"ATat", np.uint8), n)
    gc = rng.choice(np.frombuffer(b"GCgc", np.uint8), n)
    seq = np.where(rng.random(n) < p_gc, gc, at)
    return seq


def test_gc_windows_names_lengths_and_satellites(tmp_path):
    d = str(tmp_path)
    rng = np.random.default_rng(7)
    s1 = random_seq(rng, 12_345, 0.41)
    s1[2000:4100] = ord("N")                       # a gap: GC 0 for whole windows, partial at its edges
    s1[4100:4200] = ord("G")                       # GC 100
    s2 = random_seq(rng, 700, 0.6)
    s3 = random_seq(rng, 100, 0.5)                 # exactly one window
    seqs = [("1", s1.tobytes()), ("chr2 some description", s2.tobytes()), ("MT", s3.tobytes())]
    formats.write_fasta(os.path.join(d, "ref.fa"), seqs, width=61, index=False)
    with open(os.path.join(d, "sat.bed"), "w") as f:
        f.write("#chrom\tstart\tend\tname\n")
        f.write("track name=rmsk\n")
        f.write("1\t500\t900\tSatellite/centr\n")
        f.write("1\t850\t1200\t(GAATG)n\tsatellite\n")  # overlaps the first: merged
        f.write("1\t1200\t1300\tSATELLITE\n")            # abuts: merged
        f.write("1\t5000\t5100\tLINE/L1\n")              # filtered by --match
        f.write("1\t12000\t99999\tSatellite\n")          # clipped to the chromosome
        f.write("chr2\t10\t20\tSatellite\n")
        f.write("7\t10\t20\tSatellite\n")                # not in the FASTA
        f.write("1\t300\t300\tSatellite\n")              # empty
    r = annotate(["--ref", "ref.fa", "--out", "a.cga", "--satellites", "sat.bed", "--match", "satellite"], d)
    assert r.returncode == 0, r.stderr
    assert "kept 5, skipped 3" in r.stdout
    step, chroms = formats.read_annotation(os.path.join(d, "a.cga"))
    assert step == 100
    assert [(n, L) for n, L, *_ in chroms] == [("1", 12_345), ("chr2", 700), ("MT", 100)]
    for (name, L, gc, ss, se), (_, seq) in zip(chroms, seqs):
        np.testing.assert_array_equal(gc, gc_windows(seq, 100), err_msg=name)
    gc1 = chroms[0][2]
    assert gc1[21] == 0 and gc1[40] == 0 and gc1[41] == 100 and len(gc1) == 124
    assert list(zip(chroms[0][3], chroms[0][4])) == [(500, 1300), (12000, 12345)]
    assert list(zip(chroms[1][3], chroms[1][4])) == [(10, 20)]
    assert len(chroms[2][3]) == 0
    # without --match every row counts; CRLF line ends and a missing trailing newline do not change the numbers
    raw = open(os.path.join(d, "ref.fa"), "rb").read().replace(b"\n", b"\r\n").rstrip(b"\r\n")
    open(os.path.join(d, "crlf.fa"), "wb").write(raw)
    r = annotate(["--ref", "crlf.fa", "--out", "b.cga", "--satellites", "sat.bed", "--gc-window", "64"], d)
    assert r.returncode == 0, r.stderr
    step, chroms_b = formats.read_annotation(os.path.join(d, "b.cga"))
    assert step == 64
    for (name, L, gc, ss, se), (_, seq) in zip(chroms_b, seqs):
        assert L == len(seq)
        np.testing.assert_array_equal(gc, gc_windows(seq, 64), err_msg=name)
    assert (5000, 5100) in list(zip(chroms_b[0][3], chroms_b[0][4]))


def test_errors(tmp_path):
    d = str(tmp_path)
    assert annotate([], d).returncode == 3
    r = annotate(["--ref", "missing.fa", "--out", "a.cga"], d)
    assert r.returncode == 1 and "Unable to open file missing.fa in read mode" in r.stderr
    open(os.path.join(d, "empty.fa"), "w").write(">1\n>2\n")
    r = annotate(["--ref", "empty.fa", "--out", "a.cga"], d)
    assert r.returncode == 1 and "holds no sequence" in r.stderr
    open(os.path.join(d, "ok.fa"), "w").write(">1\nACGT\n")
    assert annotate(["--ref", "ok.fa", "--out", "a.cga", "--gc-window", "0"], d).returncode == 3
    assert annotate(["--ref", "ok.fa", "--out", "a.cga", "--satellites", "nope.bed"], d).returncode == 1
    assert annotate(["--ref", "ok.fa", "--out", "a.cga"], d).returncode == 0
    assert formats.read_annotation(os.path.join(d, "a.cga"))[1][0][2].tolist() == [50]


def test_conga_loads_the_container(tmp_path):
    """The host loader of `conga --sonic` accepts what the tool writes (--dump-reads needs no GPU)."""
    d = str(tmp_path)
    rng = np.random.default_rng(3)
    seq = random_seq(rng, 5000, 0.45).tobytes()
    formats.write_fasta(os.path.join(d, "ref.fa"), [("1", seq)])
    assert annotate(["--ref", "ref.fa", "--out", "a.cga"], d).returncode == 0
    pos = np.array([5, 10, 4999, 5000], np.int32)
    formats.write_bam(os.path.join(d, "r.bam"), "s", [("1", 6000, pos, np.array([1, 2, 3, 4], np.uint8))])
    r = subprocess.run([CONGA, "-i", "r.bam", "--out", "o", "--ref", "ref.fa", "--sonic", "a.cga", "--dump-reads"], cwd=d,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "1\t3\t5014\t6" in r.stdout


@pytest.mark.gpu
def test_cli_run_on_a_derived_annotation_matches_the_oracle(tmp_path, oracle):
    """FASTA -> conga-annotate -> conga: same bytes as the oracle fed with the GC windows computed here in numpy."""
    from conga_amd import synth
    d = str(tmp_path)
    rng = np.random.default_rng(11)
    L = 300_000
    # GC content that drifts along the chromosome, with a gap
    p = np.clip(0.41 + 0.15 * np.sin(np.arange(L) / 9000.0), 0.05, 0.95)
    seq = np.where(rng.random(L) < p, rng.choice(np.frombuffer(b"GC", np.uint8), L), rng.choice(np.frombuffer(b"AT", np.uint8), L))
    seq[50_000:61_000] = ord("N")
    formats.write_fasta(os.path.join(d, "ref.fa"), [("1", seq.tobytes())])
    assert annotate(["--ref", "ref.fa", "--out", "a.cga"], d).returncode == 0
    gc = gc_windows(seq.tobytes(), 100)
    c = synth.make_chrom("1", L, cov=3.0, n_dels=40, n_dups=10, gaps=False)
    formats.write_bam(os.path.join(d, "r.bam"), "S", [("1", L, c.pos, c.mapq)])
    synth.write_bed(os.path.join(d, "dels.bed"), [("1", s, e) for s, e in zip(c.del_start, c.del_end)])
    synth.write_bed(os.path.join(d, "dups.bed"), [("1", s, e) for s, e in zip(c.dup_start, c.dup_end)])
    r = subprocess.run([CONGA, "-i", "r.bam", "--out", "got", "--ref", "ref.fa", "--sonic", "a.cga", "--dels", "dels.bed",
                        "--dups", "dups.bed"], cwd=d, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ds = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dels.bed"), "1", 1000))
    us = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dups.bed"), "1", 1000))
    rd, _ = oracle.count_reads(L, c.pos, c.mapq, -1)
    E, _, _ = oracle.calc_mean_per_chr(rd, gc)
    oracle.find_depths(rd, None, gc, E, "D", ds)
    oracle.find_depths(rd, None, gc, E, "E", us)
    want = [os.path.join(d, "want_%s.bed" % k) for k in ("svs", "dels", "dups")]
    oracle.output_svs("1", ds, us, want[0], want[1], want[2], have_mappability=False, c_score=0.5, write_headers=True)
    for k, w in zip(("svs", "dels", "dups"), want):
        assert open(os.path.join(d, "got_%s.bed" % k), "rb").read() == open(w, "rb").read(), k
