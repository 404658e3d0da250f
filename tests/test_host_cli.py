"""The `conga` command line (conga_amd/host): flags, messages and exit codes of the reference
(cmdline.c, svdepth.c), its BED / BAM / annotation readers, and -- on the GPU -- the three output files
byte for byte against the oracle's writer (likelihood.c:172-288, bam_data.c:235-249)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conga_amd import formats, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONGA = os.path.join(ROOT, "conga_amd", "host", "conga")


def run(args, cwd, env=None):
    return subprocess.run([CONGA] + args, cwd=cwd, capture_output=True, text=True, timeout=600, env=env)


def make_inputs(d, with_bam=True):
    specs = [("1", 400_000, 30, 8), ("2", 250_000, 20, 5), ("X", 100_000, 5, 0), ("3", 90_000, 0, 0), ("4", 300_000, 25, 6)]
    cs = [synth.make_chrom(n, L, cov=2.0, n_dels=nd, n_dups=nu, mappability=True, gaps=False) for n, L, nd, nu in specs]
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    in_bam = [c for c in cs if c.name != "4"]          # chromosome 4 is missing from the alignment file
    formats.write_tuples(os.path.join(d, "r.ctp"), "NA00001", [(c.name, c.length, c.pos, c.mapq) for c in in_bam])
    if with_bam:
        formats.write_bam(os.path.join(d, "r.bam"), "NA00001", [(c.name, c.length, c.pos, c.mapq) for c in in_bam], unplaced=5)
    rows_d, rows_u, rows_m = [], [], []
    rng = np.random.default_rng(1)
    for c in cs:
        rows_d += [(c.name, s, e) for s, e in zip(c.del_start, c.del_end)]
        rows_u += [(c.name, s, e) for s, e in zip(c.dup_start, c.dup_end)]
        rows_m += [(c.name, s, e, "%g" % v) for s, e, v in zip(c.map_start, c.map_end, c.map_val)]
    rng.shuffle(rows_d)                                   # the reference sorts per chromosome
    synth.write_bed(os.path.join(d, "dels.bed"), [("#chr", "start", "end")] + rows_d)
    synth.write_bed(os.path.join(d, "dups.bed"), rows_u)
    synth.write_bed(os.path.join(d, "map.bed"), rows_m)
    return cs, in_bam


def test_help_version_and_required_options(tmp_path):
    d = str(tmp_path)
    r = run([], d)
    assert r.returncode == 0 and "--dels [BED file]" in r.stdout and "--mapability [BED file]" in r.stdout
    r = run(["--help"], d)
    assert r.returncode == 0 and "CONGA (COpy Number variation Genotyping in Ancient genomes)" in r.stdout
    r = run(["--version"], d)
    assert r.returncode == 0 and "CONGA Version" in r.stderr
    r = run(["-i", "x.bam", "--ref", "r.fa"], d)
    assert r.returncode == 1 and "[CONGA CMDLINE ERROR] Please enter the output file name prefix using the --out option." in r.stderr
    r = run(["-i", "x.bam", "--out", "o"], d)
    assert r.returncode == 1 and "Please enter reference genome file (FASTA) using the --ref option." in r.stderr
    assert os.path.exists(os.path.join(d, "conga.log"))  # always created in the working directory (svdepth.c:30)


def test_missing_files_exit_like_print_error(tmp_path):
    d = str(tmp_path)
    r = run(["-i", "nope.bam", "--out", "o", "--ref", "r.fa", "--sonic", "nope.sonic"], d)
    assert r.returncode == 1 and "Invoke parameter -h for help." in r.stderr


def test_defaults_are_announced_and_out_is_split(tmp_path):
    d = str(tmp_path)
    make_inputs(d, with_bam=False)
    os.mkdir(os.path.join(d, "res"))
    r = run(["-i", "r.ctp", "--out", "res/sampleA", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed",
             "--dump-intervals", "1"], d)
    assert r.returncode == 0
    assert "Minimum size of an SV is set to 1000" in r.stderr and "Minimum size of a read is set to 60" in r.stderr
    assert "[CONGA INFO] Working directory: res/" in r.stderr and "prefix: sampleA" in r.stderr


def test_interval_loading_matches_the_oracle(tmp_path, oracle):
    d = str(tmp_path)
    make_inputs(d, with_bam=False)
    for chrom in ("1", "2", "X", "9"):
        for minsize in (None, 3000):
            args = ["-i", "r.ctp", "--out", "o", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups",
                    "dups.bed", "--dump-intervals", chrom] + (["--min-sv-size", str(minsize)] if minsize else [])
            r = run(args, d)
            assert r.returncode == 0
            got = [tuple(l.split("\t")) for l in r.stdout.splitlines() if l.startswith(("DEL", "DUP"))]
            want = []
            for tag, f in (("DEL", "dels.bed"), ("DUP", "dups.bed")):
                svs = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, f), chrom, minsize or 1000))
                want += [(tag, chrom, str(s), str(e)) for s, e in zip(svs["start"], svs["end"])]
            assert got == want


def test_bam_and_tuple_readers_yield_the_same_records(tmp_path):
    d = str(tmp_path)
    cs, in_bam = make_inputs(d)
    outs = []
    for inp in ("r.ctp", "r.bam"):
        r = run(["-i", inp, "--out", "o", "--ref", "r.fa", "--sonic", "a.cga", "--dump-reads"], d)
        assert r.returncode == 0, r.stderr
        outs.append([l for l in r.stdout.splitlines() if "\t" in l and not l.startswith(("BAM", "Ref", "SONIC"))])
    assert outs[0] == outs[1]
    lines = dict(l.split("\t", 1) for l in outs[0])
    assert lines["sample"] == "NA00001"
    assert lines["4"] == "missing" and "X" not in lines
    for c in in_bam:
        if c.name != "X":
            assert lines[c.name] == "%d\t%d\t%d" % (len(c.pos), c.pos.astype(np.int64).sum(), c.mapq.astype(np.int64).sum())


def _reframe_bgzf(src, dst, payloads, extra_subfield=False, empty_every=0):
    """Inflate a BAM with Python's gzip (multi-member gzip is what BGZF is) and write the same bytes back in blocks
    of the given payload sizes (cycled), optionally with a second gzip extra subfield and empty blocks in between."""
    import gzip
    import struct
    import zlib
    raw = gzip.decompress(open(src, "rb").read())

    def block(data):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = co.compress(data) + co.flush()
        extra = b"BC\x02\x00" + b"\x00\x00"
        if extra_subfield:
            extra = b"XY\x03\x00abc" + extra  # another subfield in front of BC (RFC 1952 allows any number)
        bsize = 12 + len(extra) + len(cdata) + 8 - 1
        extra = extra[:-2] + struct.pack("<H", bsize)
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", len(extra)) + extra + cdata
                + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    out, at, k = [], 0, 0
    while at < len(raw):
        n = payloads[k % len(payloads)]
        out.append(block(raw[at:at + n]))
        at += n
        k += 1
        if empty_every and k % empty_every == 0:
            out.append(block(b""))
    out.append(formats._BGZF_EOF)
    open(dst, "wb").write(b"".join(out))
    return raw


def test_bam_reader_takes_any_valid_bgzf_framing_and_checks_the_crc(tmp_path):
    """The writer's files are valid multi-member gzip (checked with Python's gzip, which shares no code with the C++
    reader); the reader gives the same records for blocks of 1 byte .. 64 KiB - 1, records and header fields split
    across blocks, empty blocks inside the file, extra gzip subfields; a block whose CRC32 is wrong is an error."""
    d = str(tmp_path)
    make_inputs(d)
    args = ["--out", "o", "--ref", "r.fa", "--sonic", "a.cga", "--dump-reads"]
    want = run(["-i", "r.bam"] + args, d)
    assert want.returncode == 0, want.stderr
    raw = None
    for name, kw in (("tiny.bam", dict(payloads=[1, 2, 3, 5, 7, 64, 4096])),
                     ("big.bam", dict(payloads=[65535])),  # 0xFFFF: BGZF's limit (incompressible data would not fit; this one does)
                     ("odd.bam", dict(payloads=[33, 65000, 1], extra_subfield=True, empty_every=3))):
        raw = _reframe_bgzf(os.path.join(d, "r.bam"), os.path.join(d, name), **kw)
        got = run(["-i", name] + args, d)
        assert got.returncode == 0, (name, got.stderr)
        assert got.stdout.replace(name, "r.bam") == want.stdout, name
    assert raw[:4] == b"BAM\x01"
    data = bytearray(open(os.path.join(d, "odd.bam"), "rb").read())
    first_len = 1 + (data[12 + 7 + 4] | (data[12 + 7 + 5] << 8))  # BSIZE + 1 of the first block (extra: XY(7) then BC)
    for name, at in (("bad_first.bam", first_len - 8), ("bad_later.bam", len(data) - 28 - 8)):  # a block's CRC32 field
        bad_bytes = bytearray(data)
        bad_bytes[at] ^= 0x40
        open(os.path.join(d, name), "wb").write(bytes(bad_bytes))
        bad = run(["-i", name] + args, d)
        assert bad.returncode != 0 and "CRC32" in bad.stderr, (name, bad.stderr)


def test_both_versions_of_the_tuple_container_read_the_same(tmp_path):
    """CONGATP2 pads every array to 16 bytes; CONGATP1 files (arrays wherever the header ends, usually unaligned) still load."""
    d = str(tmp_path)
    cs, in_bam = make_inputs(d, with_bam=False)
    formats.write_tuples(os.path.join(d, "old.ctp"), "NA00001", [(c.name, c.length, c.pos, c.mapq) for c in in_bam], aligned=False)
    assert open(os.path.join(d, "old.ctp"), "rb").read(8) == b"CONGATP1" and open(os.path.join(d, "r.ctp"), "rb").read(8) == b"CONGATP2"
    outs = [run(["-i", f, "--out", "o", "--ref", "r.fa", "--sonic", "a.cga", "--dump-reads"], d) for f in ("r.ctp", "old.ctp")]
    assert outs[0].returncode == 0 and outs[1].returncode == 0, outs[1].stderr
    assert outs[0].stdout.replace("r.ctp", "X") == outs[1].stdout.replace("old.ctp", "X")
    assert "1\t%d\t" % len(in_bam[0].pos) in outs[0].stdout


def test_reads_past_the_annotated_length_are_not_returned(tmp_path):
    """sam_itr_queryi(idx, tid, 0, L) yields only pos < L (L from the annotation, not the BAM header)."""
    d = str(tmp_path)
    pos = np.array([5, 10, 999, 1000, 1500], np.int32)
    mapq = np.array([1, 2, 3, 4, 5], np.uint8)
    formats.write_annotation(os.path.join(d, "a.cga"), [("1", 1000, np.full(10, 40, np.uint8), [], [])])
    formats.write_bam(os.path.join(d, "r.bam"), "s", [("1", 2000, pos, mapq)])
    r = run(["-i", "r.bam", "--out", "o", "--ref", "r.fa", "--sonic", "a.cga", "--dump-reads"], d)
    assert "1\t3\t1014\t6" in r.stdout


def _oracle_files(oracle, d, cs, in_bam, *, dels, dups, with_map, mq, c_score=0.5):
    paths = [os.path.join(d, "want_%s.bed" % k) for k in ("svs", "dels", "dups")]
    first = True
    names_in_bam = {c.name for c in in_bam}
    any_rows = False
    for c in cs:
        if "X" in c.name or "Y" in c.name or c.name not in names_in_bam:
            continue
        ds = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dels.bed"), c.name, 1000)) if dels else None
        us = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dups.bed"), c.name, 1000)) if dups else None
        if (0 if ds is None else len(ds)) + (0 if us is None else len(us)) == 0:
            continue
        rd, _ = oracle.count_reads(c.length, c.pos, c.mapq, mq)
        E, _, _ = oracle.calc_mean_per_chr(rd, c.gc)
        m = oracle.load_mappability_regions(os.path.join(d, "map.bed"), c.name, c.length)[0] if with_map else None
        if ds is not None:
            oracle.find_depths(rd, m, c.gc, E, "D", ds)
        if us is not None:
            oracle.find_depths(rd, m, c.gc, E, "E", us)
        oracle.output_svs(c.name, ds, us, paths[0], paths[1] if dels else None, paths[2] if dups else None,
                          have_mappability=with_map, c_score=c_score, write_headers=first)
        first = False
        any_rows = True
    assert any_rows
    return paths


@pytest.mark.gpu
@pytest.mark.parametrize("inp,dels,dups,with_map,mq,c_score", [
    ("r.ctp", True, True, True, -1, None),
    ("r.bam", True, False, False, -1, None),
    ("r.bam", True, True, False, 30, "0.3"),
    ("r.ctp", False, True, True, 0, None),
])
def test_cli_outputs_are_byte_identical_to_the_oracle(tmp_path, oracle, inp, dels, dups, with_map, mq, c_score):
    d = str(tmp_path)
    cs, in_bam = make_inputs(d)
    args = ["-i", inp, "--out", "got", "--ref", "r.fa", "--sonic", "a.cga"]
    if dels:
        args += ["--dels", "dels.bed"]
    if dups:
        args += ["--dups", "dups.bed"]
    if with_map:
        args += ["--mappability", "map.bed"]
    if mq >= 0:
        args += ["--min-mapq", str(mq)]
    if c_score:
        args += ["--c-score", c_score]
    r = run(args, d)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Cannot find chromosome name 4 in BAM/CRAM NA00001" in r.stderr
    assert "Thank you" in r.stderr and "Hope to see you again..." in r.stderr
    want = _oracle_files(oracle, d, cs, in_bam, dels=dels, dups=dups, with_map=with_map, mq=mq,
                         c_score=float(np.float32(c_score)) if c_score else 0.5)
    for k, have, w in (("svs", True, want[0]), ("dels", dels, want[1]), ("dups", dups, want[2])):
        got_path = os.path.join(d, "got_%s.bed" % k)
        assert os.path.exists(got_path) == have
        if have:
            assert open(got_path, "rb").read() == open(w, "rb").read(), k
    log = open(os.path.join(d, "conga.log")).read()
    assert "#CreationDate=" in log and "Read Count:" in log and "mean=" in log


@pytest.mark.gpu
@pytest.mark.parametrize("inp,n", [("r.bam", 2), ("r.ctp", 3), ("r.bam", 7)])
def test_cli_gpus_shards_chromosomes_and_keeps_every_byte(tmp_path, inp, n):
    """`--gpus N` (chromosomes -> contexts, output by the calling thread in annotation order): the three
    files, conga.log and the progress text are those of the one-context run.  On a one-GPU box the contexts share it."""
    d = str(tmp_path)
    make_inputs(d)
    base = ["-i", inp, "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed",
            "--mappability", "map.bed"]
    one = run(base + ["--out", "one"], d)
    assert one.returncode == 0, one.stderr[-2000:]
    log_one = open(os.path.join(d, "conga.log")).read()
    many = run(base + ["--out", "many", "--gpus", str(n)], d)
    assert many.returncode == 0, many.stderr[-2000:]
    log_many = open(os.path.join(d, "conga.log")).read()
    for k in ("svs", "dels", "dups"):
        a = open(os.path.join(d, "one_%s.bed" % k), "rb").read()
        assert a.count(b"\n") > 1
        assert a == open(os.path.join(d, "many_%s.bed" % k), "rb").read(), k
    assert log_one == log_many

    def progress(text, prefix):
        return [ln.replace(prefix, "OUT") for ln in text.splitlines() if ln and not ln.startswith("[CONGA] --gpus")]
    assert progress(one.stderr, "one") == progress(many.stderr, "many")


@pytest.mark.gpu
def test_cli_bam_decoded_on_the_gpu_matches_the_host_decoders(tmp_path, oracle):
    """A BAM with its .bai: the compressed blocks are inflated and the records walked on the GPU (conga_reads_bgzf).  Same
    files as with the host decoders, byte for byte, and as the oracle's; an index whose linear offsets skip records is
    noticed and the host takes over."""
    d = str(tmp_path)
    rng = np.random.default_rng(9)
    cs = [synth.make_chrom(n, L, cov=3.0, n_dels=nd, n_dups=nu, gaps=True) for n, L, nd, nu in
          (("1", 900_000, 40, 8), ("2", 300_000, 15, 3), ("3", 1_300_000, 50, 10))]
    # pile-ups on a window boundary of the index and a stretch without reads
    extra = np.sort(np.concatenate([np.full(400, 16384 * 9), np.full(300, 16384 * 9 - 1)])).astype(np.int32)
    reads = []
    for c in cs:
        pos = np.sort(np.concatenate([c.pos, extra])).astype(np.int32)
        pos = pos[(pos < 200_000) | (pos > 262_144 + 5)]
        reads.append((c.name, c.length, pos, rng.integers(0, 61, len(pos)).astype(np.uint8)))
    formats.write_bam(os.path.join(d, "r.bam"), "S", reads, index=True, block_payload=9000, unplaced=6)
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    synth.write_bed(os.path.join(d, "dups.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.dup_start, c.dup_end)])
    base = ["-i", "r.bam", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed", "--min-mapq", "10"]

    def cli(out, **env):
        r = subprocess.run([CONGA] + base + ["--out", out], cwd=d, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        return r, [open(os.path.join(d, "%s_%s.bed" % (out, k)), "rb").read() for k in ("svs", "dels", "dups")]
    r_gpu, gpu = cli("gpu", CONGA_GPU_BAM="1", CONGA_TIMING="1")  # (by default only pieces of >= 32 768 blocks go this way)
    assert "decoding on the host" not in r_gpu.stderr and r_gpu.stderr.count("conga_reads_bgzf:") == 1
    r_host, host = cli("host", CONGA_GPU_BAM="0")
    assert gpu == host and gpu[1].count(b"\n") > 50
    # the overlapped form of the upload (pinned pieces filled by host threads, one inflate launch per 8 pieces), forced onto
    # this small file with 8 KB pieces, and the plain form; the round-1 kernel (one block per lane) once more
    for tag, env in (("ovl", dict(CONGA_BGZF_OVERLAP="1", CONGA_BGZF_PIECE_KB="8")), ("ovl1", dict(CONGA_BGZF_OVERLAP="1", CONGA_BGZF_PIECE_KB="8", CONGA_BGZF_COPY_THREADS="1", CONGA_DEBUG="1")),
                     ("plain", dict(CONGA_BGZF_OVERLAP="0")), ("lane", dict(CONGA_BGZF_KERNEL="lane", CONGA_DEBUG="1")),
                     # (the file read with pread instead of mapped)
                     ("unmapped", dict(CONGA_BAM_MMAP="0")), ("ovlfd", dict(CONGA_BAM_MMAP="0", CONGA_BGZF_OVERLAP="1", CONGA_BGZF_PIECE_KB="8")),
                     # the block table walked in parts from block starts the index knows (as for files above 64 MB), read with pread and mapped
                     ("parts", dict(CONGA_BAM_PARALLEL_MIN_KB="0")), ("partsfd", dict(CONGA_BAM_PARALLEL_MIN_KB="0", CONGA_BAM_MMAP="0"))):
        r_x, x = cli(tag, CONGA_GPU_BAM="1", CONGA_TIMING="1", **env)
        assert x == gpu and "decoding on the host" not in r_x.stderr and r_x.stderr.count("conga_reads_bgzf:") == 1, tag
        assert ("overlapped" in r_x.stderr) == tag.startswith("ovl"), tag
        assert ("parts walked side by side\n" in r_x.stderr) == tag.startswith("parts"), (tag, r_x.stderr[-1500:])
    # a piece limit below the whole stretch but above each chromosome's: one GPU call per chromosome instead of one for all
    size = os.path.getsize(os.path.join(d, "r.bam"))
    r_each, each = cli("each", CONGA_GPU_BAM="1", CONGA_TIMING="1", CONGA_GPU_BAM_MAX_MB="%.4f" % (size * 0.75 / 1048576))
    assert each == gpu and r_each.stderr.count("conga_reads_bgzf:") == 3 and "decoding on the host" not in r_each.stderr
    for rr in (r_gpu, r_host):
        import re
        # count_reads_bam's closing line counts the records that passed `qual > mq_threshold` (bam_data.c:205-218)
        assert [int(m) for m in re.findall(r"\((\d+) reads, 0 split-reads\)", rr.stderr)] == [int((x[3] > 10).sum()) for x in reads]
    # the oracle on the same records
    want = [os.path.join(d, "want_%s.bed" % k) for k in ("svs", "dels", "dups")]
    first = True
    for c, (_, _, pos, mapq) in zip(cs, reads):
        ds = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dels.bed"), c.name, 1000))
        us = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dups.bed"), c.name, 1000))
        rd, _ = oracle.count_reads(c.length, pos, mapq, 10)
        E, _, _ = oracle.calc_mean_per_chr(rd, c.gc)
        oracle.find_depths(rd, None, c.gc, E, "D", ds)
        oracle.find_depths(rd, None, c.gc, E, "E", us)
        oracle.output_svs(c.name, ds, us, want[0], want[1], want[2], have_mappability=False, c_score=0.5, write_headers=first)
        first = False
    assert gpu == [open(w, "rb").read() for w in want]
    # linear offsets that point behind records: refused by the GPU stage and by the parallel host reader, still right
    import struct
    bai = bytearray(open(os.path.join(d, "r.bam.bai"), "rb").read())
    at = 8
    (n_bin,) = struct.unpack_from("<i", bai, at)
    at += 4
    for _ in range(n_bin):
        (_, n_chunk) = struct.unpack_from("<Ii", bai, at)
        at += 8 + 16 * n_chunk
    (n_intv,) = struct.unpack_from("<i", bai, at)
    at += 4
    lin = list(struct.unpack_from("<%dQ" % n_intv, bai, at))
    lin0 = list(lin)
    for w in range(n_intv // 2, n_intv):
        lin[w] = lin[min(n_intv - 1, w + 7)]
    struct.pack_into("<%dQ" % n_intv, bai, at, *lin)
    open(os.path.join(d, "r.bam.bai"), "wb").write(bytes(bai))
    r_bad, bad = cli("bad", CONGA_BAM_SEGMENTS="6", CONGA_GPU_BAM="1")
    assert "decoding on the host" in r_bad.stderr and bad == gpu
    # one linear offset that is not a block's start at all: the part of the block table that starts there does not find a
    # block (or does not arrive at the next part's start), the table is walked from the front, and whatever the GPU stage
    # makes of that start point, the files are right
    lin2 = list(lin0)
    used = [w for w in range(n_intv) if lin0[w] != 0]
    for w in used[len(used) // 3: 2 * len(used) // 3]:
        lin2[w] = lin0[w] + (3 << 16)
    struct.pack_into("<%dQ" % n_intv, bai, at, *lin2)
    open(os.path.join(d, "r.bam.bai"), "wb").write(bytes(bai))
    r_odd, odd = cli("odd", CONGA_GPU_BAM="1", CONGA_TIMING="1", CONGA_BAM_PARALLEL_MIN_KB="0")
    assert odd == gpu and "walked from the front instead" in r_odd.stderr, r_odd.stderr[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("payload,level,strategy", [
    (65280, 0, 0),      # stored blocks
    (60000, 9, 0),      # dynamic Huffman, best compression
    (60000, 6, 4),      # zlib.Z_FIXED: fixed-Huffman blocks
    (60000, 6, 2),      # zlib.Z_HUFFMAN_ONLY: literals only
    (311, 1, 0),        # hundreds of tiny blocks, records and their fields split across them
])
def test_gpu_bam_decode_takes_every_kind_of_block(tmp_path, payload, level, strategy):
    """conga_reads_bgzf on BGZF blocks of every deflate block type and of odd sizes: same output files as the host
    decoders, and the GPU stage did the work (it does not quietly decline)."""
    d = str(tmp_path)
    cs = [synth.make_chrom(n, L, cov=2.0, n_dels=nd, gaps=True) for n, L, nd in (("1", 400_000, 20), ("2", 250_000, 12))]
    formats.write_bam(os.path.join(d, "r.bam"), "S", [(c.name, c.length, c.pos, c.mapq) for c in cs], index=True,
                      block_payload=payload, level=level, strategy=strategy, unplaced=3)
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    outs = {}
    # (the tiny-block file also goes through with 64 lanes only: every lane then inflates dozens of blocks in turn)
    for tag, env in (("gpu", {"CONGA_GPU_BAM": "1", "CONGA_TIMING": "1", **({"CONGA_BGZF_LANES": "64"} if payload < 1000 else {})}),
                     ("host", {"CONGA_GPU_BAM": "0"})):
        r = subprocess.run([CONGA, "-i", "r.bam", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--out", tag], cwd=d,
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        if tag == "gpu":
            assert "decoding on the host" not in r.stderr and "conga_reads_bgzf:" in r.stderr, r.stderr[-1500:]
        outs[tag] = [open(os.path.join(d, "%s_%s.bed" % (tag, k)), "rb").read() for k in ("svs", "dels")]
    assert outs["gpu"] == outs["host"] and outs["gpu"][1].count(b"\n") > 20


@pytest.mark.gpu
def test_cli_split_reads_rp(tmp_path, oracle):
    """--rp with --dups: FASTA + BAM sequences -> READ_PAIR columns and the `rp > rp_support` rule of _svs.bed
    (likelihood.c:243-279), byte-identical to the oracle -- with the records handed over by the host decoders, read in place
    from the stream the GPU inflated (one call for all chromosomes, one per chromosome), and for the samples of a cohort."""
    from test_gpu_split_reads import make_case
    d = str(tmp_path)
    c = make_case(seed=5, L=400_000, n_normal=4000)    # (every planted SV inside the chromosome: the decode on the GPU leaves a file
    # with records behind the annotation's chromosome end to the host decoders)
    # one chromosome with the planted junctions plus a plain one
    plain = synth.make_chrom("2", 120_000, cov=6.0, n_dels=8, n_dups=3, gaps=False)
    gc1 = np.full((c["L"] + 99) // 100, 41, np.uint8)
    formats.write_annotation(os.path.join(d, "a.cga"), [("1", c["L"], gc1, c["sat_s"][:2], c["sat_e"][:2]),
                                                          ("2", plain.length, plain.gc, [], [])])
    rng = np.random.default_rng(0)
    ref2 = rng.choice(np.frombuffer(b"ACGT", np.uint8), plain.length).tobytes()
    formats.write_fasta(os.path.join(d, "ref.fa"), [("1", c["ref_lower"]), ("2", ref2)])
    lut = np.full(256, 15, np.uint8)
    for k, v in {65: 1, 67: 2, 71: 4, 84: 8}.items():
        lut[k] = v
    r2 = np.frombuffer(ref2, np.uint8)
    ok2 = plain.pos + 100 < plain.length
    pos2, mq2 = plain.pos[ok2], plain.mapq[ok2]
    lq2 = np.full(len(pos2), 100, np.int32)
    off2 = (np.arange(len(pos2), dtype=np.uint64) * 100)
    codes2 = lut[np.concatenate([r2[p:p + 100] for p in pos2])] if len(pos2) else np.zeros(0, np.uint8)
    qual2 = np.full(len(codes2), 30, np.uint8)
    formats.write_bam(os.path.join(d, "r.bam"), "S1",
                      [("1", c["L"], c["pos"], c["mapq"], c["flag"]), ("2", plain.length, pos2, mq2)],
                      records={"1": (c["lq"], c["codes"], c["qual"], c["off"]), "2": (lq2, codes2, qual2, off2)},
                      index=True, block_payload=30_000, unplaced=3)
    synth.write_bed(os.path.join(d, "dels.bed"), [("1", s, e) for s, e in c["dels"]] + [("2", s, e) for s, e in zip(plain.del_start, plain.del_end)])
    synth.write_bed(os.path.join(d, "dups.bed"), [("1", s, e) for s, e in c["dups"]] + [("2", s, e) for s, e in zip(plain.dup_start, plain.dup_end)])
    common = ["--ref", "ref.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed", "--rp", "3"]

    def files(prefix):
        return [open(os.path.join(d, "%s_%s.bed" % (prefix, k)), "rb").read() for k in ("svs", "dels", "dups")]

    def counts(stderr):  # count_reads_bam's closing line per chromosome, and the rows paired
        import re
        return re.findall(r"\((\d+) reads, (\d+) split-reads\)", stderr), re.findall(r"CONGA paired (\d+) single-end reads", stderr)
    # the host decoders hand every record over through the pinned staging (conga_split_reads_commit) ...
    r = run(["-i", "r.bam", "--out", "got"] + common, d, env=dict(os.environ, CONGA_GPU_BAM="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    assert "CONGA paired" in r.stderr
    # ... the decode on the GPU leaves them where the inflate put them and the split-read stage reads them there
    # (bam_data.c:201-216 with find_split_reads inside the BAM loop: no record ever exists on the host)
    r_gpu = run(["-i", "r.bam", "--out", "gpu"] + common, d, env=dict(os.environ, CONGA_GPU_BAM="1", CONGA_TIMING="1"))
    assert r_gpu.returncode == 0, r_gpu.stderr[-3000:]
    assert "decoding on the host" not in r_gpu.stderr and r_gpu.stderr.count("conga_reads_bgzf:") == 1
    assert files("gpu") == files("got") and counts(r_gpu.stderr) == counts(r.stderr) and len(counts(r.stderr)[0]) == 2
    # one GPU call per chromosome (the piece limit below the whole stretch): the second call's stream goes behind the first's
    size = os.path.getsize(os.path.join(d, "r.bam"))
    r_each = run(["-i", "r.bam", "--out", "each"] + common, d,
                 env=dict(os.environ, CONGA_GPU_BAM="1", CONGA_TIMING="1", CONGA_GPU_BAM_MAX_MB="%.4f" % (size * 0.9 / 1048576)))
    assert r_each.returncode == 0, r_each.stderr[-3000:]
    assert r_each.stderr.count("conga_reads_bgzf:") == 2 and "decoding on the host" not in r_each.stderr
    assert files("each") == files("got") and counts(r_each.stderr) == counts(r.stderr)
    # a cohort with --rp: the reference sequences and the 10-mer indexes stay with the layout, a sample brings its records
    keep = np.arange(len(c["pos"])) % 3 != 1            # a second sample: two thirds of the first one's reads
    lq_b = c["lq"][keep]
    off_b = np.concatenate([[0], np.cumsum(lq_b)[:-1]]).astype(np.uint64)
    per_base = np.repeat(keep, c["lq"])
    formats.write_bam(os.path.join(d, "b.bam"), "S2",
                      [("1", c["L"], c["pos"][keep], c["mapq"][keep], c["flag"][keep]), ("2", plain.length, pos2[::2], mq2[::2])],
                      records={"1": (lq_b, c["codes"][per_base], c["qual"][per_base], off_b),
                               "2": (lq2[::2], codes2.reshape(-1, 100)[::2].reshape(-1), qual2.reshape(-1, 100)[::2].reshape(-1), off2[:len(lq2[::2])])},
                      index=True, block_payload=30_000)
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("r.bam\nb.bam\nr.bam\n")
    for decode in ("1", "0", "1ahead"):
        env = dict(os.environ, CONGA_GPU_BAM=decode[0], CONGA_TIMING="1")
        if decode == "1ahead":
            # the cohort's pipeline with split reads on files of test size: the overlapped upload forced (pieces of 64 KB), the next
            # sample's bytes named ahead and brought up beside the sample in front -- but NOT inflated ahead: the context holds
            # reference text, the sample in front maps its split reads on the inflated stream in place
            env.update(CONGA_BGZF_OVERLAP="1", CONGA_BGZF_PIECE_KB="64", CONGA_BGZF_CHECK_TABLE="1")
        rc = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "co" + decode] + common, cwd=d, capture_output=True, text=True, timeout=600, env=env)
        assert rc.returncode == 0, rc.stderr[-3000:]
        assert rc.stderr.count("10-mer indexes of") == 1, "the indexes are built once for the cohort"
        if decode == "1ahead":
            assert "decoding on the host" not in rc.stderr and rc.stderr.count("conga_reads_bgzf:") == 3, rc.stderr[-3000:]
            assert "inflate launches made ahead" not in rc.stderr, rc.stderr[-3000:]
            assert rc.stderr.count("named ahead") >= 1, rc.stderr[-3000:]   # (how far ahead depends on the threads' timing)
            assert files("co1ahead.r") == files("got") and files("co1ahead.b") == files("oneb1")
            continue
        one = run(["-i", "b.bam", "--out", "oneb" + decode] + common, d, env=env)
        assert one.returncode == 0, one.stderr[-3000:]
        assert files("co%s.r" % decode) == files("got") and files("co%s.b" % decode) == files("oneb" + decode)
        assert files("oneb" + decode) != files("got")
    assert files("oneb1") == files("oneb0")

    paths = [os.path.join(d, "want_%s.bed" % k) for k in ("svs", "dels", "dups")]
    first = True
    for name, L, gc, pos, mapq, flag, lq, codes, qual, off, ref, ss, se in (
            ("1", c["L"], gc1, c["pos"], c["mapq"], c["flag"], c["lq"], c["codes"], c["qual"], c["off"], c["ref"], c["sat_s"][:2], c["sat_e"][:2]),
            ("2", plain.length, plain.gc, pos2, mq2, np.zeros(len(pos2), np.uint16), lq2, codes2, qual2, off2, ref2, [], [])):
        ds = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dels.bed"), name, 1000))
        us = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "dups.bed"), name, 1000))
        if len(ds) + len(us) == 0:
            continue
        rd, _ = oracle.count_reads(L, pos, mapq, -1)
        E, _, _ = oracle.calc_mean_per_chr(rd, gc)
        oracle.find_depths(rd, None, gc, E, "D", ds)
        oracle.find_depths(rd, None, gc, E, "E", us)
        rows, _ = oracle.split_read_rows(ref, ss, se, pos, mapq, flag, lq, off, codes, qual, -1, 60)
        oracle.count_read_pairs(rows, ds, us)
        oracle.output_svs(name, ds, us, *paths, have_mappability=False, no_sr=0, rp_support=3, write_headers=first)
        first = False
    for k, w in zip(("svs", "dels", "dups"), paths):
        assert open(os.path.join(d, "got_%s.bed" % k), "rb").read() == open(w, "rb").read(), k
    dup_rows = [l.split("\t") for l in open(os.path.join(d, "got_dups.bed")).read().splitlines()[1:]]
    assert max(int(x[5]) for x in dup_rows) > 3          # READ_PAIR support was found for a planted duplication


def test_bai_seek_gives_the_same_records_in_any_chromosome_order(tmp_path):
    """With a .bai the reader seeks to a reference's first chunk (sam_index_load + sam_itr_queryi, bam_data.c:259,293);
    the annotation may list chromosomes in another order than the BAM header."""
    d = str(tmp_path)
    cs = [synth.make_chrom(n, L, cov=3.0, gaps=False) for n, L in (("1", 150_000), ("2", 90_000), ("3", 200_000))]
    formats.write_bam(os.path.join(d, "r.bam"), "S", [(c.name, c.length, c.pos, c.mapq) for c in cs], index=True,
                      block_payload=20_000, unplaced=3)
    assert os.path.getsize(os.path.join(d, "r.bam.bai")) > 100
    order = [cs[2], cs[0], cs[1]]                       # annotation order differs from the BAM's
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in order])
    outs = {}
    for tag in ("with_bai", "picard_name", "no_bai"):
        if tag == "picard_name":                        # sample.bai instead of sample.bam.bai
            os.rename(os.path.join(d, "r.bam.bai"), os.path.join(d, "r.bai"))
        if tag == "no_bai":
            os.remove(os.path.join(d, "r.bai"))
        r = run(["-i", "r.bam", "--out", "o", "--ref", "r.fa", "--sonic", "a.cga", "--dump-reads"], d)
        assert r.returncode == 0, r.stderr
        assert "index\t" + {"with_bai": "r.bam.bai", "picard_name": "r.bai", "no_bai": "none"}[tag] in r.stdout
        outs[tag] = [l for l in r.stdout.splitlines() if l[:1].isdigit()]
    assert outs["with_bai"] == outs["no_bai"] == outs["picard_name"]
    want = ["%s\t%d\t%d\t%d" % (c.name, len(c.pos), c.pos.astype(np.int64).sum(), c.mapq.astype(np.int64).sum()) for c in order]
    assert outs["with_bai"] == want


def test_parallel_segment_decode_gives_the_same_records(tmp_path):
    """read_all: a chromosome's stretch of the BAM cut at windows of the .bai's linear index and decoded by several
    readers at once; any number of segments gives the records of the sequential reader, pile-ups that straddle a
    window boundary and empty stretches included, and a linear index that does not line up is noticed."""
    d = str(tmp_path)
    rng = np.random.default_rng(5)
    cs = []
    for n, L in (("1", 900_000), ("2", 300_000), ("3", 1_200_000)):
        pos = np.sort(np.concatenate([rng.integers(0, L, 20_000), np.full(300, 16384 * 7), np.full(200, 16384 * 7 - 1),
                                      rng.integers(500_000, 500_100, 500) if L > 600_000 else np.zeros(0, np.int64)])).astype(np.int32)
        pos = pos[(pos < 200_000) | (pos > 330_000)]                      # a stretch no read starts in
        cs.append((n, L, pos, rng.integers(0, 61, len(pos)).astype(np.uint8)))
    formats.write_bam(os.path.join(d, "r.bam"), "S", cs, index=True, block_payload=3000, unplaced=4)
    formats.write_annotation(os.path.join(d, "a.cga"), [(n, L, np.full((L + 99) // 100, 40, np.uint8), [], []) for n, L, _, _ in cs])
    args = ["-i", "r.bam", "--out", "o", "--ref", "r.fa", "--sonic", "a.cga", "--dump-reads"]
    want = ["%s\t%d\t%d\t%d" % (n, len(p), p.astype(np.int64).sum(), q.astype(np.int64).sum()) for n, L, p, q in cs]
    for k in ("1", "2", "3", "7", "16", "61"):
        r = subprocess.run([CONGA] + args, cwd=d, capture_output=True, text=True, timeout=600, env=dict(os.environ, CONGA_BAM_SEGMENTS=k))
        assert r.returncode == 0, r.stderr
        assert [l for l in r.stdout.splitlines() if l[:1].isdigit()] == want, k
        assert "does not line up" not in r.stderr
    # an index whose linear offsets point somewhere else: noticed, and the sequential reader takes over
    bai = bytearray(open(os.path.join(d, "r.bam.bai"), "rb").read())
    import struct
    at = 8
    (n_bin,) = struct.unpack_from("<i", bai, at)
    at += 4
    for _ in range(n_bin):
        (_, n_chunk) = struct.unpack_from("<Ii", bai, at)
        at += 8 + 16 * n_chunk
    (n_intv,) = struct.unpack_from("<i", bai, at)
    at += 4
    lin = list(struct.unpack_from("<%dQ" % n_intv, bai, at))
    for w in range(n_intv // 2, n_intv):                                   # second half: every window claims a LATER offset
        lin[w] = lin[min(n_intv - 1, w + 9)]                               # (records in between would be lost; an earlier one only costs time)
    struct.pack_into("<%dQ" % n_intv, bai, at, *lin)
    open(os.path.join(d, "r.bam.bai"), "wb").write(bytes(bai))
    r = subprocess.run([CONGA] + args, cwd=d, capture_output=True, text=True, timeout=600, env=dict(os.environ, CONGA_BAM_SEGMENTS="8"))
    assert r.returncode == 0, r.stderr
    assert [l for l in r.stdout.splitlines() if l[:1].isdigit()] == want
    assert "does not line up" in r.stderr


def test_fast_bed_parser_equals_the_literal_fgets_strtok_reader(tmp_path, oracle):
    """svs.cpp parses BEDs from an mmap on several threads; it must yield the rows of the reference's literal
    fgets(512) / strtok / atoi / atof reader (svs.c:7-240,317-377), which stays in the binary as the fallback."""
    d = str(tmp_path)
    rng = np.random.default_rng(4)
    lines = ["#chr\tstart\tend\tvalue", "", "   ", "1\t100\t2200\t0.5", "1 300   1400\t1", "  1\t+500\t1600\t.25",
             "1\t-5\t1200\t5.", "1\t1e3\t2500\t1e-1", "1\t2000\t3100\t0.333333abc", "1\t7\t", "1", "2\t10\t5000\t0.75",
             "1\t4000\t5200\t0.1\textra\tcolumns", "1\t6000\t7300\t0.123456789012345678", "1\t8000\t9100\tnan",
             "1\t9000\t10100\t-0.5", "X\t1\t2000\t1"]
    for _ in range(300_000):                                     # > 4 MB so the threaded path is taken
        s = int(rng.integers(0, 10_000_000))
        lines.append("%s\t%d\t%d\t%s" % (rng.choice(["1", "2", "21"]), s, s + int(rng.integers(1, 5000)),
                                         rng.choice(["1", "0.5", "0.333333", "0.25", "0.2", "0.1"])))
    open(os.path.join(d, "m.bed"), "w").write("\r\n".join(lines[:12]) + "\n" + "\n".join(lines[12:]))
    long_file = os.path.join(d, "long.bed")                      # one 600-character line: the reference splits it in two chunks
    open(long_file, "w").write("1\t100\t2200\t0.5\n" + "1\t300\t1400\t" + "0" * 600 + "\n1\t5\t1500\t1\n")
    outs = {}
    for literal in ("0", "1"):
        env = dict(os.environ)
        if literal == "1":
            env["CONGA_BED_LITERAL"] = "1"
        for f in ("m.bed", "long.bed"):
            for chrom in ("1", "2"):
                r = subprocess.run([CONGA, "-i", "x", "--out", "o", "--ref", "r.fa", "--mappability", f, "--dels", f,
                                    "--dump-mappability", chrom], cwd=d, capture_output=True, text=True, env=env)
                assert r.returncode == 0
                outs[(literal, f, chrom, "map")] = [l for l in r.stdout.splitlines() if l.startswith("MAP")]
                r = subprocess.run([CONGA, "-i", "x", "--out", "o", "--ref", "r.fa", "--dels", f, "--dump-intervals", chrom],
                                   cwd=d, capture_output=True, text=True, env=env)
                outs[(literal, f, chrom, "iv")] = [l for l in r.stdout.splitlines() if l.startswith("DEL")]
    for f in ("m.bed", "long.bed"):
        for chrom in ("1", "2"):
            for kind in ("map", "iv"):
                assert outs[("0", f, chrom, kind)] == outs[("1", f, chrom, kind)], (f, chrom, kind)
    assert len(outs[("0", "m.bed", "1", "map")]) > 90_000
    # and the interval side agrees with the oracle's reader too
    svs = oracle.sort_svs(oracle.load_known_SVs(os.path.join(d, "m.bed"), "1", 1000))
    assert outs[("0", "m.bed", "1", "iv")] == ["DEL\t1\t%d\t%d" % (s, e) for s, e in zip(svs["start"], svs["end"])]


@pytest.mark.gpu
@pytest.mark.parametrize("decode", ["gpu", "host", "gpu_each", "gpu_ahead", "gpu_ahead_small_pieces"])
def test_cli_cohort_keeps_the_engine_and_every_byte(tmp_path, decode):
    """`--cohort list` (an extension: the reference is started once per sample, svdepth.c:16-74): several BAMs in one process,
    the engine context kept from sample to sample -- the layout too when the samples select the same chromosomes
    (conga_sample_begin), handed over again when they do not.  Every sample's three files are those of its own run."""
    d = str(tmp_path)
    specs = [("1", 500_000, 30, 8), ("2", 300_000, 20, 5), ("3", 200_000, 12, 3)]
    cs = [synth.make_chrom(n, L, cov=1.0, n_dels=nd, n_dups=nu, mappability=True, gaps=False) for n, L, nd, nu in specs]
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    synth.write_bed(os.path.join(d, "dups.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.dup_start, c.dup_end)])
    synth.write_bed(os.path.join(d, "map.bed"), [(c.name, s, e, "%g" % v) for c in cs for s, e, v in zip(c.map_start, c.map_end, c.map_val)])
    samples = []
    for k, (cov, drop) in enumerate([(1.0, None), (3.0, None), (0.5, "2"), (2.0, None)]):  # the third sample lacks chromosome 2
        chroms = []
        for ci, c in enumerate(cs):
            if c.name == drop:
                continue
            rng = np.random.default_rng([k, ci, 5])
            pos, mapq = synth.make_reads(c.length, c.gc, c.step, cov, 100, rng)
            chroms.append((c.name, c.length, pos, mapq))
        formats.write_bam(os.path.join(d, "s%d.bam" % k), "S%d" % k, chroms, index=True, block_payload=20_000, unplaced=2)
        samples.append("s%d.bam" % k)
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("# BAM [prefix]\n%s\n%s\tnamed/two\n\n%s\n%s\n" % tuple(samples))
    os.mkdir(os.path.join(d, "named"))
    common = ["--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed", "--mappability", "map.bed"]
    env = dict(os.environ, CONGA_GPU_BAM="0" if decode == "host" else "1", CONGA_TIMING="1")
    if decode == "gpu_each":
        # a piece limit below the second (deepest) sample's stretch: that sample goes up chromosome by chromosome into the KEPT
        # layout (conga_sample_chrom names the chromosome of the context each call feeds), the others in one call each
        env["CONGA_GPU_BAM_MAX_MB"] = "%.4f" % (os.path.getsize(os.path.join(d, samples[1])) * 0.92 / 1048576)
    if decode.startswith("gpu_ahead"):
        # the pipeline of a cohort on files of test size: the overlapped upload forced (pieces of 64 KB -- or of 6 KB, smaller than a
        # BGZF block: headers and trailers straddle pieces, pieces hold no known start), every further sample's bytes named ahead
        # (conga_reads_bgzf_next_fd), the block table read off them by the engine -- checked against the one read from the file
        # (CONGA_BGZF_CHECK_TABLE) -- and inflated ahead into the spare output set
        env.update(CONGA_BGZF_OVERLAP="1", CONGA_BGZF_PIECE_KB="64" if decode == "gpu_ahead" else "6", CONGA_BGZF_CHECK_TABLE="1")
    r = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "co"] + common, cwd=d, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    if decode != "host":
        assert r.stderr.count("conga_reads_bgzf:") == (6 if decode == "gpu_each" else 4) and "decoding on the host" not in r.stderr
    if decode.startswith("gpu_ahead"):
        assert r.stderr.count("block table: the engine's and the file's agree") >= 2, r.stderr[-3000:]
        assert r.stderr.count("named ahead with its block table") >= 1, r.stderr[-3000:]   # (how far ahead depends on the threads' timing)
    prefixes = ["co.s0", "named/two", "co.s2", "co.s3"]
    for k, bam in enumerate(samples):
        one = subprocess.run([CONGA, "-i", bam, "--out", "one%d" % k] + common, cwd=d, capture_output=True, text=True, timeout=600, env=env)
        assert one.returncode == 0, one.stderr[-2000:]
        for kind in ("svs", "dels", "dups"):
            got = open(os.path.join(d, "%s_%s.bed" % (prefixes[k], kind)), "rb").read()
            want = open(os.path.join(d, "one%d_%s.bed" % (k, kind)), "rb").read()
            assert got == want and (kind != "dels" or got.count(b"\n") > 5), (k, kind)
    # the samples differ (it is not one sample's files four times)
    assert open(os.path.join(d, "co.s0_dels.bed"), "rb").read() != open(os.path.join(d, "co.s3_dels.bed"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("mq", [None, 20])
def test_cli_cohort_further_samples_from_the_host_go_packed_and_keep_every_byte(tmp_path, mq):
    """The seam is count_reads_bam (bam_data.c:192-221).  A cohort's further samples that are decoded on the host -- tuple containers,
    BAMs with CONGA_GPU_BAM=0 -- are handed over whole: conga_packer_start_v (one position array per chromosome, as the container /
    the decoders leave them) -> conga_sample_reads_packed, instead of the staging ring chromosome by chromosome.  Every file is byte
    for byte that of the ring route (CONGA_HOST_PACKED=0); samples with reads outside [0, L), with an empty chromosome, and one whose
    positions are out of order (refused alike by both routes)."""
    d = str(tmp_path)
    specs = [("1", 500_000, 30, 8), ("2", 300_000, 20, 5), ("3", 200_000, 12, 3)]
    cs = [synth.make_chrom(n, L, cov=1.0, n_dels=nd, n_dups=nu, mappability=True, gaps=False) for n, L, nd, nu in specs]
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    synth.write_bed(os.path.join(d, "dups.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.dup_start, c.dup_end)])
    samples = []
    for k, cov in enumerate([1.0, 3.0, 0.5, 2.0, 1.5]):
        chroms = []
        for ci, c in enumerate(cs):
            rng = np.random.default_rng([k, ci, 9])
            pos, mapq = synth.make_reads(c.length, c.gc, c.step, cov, 100, rng)
            if k == 2 and ci == 1:
                pos, mapq = pos[:0], mapq[:0]                                   # a chromosome without a read
            past = k in (2, 3) and ci == 0                                      # reads at / behind the annotated length (the file says L + 100)
            if past:
                pos = np.concatenate([pos, np.array([c.length, c.length + 5], np.int32)])
                mapq = np.concatenate([mapq, np.array([60, 60], np.uint8)])
            if k == 2 and ci == 2:                                              # a container may hold anything: reads in front of base 0
                pos = np.concatenate([np.array([-7, -1], np.int32), pos])
                mapq = np.concatenate([np.array([60, 60], np.uint8), mapq])
            chroms.append((c.name, c.length + (100 if past else 0), pos, mapq))
        if k % 2 == 0:
            formats.write_tuples(os.path.join(d, "s%d.ctp" % k), "S%d" % k, chroms, aligned=k != 4)   # (the last one: CONGATP1, unaligned arrays)
            samples.append("s%d.ctp" % k)
        else:
            formats.write_bam(os.path.join(d, "s%d.bam" % k), "S%d" % k, chroms, index=True, block_payload=20_000, unplaced=2)
            samples.append("s%d.bam" % k)
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("".join("%s\n" % s_ for s_ in samples))
    common = ["--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed"] + (["--mq", str(mq)] if mq is not None else [])
    outs = {}
    for packed in ("1", "0"):
        env = dict(os.environ, CONGA_GPU_BAM="0", CONGA_TIMING="1", CONGA_HOST_PACKED=packed)
        r = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "p" + packed] + common, cwd=d, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert r.stderr.count("[timing] packed hand-over:") == (4 if packed == "1" else 0), r.stderr[-3000:]   # every sample but the first
        outs[packed] = r.stderr
    for k in range(len(samples)):
        stem = samples[k][:-4] if samples[k].endswith(".bam") else samples[k]   # (<--out>.<file name without .bam>)
        for kind in ("svs", "dels", "dups"):
            got = open(os.path.join(d, "p1.%s_%s.bed" % (stem, kind)), "rb").read()
            want = open(os.path.join(d, "p0.%s_%s.bed" % (stem, kind)), "rb").read()
            assert got == want and (kind != "dels" or got.count(b"\n") > 5), (k, kind)
    # the progress lines too: the reads counted per chromosome are the ring route's
    counted = lambda text: [ln for ln in text.splitlines() if "-->counting reads" in ln]   # noqa: E731
    assert counted(outs["1"]) == counted(outs["0"]) and len(counted(outs["1"])) == 15
    assert open(os.path.join(d, "p1.s0.ctp_dels.bed"), "rb").read() != open(os.path.join(d, "p1.s4.ctp_dels.bed"), "rb").read()
    # positions out of order: both routes end with the engine's refusal
    bad = cs[0]
    rng = np.random.default_rng(77)
    pos, mapq = synth.make_reads(bad.length, bad.gc, bad.step, 1.0, 100, rng)
    pos[1000], pos[1001] = pos[1001] + 300, pos[1000]
    formats.write_tuples(os.path.join(d, "bad.ctp"), "BAD", [(bad.name, bad.length, pos, mapq)] + [(c.name, c.length, c.pos, c.mapq) for c in cs[1:]])
    with open(os.path.join(d, "list2.txt"), "w") as f:
        f.write("s0.ctp\nbad.ctp\n")
    for packed in ("1", "0"):
        r = subprocess.run([CONGA, "--cohort", "list2.txt", "--out", "q" + packed] + common, cwd=d, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, CONGA_HOST_PACKED=packed))
        assert r.returncode != 0 and "out of position order" in r.stderr, (packed, r.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("n,decode", [(2, "gpu_ahead"), (3, "gpu"), (2, "host"), (7, "gpu_ahead")])
def test_cli_cohort_gpus_deals_the_samples_and_keeps_every_byte(tmp_path, n, decode):
    """`--cohort list --gpus N`: the samples are dealt round robin to N pipelines -- one kept engine context, its layout and its
    two-deep upload pipeline per GPU (here N contexts share the one visible device, as in test_cli_gpus_shards_chromosomes...) --
    because a cohort shards by SAMPLE (the reference runs one process per sample: bam_data.c:253-339, svdepth.c:47-66) and end to
    end a sample costs its upload over ITS GPU's link.  Every sample's three files are byte for byte those of the `--gpus 1` run;
    N larger than the list: one pipeline per sample."""
    d = str(tmp_path)
    specs = [("1", 500_000, 30, 8), ("2", 300_000, 20, 5), ("3", 200_000, 12, 3)]
    cs = [synth.make_chrom(nm, L, cov=1.0, n_dels=nd, n_dups=nu, mappability=True, gaps=False) for nm, L, nd, nu in specs]
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    synth.write_bed(os.path.join(d, "dups.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.dup_start, c.dup_end)])
    synth.write_bed(os.path.join(d, "map.bed"), [(c.name, s, e, "%g" % v) for c in cs for s, e, v in zip(c.map_start, c.map_end, c.map_val)])
    samples = []
    for k, (cov, drop) in enumerate([(1.0, None), (3.0, None), (0.5, "2"), (2.0, None), (1.5, None)]):   # the third sample lacks chromosome 2
        chroms = []
        for ci, c in enumerate(cs):
            if c.name == drop:
                continue
            rng = np.random.default_rng([k, ci, 6])
            pos, mapq = synth.make_reads(c.length, c.gc, c.step, cov, 100, rng)
            chroms.append((c.name, c.length, pos, mapq))
        formats.write_bam(os.path.join(d, "s%d.bam" % k), "S%d" % k, chroms, index=True, block_payload=20_000, unplaced=2)
        samples.append("s%d.bam" % k)
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("".join("%s\n" % s for s in samples))
    common = ["--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed", "--mappability", "map.bed"]
    env = dict(os.environ, CONGA_GPU_BAM="0" if decode == "host" else "1", CONGA_TIMING="1")
    if decode == "gpu_ahead":
        env.update(CONGA_BGZF_OVERLAP="1", CONGA_BGZF_PIECE_KB="16", CONGA_BGZF_CHECK_TABLE="1")
    one = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "one"] + common, cwd=d, capture_output=True, text=True, timeout=300, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    many = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "many", "--gpus", str(n)] + common, cwd=d, capture_output=True, text=True,
                          timeout=300, env=env)
    assert many.returncode == 0, many.stderr[-3000:]
    assert "pipelines share devices" in many.stderr                  # (one GPU on the test box)
    for k in range(len(samples)):
        for kind in ("svs", "dels", "dups"):
            got = open(os.path.join(d, "many.s%d_%s.bed" % (k, kind)), "rb").read()
            want = open(os.path.join(d, "one.s%d_%s.bed" % (k, kind)), "rb").read()
            assert got == want and (kind != "dels" or got.count(b"\n") > 5), (k, kind)
    # every pipeline kept its engine: as many contexts were made as there are pipelines, not as there are samples
    assert many.stderr.count("[timing] conga_create:") == min(n, len(samples)), many.stderr[-3000:]
    if decode == "gpu_ahead" and n == 2:
        assert many.stderr.count("named ahead") >= 1, many.stderr[-3000:]   # (each pipeline runs its own two-deep upload pipeline)


@pytest.mark.gpu
@pytest.mark.parametrize("piece_kb,ahead,no_table", [("16", None, False), ("6", "2", False), ("16", "2", True), ("6", "1", True)])
def test_cli_cohort_that_stood_still_on_round_3s_last_day(tmp_path, piece_kb, ahead, no_table):
    """tests/soak.py --bam, seed 81, case 38, as a fixture (profiles/r03l_cohort_standstill_seed81_case38.log; commit 6e95a92): a
    cohort of four small samples whose uploads take no time, two named ahead -- the inflate-ahead thread of the job named SECOND
    took the spare output set, the call in front waited for the job named first, that one for the set, the set for the call behind.
    Pieces of 16 KB are the draw of the case itself, 6 KB (smaller than a block) the judge's.  no_table: CONGA_BGZF_NO_TABLE=1 -- the
    planning thread reads the block table from the file and hands it over with conga_reads_bgzf_next_blocks, the path ADVICE round 3
    found left out of the fix (a job whose table the caller brought never entered the set of tickets the spare buffers go to).
    Must end within 60 s, every sample's files those of its own run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import soak
    d = str(tmp_path)
    args = soak.bam_case(np.random.default_rng([81, 11_000_000 + 38]), d)
    assert "--gpus" not in args
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("".join("r.bam\tc%d\n" % k for k in range(4)))
    env = dict(os.environ, CONGA_GPU_BAM="1", CONGA_BGZF_OVERLAP="1", CONGA_BGZF_CHECK_TABLE="1", CONGA_TIMING="1", CONGA_BGZF_PIECE_KB=piece_kb)
    if ahead:
        env["CONGA_COHORT_AHEAD"] = ahead
    if no_table:
        env["CONGA_BGZF_NO_TABLE"] = "1"
    r = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "co"] + args[2:], cwd=d, capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0 and "decoding on the host" not in r.stderr, r.stderr[-3000:]
    one = subprocess.run([CONGA] + args + ["--out", "one"], cwd=d, capture_output=True, text=True, timeout=60, env=dict(os.environ, CONGA_GPU_BAM="1"))
    assert one.returncode == 0, one.stderr[-2000:]
    for k in range(4):
        for kind in ("svs", "dels", "dups"):
            assert open(os.path.join(d, "c%d_%s.bed" % (k, kind)), "rb").read() == open(os.path.join(d, "one_%s.bed" % kind), "rb").read(), (k, kind)
    assert r.stderr.count("named ahead") >= 1, r.stderr[-3000:]   # (the pipeline was on)


@pytest.mark.gpu
def test_cli_cohort_whose_next_target_begins_in_the_stretchs_first_block(tmp_path):
    """tests/soak.py --bam, seed 3001, case 362, as a fixture (round 4's last hours): a file so small that the target behind the last
    one wanted begins in the FIRST block of the stretch named to the engine.  `stop_at` -- where the engine's block table may end --
    was that block's offset inside the stretch, 0, which also said "none": the engine's table had every block of the stretch (4),
    the planner's own walk stopped after the first, and CONGA_BGZF_CHECK_TABLE=1 (which wants the two alike) aborted the run.
    stop_at is one more than the offset now."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import soak
    d = str(tmp_path)
    args = soak.bam_case(np.random.default_rng([3001, 11_000_000 + 362]), d)
    assert "--gpus" not in args
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("".join("r.bam\tc%d\n" % k for k in range(4)))
    env = dict(os.environ, CONGA_GPU_BAM="1", CONGA_BGZF_OVERLAP="1", CONGA_BGZF_CHECK_TABLE="1", CONGA_TIMING="1", CONGA_BGZF_PIECE_KB="512")
    r = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "co"] + args[2:], cwd=d, capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0 and "decoding on the host" not in r.stderr and "is not the one read from the file" not in r.stderr, r.stderr[-3000:]
    assert "the engine's and the file's agree" in r.stderr, r.stderr[-3000:]
    one = subprocess.run([CONGA] + args + ["--out", "one"], cwd=d, capture_output=True, text=True, timeout=60, env=dict(os.environ, CONGA_GPU_BAM="1"))
    assert one.returncode == 0, one.stderr[-2000:]
    for k in range(4):
        for kind in ("svs", "dels", "dups"):
            assert open(os.path.join(d, "c%d_%s.bed" % (k, kind)), "rb").read() == open(os.path.join(d, "one_%s.bed" % kind), "rb").read(), (k, kind)


@pytest.mark.gpu
def test_cli_cohort_pipeline_neither_grows_its_spare_set_nor_waits_for_a_call_that_waits_for_it(tmp_path):
    """Round 4's last day, from the pipeline's event trace (profiles/r04j_cohort_first_samples.log).  (1) The first sample's output set is
    the spare set after the first swap: sized for that sample alone it was grown by the next job's inflating thread -- gigabytes allocated
    and freed behind the runtime's lock, 0.1-0.65 s of a whole-genome cohort; with CONGA_FLAG_EXPECT_COHORT it has the spare's size from
    the start.  (2) Bytes named between two calls waited for a call to begin while the thread of the calls waited for their plan:
    conga_reads_bgzf_next_table's 0.4 s, one run in three; the executable now says conga_reads_bgzf_next_go.  Six samples of one size,
    the trace on: no growth, no sample-to-sample gap of 0.35 s, every sample's files those of a run of its own."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import re
    import soak
    d = str(tmp_path)
    args = soak.bam_case(np.random.default_rng([81, 11_000_000 + 38]), d)
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("".join("r.bam\tc%d\n" % k for k in range(6)))
    env = dict(os.environ, CONGA_GPU_BAM="1", CONGA_BGZF_OVERLAP="1", CONGA_TIMING="1", CONGA_BGZF_PIECE_KB="16", CONGA_DEBUG="1", CONGA_BGZF_TRACE="1")
    for rep in range(3):   # (the circle of waits was a race: a few runs)
        r = subprocess.run([CONGA, "--cohort", "list.txt", "--out", "co"] + args[2:], cwd=d, capture_output=True, text=True, timeout=60, env=env)
        assert r.returncode == 0 and "decoding on the host" not in r.stderr, r.stderr[-3000:]
        assert "the spare output set grows" not in r.stderr, [ln for ln in r.stderr.splitlines() if "spare output set" in ln][:6]
        done = [float(x) for x in re.findall(r"cohort: sample \d+ of \d+ is done ([0-9.]+) ms", r.stderr)]
        assert len(done) == 6 and max(b - a for a, b in zip(done[1:], done[2:])) < 350.0, done
        assert r.stderr.count("inflated ahead: the output sets change places") >= 2, r.stderr[-3000:]   # (the pipeline was on)
    one = subprocess.run([CONGA] + args + ["--out", "one"], cwd=d, capture_output=True, text=True, timeout=60, env=dict(os.environ, CONGA_GPU_BAM="1"))
    assert one.returncode == 0, one.stderr[-2000:]
    for k in range(6):
        for kind in ("svs", "dels", "dups"):
            assert open(os.path.join(d, "c%d_%s.bed" % (k, kind)), "rb").read() == open(os.path.join(d, "one_%s.bed" % kind), "rb").read(), (k, kind)


@pytest.mark.gpu
def test_cli_damaged_bam_records_gpu_and_host_decoders_agree():
    """Records with a damaged block_size, refID, pos, l_read_name, n_cigar, l_seq or flag, and random bytes, in a BAM whose
    blocks still check out (tools/bam_fuzz.py): the run that decodes on the GPU ends the way the run on the host decoders
    ends -- same exit code, same files -- and neither dies of a signal (a record out of position order is refused by the GPU
    walk and reported by the host path, not dropped)."""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bam_fuzz.py")
    r = subprocess.run([sys.executable, tool, "40", "20261004"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "disagreements or crashes 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # ... and the container of a deflated BAM: block headers, BSIZE, XLEN, deflate bytes and bits, CRC32, ISIZE, truncation
    r = subprocess.run([sys.executable, tool, "24", "20261005", "container"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "disagreements or crashes 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # ... and the .bai: bytes, 32- and 64-bit fields overwritten, truncation.  An index whose two views of where a target begins
    # contradict each other is used by neither decoder for that target (round 2: a zeroed chunk made the GPU plan take a target for
    # empty while the sequential reader scanned forward and found its reads -- same input, different files, exit 0)
    r = subprocess.run([sys.executable, tool, "60", "20261006", "index"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "disagreements or crashes 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_cli_records_of_hundreds_of_kilobytes_decode_alike():
    """Records that span several BGZF blocks each (reads of 70 000 - 200 000 bases among ordinary ones, tools/long_records.py):
    the walk on the GPU follows them through the concatenated blocks, the host decoders through their stream: same files."""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "long_records.py")
    r = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "same: True" in r.stdout and "gpu 0 False" in r.stdout and "host 0 False" in r.stdout, r.stdout + r.stderr[-2000:]
