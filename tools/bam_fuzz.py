#!/usr/bin/env python3
"""Damaged BAM records through the `conga` executable, decoded on the GPU and by the host decoders: what each makes of them.
The BAM is written with stored deflate blocks, so bytes of records can be changed in place (the block's CRC32 is redone) with
the index still pointing at the right places."""
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import formats, synth  # noqa: E402

CONGA = os.path.join(ROOT, "conga_amd", "host", "conga")


def blocks_of(raw):
    at, out = 0, []
    while at + 18 <= len(raw):
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        out.append((at, bsize))
        at += bsize
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    container = len(sys.argv) > 3 and sys.argv[3] == "container"   # damage the BGZF container of a deflated BAM instead of records
    index = len(sys.argv) > 3 and sys.argv[3] == "index"           # damage the .bai instead
    d = tempfile.mkdtemp(prefix="conga_bamfuzz_")
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
    cs = [synth.make_chrom(n, L, cov=2.0, n_dels=nd, gaps=True) for n, L, nd in (("1", 500_000, 20), ("2", 300_000, 12))]
    formats.write_bam(os.path.join(d, "good.bam"), "S", [(c.name, c.length, c.pos, c.mapq) for c in cs], index=True, block_payload=30000,
                      level=6 if container else 0, unplaced=3)
    formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
    good = bytearray(open(os.path.join(d, "good.bam"), "rb").read())
    bai = open(os.path.join(d, "good.bam.bai"), "rb").read()
    blks = blocks_of(good)
    base = ["--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed"]

    def run(bam, out, gpu):
        r = subprocess.run([CONGA, "-i", bam, "--out", out] + base, cwd=d, capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, CONGA_GPU_BAM="1" if gpu else "0"))
        files = None
        if r.returncode == 0:
            files = [open(os.path.join(d, "%s_%s.bed" % (out, k)), "rb").read() for k in ("svs", "dels")]
        return r, files

    kinds = ["block_size", "refid", "pos", "l_read_name", "n_cigar", "l_seq", "flag", "random byte", "random run", "block_size small"]
    summary = {}
    bad = 0
    ckinds = ["header byte", "bsize", "deflate byte", "crc", "isize", "xlen", "truncate", "deflate bit"]
    for case in range(n_cases):
        raw = bytearray(good)
        if index:
            kind = ["byte", "int32", "int64", "truncate"][case % 4]
            ix = bytearray(bai)
            if kind == "byte":
                ix[int(rng.integers(0, len(ix)))] = int(rng.integers(0, 256))
            elif kind == "int32":
                struct.pack_into("<i", ix, int(rng.integers(0, len(ix) - 4)), int(rng.choice([-1, 0, 1, 1 << 20, 0x7fffffff, -5])))
            elif kind == "int64":
                struct.pack_into("<Q", ix, int(rng.integers(0, len(ix) - 8)), int(rng.choice([0, 1 << 16, (1 << 63) - 1, 12345 << 16, len(good) << 16])))
            else:
                del ix[int(rng.integers(0, len(ix))):]
            bam = "c%d.bam" % case
            open(os.path.join(d, bam), "wb").write(bytes(raw))
            open(os.path.join(d, bam + ".bai"), "wb").write(bytes(ix))
            rg, fg = run(bam, "g%d" % case, True)
            rh, fh = run(bam, "h%d" % case, False)
            same = rg.returncode == rh.returncode and fg == fh
            key = (kind, rg.returncode, rh.returncode, same, "decoding on the host" in rg.stderr)
            summary[key] = summary.get(key, 0) + 1
            if rg.returncode < 0 or rh.returncode < 0 or not same:
                bad += 1
                print("CASE", case, kind, "gpu rc", rg.returncode, "host rc", rh.returncode, "same files", fg == fh)
                print("  gpu:", rg.stderr.strip().splitlines()[-2:])
                print("  host:", rh.stderr.strip().splitlines()[-2:])
            continue
        if container:
            kind = ckinds[case % len(ckinds)]
            bi = int(rng.integers(0, len(blks) - 1))
            at, bsize = blks[bi]
            if kind == "header byte":
                raw[at + int(rng.integers(0, 18))] = int(rng.integers(0, 256))
            elif kind == "bsize":
                struct.pack_into("<H", raw, at + 16, int(rng.integers(0, 65536)))
            elif kind == "deflate byte":
                raw[int(rng.integers(at + 18, at + bsize - 8))] = int(rng.integers(0, 256))
            elif kind == "deflate bit":
                raw[int(rng.integers(at + 18, at + bsize - 8))] ^= 1 << int(rng.integers(0, 8))
            elif kind == "crc":
                raw[at + bsize - 8 + int(rng.integers(0, 4))] ^= 0xFF
            elif kind == "isize":
                struct.pack_into("<I", raw, at + bsize - 4, int(rng.choice([0, 1, 65536, 70000, 0xFFFFFFFF, 29999])))
            elif kind == "xlen":
                struct.pack_into("<H", raw, at + 10, int(rng.choice([0, 4, 8, 100, 65535])))
            else:
                del raw[int(rng.integers(at, len(raw))):]
            bam = "c%d.bam" % case
            open(os.path.join(d, bam), "wb").write(bytes(raw))
            open(os.path.join(d, bam + ".bai"), "wb").write(bai)
            rg, fg = run(bam, "g%d" % case, True)
            rh, fh = run(bam, "h%d" % case, False)
            same = rg.returncode == rh.returncode and fg == fh
            key = (kind, rg.returncode, rh.returncode, same, "decoding on the host" in rg.stderr)
            summary[key] = summary.get(key, 0) + 1
            if rg.returncode < 0 or rh.returncode < 0 or not same:
                bad += 1
                print("CASE", case, kind, "gpu rc", rg.returncode, "host rc", rh.returncode, "same files", fg == fh)
                print("  gpu:", rg.stderr.strip().splitlines()[-2:])
                print("  host:", rh.stderr.strip().splitlines()[-2:])
            continue
        kind = kinds[case % len(kinds)]
        # a data block that holds records (not the first: the header), stored deflate: 5 bytes of block header, then the payload
        bi = int(rng.integers(1, len(blks) - 2))
        at, bsize = blks[bi]
        pay0, pay1 = at + 18 + 5, at + bsize - 8
        # find a record start inside this block by walking from the file's first record: simpler -- scan for the constant CIGAR + name pattern
        payload = bytes(raw[pay0:pay1])
        hits = [i for i in range(0, len(payload) - 40) if payload[i + 12:i + 13] != b"" and payload[i + 36:i + 37] == b"r"]  # name starts with 'r' at +36
        rec = hits[int(rng.integers(0, len(hits)))] if hits else 0
        p = pay0 + rec
        if kind == "block_size":
            struct.pack_into("<i", raw, p, int(rng.choice([-5, 0, 7, 1 << 20, 0x7fffffff, 33])))
        elif kind == "block_size small":
            struct.pack_into("<i", raw, p, int(rng.integers(1, 40)))
        elif kind == "refid":
            struct.pack_into("<i", raw, p + 4, int(rng.choice([-1, 1, 0, 5, -7, 1 << 30])))
        elif kind == "pos":
            struct.pack_into("<i", raw, p + 8, int(rng.choice([-1, -100, 1 << 30, 0, 499_999, 700_000])))
        elif kind == "l_read_name":
            raw[p + 12] = int(rng.integers(0, 256))
        elif kind == "n_cigar":
            struct.pack_into("<H", raw, p + 16, int(rng.choice([0, 2, 500, 65535])))
        elif kind == "l_seq":
            struct.pack_into("<i", raw, p + 20, int(rng.choice([-1, 0, 5, 1 << 20, 0x7fffffff])))
        elif kind == "flag":
            struct.pack_into("<H", raw, p + 18, int(rng.integers(0, 65536)))
        elif kind == "random byte":
            raw[int(rng.integers(pay0, pay1))] = int(rng.integers(0, 256))
        else:
            a = int(rng.integers(pay0, pay1 - 64))
            raw[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        struct.pack_into("<I", raw, at + bsize - 8, zlib.crc32(bytes(raw[pay0:pay1])) & 0xFFFFFFFF)
        bam = "c%d.bam" % case
        open(os.path.join(d, bam), "wb").write(bytes(raw))
        open(os.path.join(d, bam + ".bai"), "wb").write(bai)
        rg, fg = run(bam, "g%d" % case, True)
        rh, fh = run(bam, "h%d" % case, False)
        same = rg.returncode == rh.returncode and fg == fh
        on_host = "decoding on the host" in rg.stderr
        key = (kind, rg.returncode, rh.returncode, same, on_host)
        summary[key] = summary.get(key, 0) + 1
        if rg.returncode < 0 or rh.returncode < 0 or not same:
            bad += 1
            print("CASE", case, kind, "gpu rc", rg.returncode, "host rc", rh.returncode, "same files", fg == fh)
            print("  gpu:", rg.stderr.strip().splitlines()[-2:])
            print("  host:", rh.stderr.strip().splitlines()[-2:])
    for k in sorted(summary):
        print(k, summary[k])
    print("cases", n_cases, "disagreements or crashes", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
