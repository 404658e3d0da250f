"""A small BAM + .bai assembled by hand from the SAM/BAM specification (SAMv1 sections 4.1, 4.2, 5.1.1, 5.2), byte by byte, WITHOUT the
repository's writers (conga_amd/formats.py, tools/bamwrite.cpp): round 3's VERDICT noted that the BAM record and index layer had only
ever been checked against files the repo wrote itself.  The reference takes its records from htslib (bam_data.c:199-201,253-259,293:
sam_itr_queryi(idx, tid, 0, L) + sam_itr_next); what an htslib-written file may hold and the repo's writers never emit is here on
purpose:

  * CIGARs of several operations (M I D N S H P = X), hard clips (l_seq shorter than the read), a record with no CIGAR at all
  * optional fields of every type behind the qualities: A c C s S i I f Z H and B arrays
  * read names of 1 .. 60 characters; sequences of odd and even length; missing qualities (0xFF)
  * placed-unmapped reads (flag 0x4 with refID / pos of the mate): the iterator yields them, count_reads_bam counts them
  * secondary, supplementary, duplicate, QC-fail, paired flags; mate fields and template lengths of every sign
  * BGZF blocks cut at ARBITRARY bytes -- records straddle block borders, one record lies across three blocks, one stretch is
    blocks of 64 bytes --, an EMPTY block (ISIZE 0) in the middle of the file, stored (BTYPE 00) blocks among deflated ones, the
    28-byte EOF marker at the end
  * a reference in the header that the annotation does not know, one with no reads at all, unplaced reads (refID -1) at the end
  * the .bai with the metadata pseudo-bin 37450 (offsets of the reference's first and last byte, mapped / unmapped counts) in
    every reference that has reads, bins of all levels, the 16 kb linear index, n_no_coor

make(directory) writes hand.bam, hand.bam.bai, ref.fa and returns what a reader must find: per reference the (pos, mapq, flag,
l_seq) of its records in file order."""
import os
import struct
import zlib

import numpy as np

CIGAR_OPS = "MIDNSHP=X"
SEQ_CODES = "=ACMGRSVTWYHKDBN"


def reg2bin(beg, end):
    """SAMv1 section 5.3 (the C function given there)"""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def bgzf_block(payload, level):
    """One BGZF block (section 4.1): gzip member with the BC extra field, raw deflate, CRC32, ISIZE"""
    if level == 0 and payload:   # a stored deflate block, written out by hand
        data = b"\x01" + struct.pack("<HH", len(payload), len(payload) ^ 0xFFFF) + payload
    else:
        co = zlib.compressobj(level if level else 6, zlib.DEFLATED, -15)
        data = co.compress(payload) + co.flush()
    bsize = 18 + len(data) + 8
    assert bsize <= 65536
    head = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, bsize - 1)
    return head + data + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


EOF_MARKER = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def record(ref_id, pos, mapq, flag, name, cigar, seq, qual, next_ref=-1, next_pos=-1, tlen=0, tags=b""):
    """One alignment record (section 4.2); cigar: [(length, op letter)], seq: str of SEQ_CODES letters, qual: bytes or None"""
    ops = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for n, op in cigar)
    ref_len = sum(n for n, op in cigar if op in "MDN=X")
    end = pos + (ref_len if ref_len > 0 else 1)
    codes = [SEQ_CODES.index(c) for c in seq] + [0]
    packed = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(seq), 2))
    q = bytes([0xFF]) * len(seq) if qual is None else bytes(qual)
    assert len(q) == len(seq)
    nm = name.encode() + b"\x00"
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(nm), mapq, reg2bin(pos, end) if pos >= 0 else 4680, len(cigar), flag, len(seq),
                       next_ref, next_pos, tlen) + nm + ops + packed + q + tags
    return struct.pack("<i", len(body)) + body, end


def some_tags(rng, k):
    """optional fields of every type (section 4.2.4)"""
    t = b"NM" + b"C" + bytes([k % 7])
    if k % 2 == 0:
        t += b"RG" + b"Z" + b"rg%d" % (k % 3) + b"\x00"
    if k % 3 == 0:
        t += b"MD" + b"Z" + b"%d^AC%d" % (k % 50, 50 - k % 50) + b"\x00"
    if k % 5 == 0:
        t += b"XS" + b"A" + (b"+" if k % 10 else b"-")
    if k % 7 == 0:
        t += b"AS" + b"i" + struct.pack("<i", -k)
    if k % 11 == 0:
        t += b"Xc" + b"c" + struct.pack("<b", -5) + b"Xs" + b"s" + struct.pack("<h", -300) + b"XS" + b"S" + struct.pack("<H", 60000) \
            + b"XI" + b"I" + struct.pack("<I", 4_000_000_000) + b"Xf" + b"f" + struct.pack("<f", 1.5)
    if k % 13 == 0:
        t += b"XH" + b"H" + b"1AE301" + b"\x00"
    if k % 17 == 0:
        arr = rng.integers(-100, 100, int(rng.integers(0, 40)))
        t += b"ZB" + b"B" + b"c" + struct.pack("<i", len(arr)) + arr.astype(np.int8).tobytes()
    if k % 19 == 0:
        arr = rng.integers(0, 60000, int(rng.integers(1, 9)))
        t += b"ZS" + b"B" + b"S" + struct.pack("<i", len(arr)) + arr.astype("<u2").tobytes()
    return t


def make(d, seed=20261005):
    rng = np.random.default_rng(seed)
    refs = [("1", 300_000), ("GL000207.1", 4_262), ("2", 140_000), ("empty", 50_000)]   # (the second one is not in the annotation)
    genome = {name: "".join(rng.choice(list("ACGT"), L)) for name, L in refs}
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs) + \
        "@RG\tID:rg0\tSM:HAND\tPL:ILLUMINA\n@RG\tID:rg1\tSM:HAND\n@PG\tID:by_hand\tPN:tests/bam_by_hand.py\n@CO\ta comment line\n"
    tb = text.encode()
    stream = bytearray(b"BAM\x01" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(refs)))
    for name, L in refs:
        stream += struct.pack("<i", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<i", L)
    rec_at = []                                  # (ref, stream offset of the record, offset of its end, pos, alignment end, flag)
    expect = {name: [] for name, _ in refs}      # per reference: (pos, mapq, flag, l_seq) in file order
    k = 0
    cigars = [lambda l: [(l, "M")], lambda l: [(30, "M"), (2, "I"), (l - 32, "M")], lambda l: [(20, "S"), (l - 20, "M")],
              lambda l: [(l // 2, "M"), (1000, "N"), (l - l // 2, "M")], lambda l: [(45, "M"), (5, "D"), (l - 45, "M")],
              lambda l: [(10, "H"), (l, "M"), (7, "H")], lambda l: [(5, "S"), (20, "="), (1, "X"), (l - 36, "M"), (10, "S")],
              lambda l: [(40, "M"), (3, "P"), (1, "I"), (l - 41, "M")]]
    three_block_record = None
    for ref_id, (name, L) in enumerate(refs):
        if name == "empty":
            continue
        n = {"1": 5200, "GL000207.1": 40, "2": 2100}[name]
        pos = np.sort(rng.integers(0, L - 200, n))
        pos[:3] = 0                              # reads on the reference's first base
        if name == "2":                          # a pile on a 16 kb window's border, and one on a 128 kb bin's border
            pos[100:160] = 16384 * 3
            pos[160:170] = 131072 - 1
            pos = np.sort(pos)
        for p in pos.tolist():
            l = int(rng.choice([100, 100, 100, 101, 76, 151, 59, 60, 61, 35]))
            flag = int(rng.choice([0, 0, 0, 16, 1 + 2 + 64 + 32, 1 + 128 + 16, 256, 1024, 2048, 512, 4 + 1 + 64, 4 + 1 + 128 + 32]))
            unmapped = bool(flag & 4)
            cig = [] if unmapped or k % 97 == 0 else (cigars[k % len(cigars)](l) if l >= 70 else [(l, "M")])
            seq_len = sum(n_ for n_, op in cig if op in "MIS=X") if cig else l
            if p + 1200 > L:
                cig = [(seq_len, "M")] if cig else cig
            s = genome[name][p:p + seq_len].ljust(seq_len, "N")
            if k % 9 == 0:
                s = s[:seq_len // 3] + "N" + s[seq_len // 3 + 1:]
            if k % 23 == 0:
                s = s.replace("A", "R", 1)       # an ambiguity code
            qual = None if k % 31 == 0 else rng.integers(2, 42, seq_len).astype(np.uint8).tobytes()
            mapq = 0 if unmapped else int(rng.choice([60, 60, 60, 0, 17, 40, 255]))
            rname = ("r%d" % k) if k % 41 else "a_very_long_read_name_of_some_sixty_characters_%012d" % k
            if k % 53 == 0:
                rname = "q"
            tags = some_tags(rng, k) if k % 4 else b""
            if three_block_record is None and name == "1" and k > 2000:
                tags += b"ZZ" + b"Z" + b"x" * 900 + b"\x00"   # a long record for the stretch of tiny blocks below
                three_block_record = len(stream)
            b, end = record(ref_id, p, mapq, flag, rname, cig, s, qual, next_ref=ref_id if flag & 1 else -1,
                            next_pos=max(0, p + int(rng.integers(-400, 400))) if flag & 1 else -1, tlen=int(rng.integers(-500, 500)) if flag & 1 else 0,
                            tags=tags)
            rec_at.append((ref_id, len(stream), len(stream) + len(b), p, end, flag))
            stream += b
            expect[name].append((p, mapq, flag, seq_len))
            k += 1
    n_no_coor = 7
    for j in range(n_no_coor):                   # unplaced reads close the file (refID -1)
        b, _ = record(-1, -1, 0, 4, "u%d" % j, [], "ACGTN" * 10, None)
        stream += b
    stream = bytes(stream)

    # ---- BGZF: block borders at arbitrary bytes
    cuts = [0]
    at = 0
    while at < len(stream):
        if three_block_record is not None and three_block_record - 200 <= at < three_block_record + 1400:
            step = 64 if at < three_block_record + 300 else 400   # tiny blocks: the long record lies across a dozen of them
        else:
            step = int(rng.integers(3_000, 65_000))
            if three_block_record is not None and at < three_block_record - 200 < at + step:
                step = three_block_record - 200 - at   # (the block in front of the stretch ends where the stretch begins)
        at = min(len(stream), at + step)
        cuts.append(at)
    blocks, file_off, off = [], [], 0
    empty_after = len(cuts) // 2
    for i in range(len(cuts) - 1):
        blk = bgzf_block(stream[cuts[i]:cuts[i + 1]], 0 if i % 5 == 3 else int(rng.choice([1, 6, 9])))
        file_off.append(off)
        blocks.append(blk)
        off += len(blk)
        if i == empty_after:                     # an empty block in the middle of the file
            e = bgzf_block(b"", 6)
            assert struct.unpack_from("<I", e, len(e) - 4)[0] == 0
            blocks.append(e)
            off += len(e)
    blocks.append(EOF_MARKER)
    with open(os.path.join(d, "hand.bam"), "wb") as f:
        f.write(b"".join(blocks))

    def voffset(u):
        i = int(np.searchsorted(cuts, u, side="right")) - 1
        if i >= len(file_off):                   # the very end of the stream: behind the last data block
            return (off << 16)
        return (file_off[i] << 16) | (u - cuts[i])

    # ---- .bai (section 5.2)
    bai = bytearray(b"BAI\x01" + struct.pack("<i", len(refs)))
    for ref_id, (name, L) in enumerate(refs):
        mine = [r for r in rec_at if r[0] == ref_id]
        if not mine:
            bai += struct.pack("<i", 0) + struct.pack("<i", 0)
            continue
        bins = {}
        for _r, u0, u1, p, end, _fl in mine:
            bins.setdefault(reg2bin(p, end), []).append((voffset(u0), voffset(u1)))
        bai += struct.pack("<i", len(bins) + 1)
        for b in sorted(bins):
            chunks = sorted(bins[b])
            merged = [list(chunks[0])]
            for c0, c1 in chunks[1:]:            # chunks that touch become one
                if c0 <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], c1)
                else:
                    merged.append([c0, c1])
            bai += struct.pack("<Ii", b, len(merged)) + b"".join(struct.pack("<QQ", c0, c1) for c0, c1 in merged)
        n_unmapped = sum(1 for r in mine if r[5] & 4)
        bai += struct.pack("<Ii", 37450, 2) + struct.pack("<QQ", voffset(mine[0][1]), voffset(mine[-1][2])) \
            + struct.pack("<QQ", len(mine) - n_unmapped, n_unmapped)
        n_intv = (max(r[4] for r in mine) - 1 >> 14) + 1
        lin = [0] * n_intv
        for _r, u0, _u1, p, end, _fl in mine:
            for w in range(p >> 14, (end - 1 >> 14) + 1):
                if lin[w] == 0 or voffset(u0) < lin[w]:
                    lin[w] = voffset(u0)
        for w in range(1, n_intv):               # (htslib fills windows without alignments with their predecessor's offset)
            if lin[w] == 0:
                lin[w] = lin[w - 1]
        bai += struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", v) for v in lin)
    bai += struct.pack("<Q", n_no_coor)
    with open(os.path.join(d, "hand.bam.bai"), "wb") as f:
        f.write(bytes(bai))
    with open(os.path.join(d, "ref.fa"), "w") as f:
        for name, _L in refs:
            f.write(">%s some description\n" % name)
            s = genome[name]
            f.write("\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
    return dict(refs=refs, genome=genome, expect={n: np.array(v, np.int64).reshape(-1, 4) for n, v in expect.items()},
                three_block_record_at=three_block_record, n_blocks=len(blocks), n_records=k)
