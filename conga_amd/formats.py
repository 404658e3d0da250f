"""Binary containers the `conga` command line reads, and their Python writers/readers.

Neither htslib nor SONIC exists in this image (both are empty submodules of the reference), so:
  * `--sonic` takes this implementation's own annotation container (.cga): chromosome table, one rounded
    GC% byte per gc_step-bp window, satellite intervals;
  * `--input` takes a BAM (conga_amd/host/bam_reader.cpp, when built) or a read-tuple container (.ctp) holding
    exactly the fields of bam1_core_t that the path reads (pos, qual; flag and l_qseq for --rp).
All integers are little-endian.
"""
import struct

import numpy as np

ANNOT_MAGIC = b"CONGAAN1"
TUPLE_MAGIC = b"CONGATP2"  # every array starts on a 16-byte boundary (CONGATP1: no padding; still read)


def write_annotation(path, chroms, gc_step=100):
    """chroms: list of (name, length, gc uint8[n_win], sat_start int32[], sat_end int32[])."""
    with open(path, "wb") as f:
        f.write(ANNOT_MAGIC)
        f.write(struct.pack("<ii", gc_step, len(chroms)))
        for name, length, gc, ss, se in chroms:
            nb = name.encode()
            n_win = (length + gc_step - 1) // gc_step
            assert len(gc) == n_win, (name, len(gc), n_win)
            f.write(struct.pack("<H", len(nb)) + nb)
            f.write(struct.pack("<qqq", length, n_win, len(ss)))
        for name, length, gc, ss, se in chroms:
            f.write(np.ascontiguousarray(gc, dtype=np.uint8).tobytes())
            f.write(np.ascontiguousarray(ss, dtype="<i4").tobytes())
            f.write(np.ascontiguousarray(se, dtype="<i4").tobytes())


def read_annotation(path):
    with open(path, "rb") as f:
        assert f.read(8) == ANNOT_MAGIC
        gc_step, n = struct.unpack("<ii", f.read(8))
        table = []
        for _ in range(n):
            (ln,) = struct.unpack("<H", f.read(2))
            name = f.read(ln).decode()
            length, n_win, n_sat = struct.unpack("<qqq", f.read(24))
            table.append((name, length, n_win, n_sat))
        out = []
        for name, length, n_win, n_sat in table:
            gc = np.frombuffer(f.read(n_win), dtype=np.uint8)
            ss = np.frombuffer(f.read(4 * n_sat), dtype="<i4")
            se = np.frombuffer(f.read(4 * n_sat), dtype="<i4")
            out.append((name, length, gc, ss, se))
    return gc_step, out


def write_tuples(path, sample, chroms, aligned=True):
    """chroms: list of (name, length, pos int32[n] sorted, mapq uint8[n][, flag uint16[n], l_qseq int32[n]])."""
    with open(path, "wb") as f:
        f.write(TUPLE_MAGIC if aligned else b"CONGATP1")
        sb = sample.encode()
        f.write(struct.pack("<H", len(sb)) + sb)
        f.write(struct.pack("<i", len(chroms)))
        for c in chroms:
            nb = c[0].encode()
            f.write(struct.pack("<H", len(nb)) + nb)
            f.write(struct.pack("<qqB", c[1], len(c[2]), 1 if len(c) > 4 else 0))
        def put(a, dtype):
            if aligned:
                f.write(b"\0" * (-f.tell() % 16))
            f.write(np.ascontiguousarray(a, dtype=dtype).tobytes())
        for c in chroms:
            put(c[2], "<i4")
            put(c[3], np.uint8)
            if len(c) > 4:
                put(c[4], "<u2")
                put(c[5], "<i4")


# ------------------------------------------------------------------------------------------------
# Minimal BAM writer (BGZF + BAM records per the SAM/BAM specification) for synthetic inputs and tests.
# ------------------------------------------------------------------------------------------------
import zlib

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data, level=6, strategy=0):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    cdata = co.compress(data) + co.flush()
    bsize = len(cdata) + 25  # 12 header + 6 extra + cdata + 8 trailer - 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def write_fasta(path, chroms, width=60, index=True):
    """chroms: list of (name, bytes).  Also writes path + '.fai' (name, length, offset, linebases, linewidth)."""
    fai = []
    with open(path, "wb") as f:
        for name, seq in chroms:
            f.write(b">" + name.encode() + b"\n")
            fai.append((name, len(seq), f.tell(), width, width + 1))
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + b"\n")
    if index:
        with open(path + ".fai", "w") as f:
            for r in fai:
                f.write("\t".join(str(x) for x in r) + "\n")


def write_bam(path, sample, chroms, read_len=100, unplaced=0, block_payload=60000, records=None, index=False, level=6, strategy=0):
    """chroms: list of (name, length, pos int32[n] sorted, mapq uint8[n][, flag uint16[n]]).
    Every record gets a `read_len`M CIGAR, an all-A sequence and quality 30; `unplaced` unmapped records
    (refID -1) are appended at the end, as in a real coordinate-sorted BAM.
    records: optional {chrom name: (l_qseq int32[n], codes uint8[], qual uint8[], off uint64[n])} with one 4-bit
    base code per byte -- then every record carries its own sequence, qualities and an <l>M CIGAR.
    index=True also writes path + '.bai' (bins with one merged chunk each + the 16 kb linear index).
    level / strategy: zlib's, for the BGZF blocks (0: stored blocks; zlib.Z_FIXED: fixed-Huffman blocks ...)."""
    text = "@HD\tVN:1.6\tSO:coordinate\n"
    for c in chroms:
        text += "@SQ\tSN:%s\tLN:%d\n" % (c[0], c[1])
    text += "@RG\tID:rg1\tSM:%s\tPL:ILLUMINA\n" % sample
    tb = text.encode()
    out = bytearray()
    out += b"BAM\x01" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(chroms))
    for c in chroms:
        nb = c[0].encode() + b"\x00"
        out += struct.pack("<i", len(nb)) + nb + struct.pack("<i", c[1])
    l_seq = read_len
    seq = bytes([0x11]) * ((l_seq + 1) // 2)
    qual = bytes([30]) * l_seq
    cigar = struct.pack("<I", (read_len << 4) | 0)
    k = 0
    bins = [dict() for _ in chroms]      # per reference: bin -> [voffset begin, voffset end]
    linear = [dict() for _ in chroms]    # per reference: 16 kb window -> smallest voffset
    with open(path, "wb") as f:
        def flush(final=False):
            nonlocal out
            while len(out) >= block_payload or (final and len(out)):
                f.write(_bgzf_block(bytes(out[:block_payload]), level, strategy))
                out = out[block_payload:]

        def voffset():
            # records are appended to `out`; everything before it is already in closed blocks of block_payload bytes
            return (f.tell() + 0) << 16 | len(out) if len(out) < block_payload else None

        for tid, c in enumerate(chroms):
            pos, mapq = np.asarray(c[2]), np.asarray(c[3])
            flag = np.asarray(c[4]) if len(c) > 4 else np.zeros(len(pos), np.uint16)
            rec = records.get(c[0]) if records else None
            for j, (p, q, fl) in enumerate(zip(pos.tolist(), mapq.tolist(), flag.tolist())):
                name = ("r%d" % k).encode() + b"\x00"
                k += 1
                if rec is not None:
                    l_seq = int(rec[0][j])
                    o = int(rec[3][j])
                    codes = np.zeros((l_seq + 1) // 2 * 2, np.uint8)
                    codes[:l_seq] = rec[1][o:o + l_seq]
                    seq = ((codes[0::2] << 4) | codes[1::2]).tobytes()
                    qual = np.asarray(rec[2][o:o + l_seq], np.uint8).tobytes()
                    cigar = struct.pack("<I", (l_seq << 4) | 0)
                    read_len = l_seq
                b = _reg2bin(p, p + max(read_len, 1))
                body = struct.pack("<iiBBHHHiiii", tid, p, len(name), q, b, 1, fl, l_seq,
                                   -1, -1, 0) + name + cigar + seq + qual
                if len(out) >= block_payload:
                    flush()
                v0 = (f.tell() << 16) | len(out)
                out += struct.pack("<i", len(body)) + body
                if index and p >= 0:
                    # the record may run into the next block; its end offset is only used as a chunk end (an upper bound)
                    v1 = (f.tell() << 16) | min(len(out), 0xFFFF)
                    e = bins[tid].setdefault(b, [v0, v1])
                    e[1] = max(e[1], v1)
                    for w in range(p >> 14, ((p + max(read_len, 1) - 1) >> 14) + 1):
                        linear[tid].setdefault(w, v0)
        for _ in range(unplaced):
            name = ("u%d" % k).encode() + b"\x00"
            k += 1
            body = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, l_seq, -1, -1, 0) + name + seq + qual
            out += struct.pack("<i", len(body)) + body
        flush(final=True)
        f.write(_BGZF_EOF)
    if index:
        with open(path + ".bai", "wb") as f:
            f.write(b"BAI\x01" + struct.pack("<i", len(chroms)))
            for tid in range(len(chroms)):
                f.write(struct.pack("<i", len(bins[tid])))
                for b, (v0, v1) in sorted(bins[tid].items()):
                    f.write(struct.pack("<Ii", b, 1) + struct.pack("<QQ", v0, v1))
                n_intv = (max(linear[tid]) + 1) if linear[tid] else 0
                f.write(struct.pack("<i", n_intv))
                last = 0
                for w in range(n_intv):
                    last = linear[tid].get(w, last)
                    f.write(struct.pack("<Q", last))
            f.write(struct.pack("<Q", 0))


def write_bam_fast(path, sample, chroms, read_len=100, level=1, block_payload=65280, realistic=False, seed=1, index=False):
    """Vectorised writer for large synthetic BAMs: every record has the same layout (fixed-width read name,
    <read_len>M CIGAR, all-A sequence, quality 30), so a chromosome is one numpy structured array.
    chroms: list of (name, length, pos int32[n] sorted, mapq uint8[n]).
    index=True also writes path + '.bai': per reference one chunk in bin 0 and the 16 kb linear index."""
    text = "@HD\tVN:1.6\tSO:coordinate\n"
    for c in chroms:
        text += "@SQ\tSN:%s\tLN:%d\n" % (c[0], c[1])
    text += "@RG\tID:rg1\tSM:%s\tPL:ILLUMINA\n" % sample
    tb = text.encode()
    head = bytearray(b"BAM\x01" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(chroms)))
    for c in chroms:
        nb = c[0].encode() + b"\x00"
        head += struct.pack("<i", len(nb)) + nb + struct.pack("<i", c[1])
    n_seq = (read_len + 1) // 2
    rec = np.dtype([("block_size", "<i4"), ("ref_id", "<i4"), ("pos", "<i4"), ("l_read_name", "u1"), ("mapq", "u1"),
                    ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_ref", "<i4"),
                    ("next_pos", "<i4"), ("tlen", "<i4"), ("name", "S12"), ("cigar", "<u4"), ("seq", "u1", (n_seq,)),
                    ("qual", "u1", (read_len,))])
    with open(path, "wb") as f:
        pending = bytes(head)

        def emit(buf, final=False):
            nonlocal pending
            data = pending + buf
            n_full = len(data) // block_payload
            for b in range(n_full):
                block_off.append(f.tell())
                f.write(_bgzf_block_level(data[b * block_payload:(b + 1) * block_payload], level))
            pending = data[n_full * block_payload:]
            if final and pending:
                block_off.append(f.tell())
                f.write(_bgzf_block_level(pending, level))
                pending = b""

        serial = 0
        block_off = []                 # file offset of every BGZF block, in stream order
        first_rec = []                 # per reference: index of its first record in the stream
        for tid, c in enumerate(chroms):
            pos = np.asarray(c[2], np.int32)
            first_rec.append(serial)
            for a in range(0, len(pos), 1 << 20):
                p = pos[a:a + (1 << 20)]
                r = np.zeros(len(p), dtype=rec)
                r["block_size"] = rec.itemsize - 4
                r["ref_id"] = tid
                r["pos"] = p
                r["l_read_name"] = 12
                r["mapq"] = np.asarray(c[3], np.uint8)[a:a + (1 << 20)]
                r["bin"] = 4681 + (p >> 14)      # reg2bin of a read inside one 16 kb bin (exact bins do not matter here)
                r["n_cigar"] = 1
                r["l_seq"] = read_len
                r["next_ref"] = -1
                r["next_pos"] = -1
                ids = np.arange(serial, serial + len(p))
                serial += len(p)
                r["name"] = np.char.add("r", np.char.zfill(ids.astype("U10"), 10)).astype("S12")
                r["cigar"] = (read_len << 4)
                if realistic:   # random bases and qualities: compresses about as poorly as real data
                    rr = np.random.default_rng(seed + serial)
                    nib = np.array([1, 2, 4, 8], np.uint8)
                    r["seq"] = (nib[rr.integers(0, 4, (len(p), n_seq))] << 4) | nib[rr.integers(0, 4, (len(p), n_seq))]
                    r["qual"] = rr.integers(2, 41, (len(p), read_len), dtype=np.uint8)
                else:
                    r["seq"] = 0x11
                    r["qual"] = 30
                emit(r.tobytes())
        emit(b"", final=True)
        block_off.append(f.tell())     # (the EOF block: where a virtual offset behind the last record points)
        f.write(_BGZF_EOF)
    if index:
        boff = np.asarray(block_off, np.uint64)

        def voffset(rec_index):
            at = np.asarray(rec_index, np.uint64) * np.uint64(rec.itemsize) + np.uint64(len(head))
            return (boff[(at // np.uint64(block_payload)).astype(np.int64)] << np.uint64(16)) | (at % np.uint64(block_payload))
        with open(path + ".bai", "wb") as f:
            f.write(b"BAI\x01" + struct.pack("<i", len(chroms)))
            for tid, c in enumerate(chroms):
                pos = np.asarray(c[2], np.int64)
                if len(pos) == 0:
                    f.write(struct.pack("<ii", 0, 0))
                    continue
                v0, v1 = int(voffset(first_rec[tid])), int(voffset(first_rec[tid] + len(pos)))
                f.write(struct.pack("<i", 1) + struct.pack("<Ii", 0, 1) + struct.pack("<QQ", v0, v1))
                n_intv = int((pos[-1] + read_len - 1) >> 14) + 1
                # first record overlapping each window: the first one that ends behind the window's start
                first = np.searchsorted(pos, np.arange(n_intv, dtype=np.int64) * 16384 - read_len + 1, side="left")
                lin = voffset(first_rec[tid] + np.minimum(first, len(pos) - 1))
                lin[first >= len(pos)] = 0
                lin[np.arange(n_intv) < (pos[0] >> 14)] = 0                # nothing overlaps the windows in front of the first read
                f.write(struct.pack("<i", n_intv) + lin.astype("<u8").tobytes())
            f.write(struct.pack("<Q", 0))


def _bgzf_block_level(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(data) + co.flush()
    bsize = len(cdata) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
            + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
