"""The BGZF inflate stage of conga_reads_bgzf alone (conga_inflate_blocks), against zlib: the reference's BAM loop reads
through htslib's BGZF layer (bam_data.c:253-259,293,201), which inflates every block with zlib and checks its CRC32.
Every deflate block type, zlib level and strategy, tiny and full blocks, long repeats (overlapping matches), long
Huffman codes (second-level tables), damaged streams and wrong CRCs."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def deflate(data, level=6, strategy=0):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return co.compress(data) + co.flush()


def pack(streams):
    """-> (bytes, [(data_off, data_len, inflated_len, crc32)]) with some junk between the streams (odd alignments)."""
    raw, blocks = bytearray(b"\x01\x02\x03"), []
    for k, (comp, plain) in enumerate(streams):
        blocks.append((len(raw), len(comp), len(plain), zlib.crc32(plain) & 0xFFFFFFFF))
        raw += comp + b"\xEE" * (k % 4)
    return np.frombuffer(bytes(raw), np.uint8), blocks


def payloads(rng):
    """Plain texts that exercise different corners of deflate."""
    out = []
    out.append(bytes(rng.integers(0, 256, 65280, dtype=np.uint8)))                           # incompressible: stored blocks
    out.append(bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 65280)))                   # 2-bit entropy
    out.append(b"A" * 65280)                                                                   # distance 1, length 258 runs
    out.append(b"abcde" * 13056)                                                               # overlapping matches, distance 5
    out.append(bytes(rng.integers(0, 256, 7, dtype=np.uint8)))                                # tiny
    out.append(b"x")
    # BAM-like records: fixed fields, a running name, packed bases, noisy qualities
    rec = bytearray()
    for i in range(290):
        rec += struct.pack("<iiiBBHHHIiii", 220, 3, 1_000_000 + 37 * i, 12, 60, 4681, 1, 0, 100, -1, -1, 0)
        rec += b"read%07d\x00" % i + struct.pack("<I", 100 << 4)
        rec += bytes(rng.integers(0, 256, 50, dtype=np.uint8) & 0x33 | 0x11)
        rec += bytes(rng.integers(2, 41, 100, dtype=np.uint8))
    out.append(bytes(rec[:65280]))
    # a skewed alphabet of 256 symbols: code lengths up to 15 (second-level tables on the literal side)
    p = 0.5 ** np.arange(1, 257, dtype=np.float64)
    p[20:] = p[20] / 236
    out.append(bytes(rng.choice(256, 60000, p=p / p.sum()).astype(np.uint8)))
    # far matches with many different distances (second-level tables on the distance side)
    base = bytes(rng.integers(0, 256, 30000, dtype=np.uint8))
    far = bytearray(base)
    for _ in range(600):
        a = int(rng.integers(0, len(far) - 40))
        far += far[a:a + int(rng.integers(3, 40))]
    out.append(bytes(far[:65280]))
    out.append(b"".join(b"%d\t%d\tchr%d\n" % (i * 7919 % 100003, i, i % 23) for i in range(4000))[:65000])   # text
    return out


@pytest.mark.parametrize("loop", ["two_phase", "wave1"])
def test_every_kind_of_stream_matches_zlib(capi, loop, monkeypatch):
    """(wave1: round 2's symbol loop, kept behind CONGA_BGZF_KERNEL for comparison -- it reads the same table entries)"""
    if loop == "wave1":
        monkeypatch.setenv("CONGA_DEBUG", "1")   # (measurement switches are read only with it: conga_amd/csrc/engine_knobs.h)
        monkeypatch.setenv("CONGA_BGZF_KERNEL", "wave1")
    rng = np.random.default_rng(5)
    streams = []
    for plain in payloads(rng):
        for level, strategy in ((1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (0, 0), (4, zlib.Z_FILTERED)):
            streams.append((deflate(plain, level, strategy), plain))
    # several deflate blocks in one stream (a full flush in the middle, and a stored block between two dynamic ones)
    a, b, c = payloads(np.random.default_rng(6))[1][:20000], bytes(rng.integers(0, 256, 3000, dtype=np.uint8)), b"tail" * 2000
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    multi = co.compress(a) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(b) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(c) + co.flush()
    streams.append((multi, a + b + c))
    data, blocks = pack(streams)
    with capi.Context(device=0) as ctx:
        out, status, _ms = ctx.inflate_blocks(data, blocks)
    assert status.tolist() == [0] * len(blocks), [i for i, s in enumerate(status) if s]
    at = 0
    for k, (_comp, plain) in enumerate(streams):
        assert out[at:at + len(plain)].tobytes() == plain, "stream %d" % k
        at += len(plain)


def test_matches_of_every_length_near_and_far(capi):
    """Matches of 3..70 bytes (the sizes a lane copies on its own: up to 8, 9-16, 17-32, 33-64; beyond: all lanes) at distances
    equal to the length, one more, a few hundred (inside the batch of sixty-four symbols that is turned into output at once) and
    thousands (in front of it), and matches that repeat themselves (distance below the length), between literals -- every
    stream compared with the plain text zlib made it from."""
    rng = np.random.default_rng(11)
    streams = []
    for variant in range(6):
        plain = bytearray(rng.integers(0, 256, 6000, dtype=np.uint8).tobytes())
        for n in range(3, 71):
            for dist in (n, n + 1, int(rng.integers(n + 2, 400)), int(rng.integers(1000, 5000)), max(1, n - int(rng.integers(1, n)))):
                plain += bytes(rng.integers(0, 256, int(rng.integers(0, 4)), dtype=np.uint8))   # 0..3 literals between two matches
                a = len(plain) - dist
                for k in range(n):                                                              # (byte by byte: a match may repeat itself)
                    plain.append(plain[a + k])
            if len(plain) > 60000:
                break
        plain = bytes(plain[:65280])
        for level, strategy in ((1, 0), (6, 0), (9, 0), (6, zlib.Z_FIXED)):
            streams.append((deflate(plain, level, strategy), plain))
    # a stream whose LAST symbol is such a match, its source the stream's very first bytes: nothing may be read in front of the
    # output or written behind its end (the blocks' outputs lie side by side: the next stream's bytes would be the ones hit)
    for n in range(3, 71):
        head = bytes(rng.integers(0, 256, n + int(rng.integers(0, 40)), dtype=np.uint8))
        plain = head + head[:n]
        streams.append((deflate(plain, 9, 0), plain))
        plain = head + bytes(rng.integers(0, 256, 300, dtype=np.uint8)) + head[:n]
        streams.append((deflate(plain, 6, zlib.Z_FIXED), plain))
    data, blocks = pack(streams)
    with capi.Context(device=0) as ctx:
        out, status, _ms = ctx.inflate_blocks(data, blocks)
    assert status.tolist() == [0] * len(blocks), [i for i, s in enumerate(status) if s]
    at = 0
    for k, (_comp, plain) in enumerate(streams):
        assert out[at:at + len(plain)].tobytes() == plain, "stream %d" % k
        at += len(plain)


def test_damaged_streams_are_refused_not_believed(capi):
    rng = np.random.default_rng(8)
    plain = payloads(rng)[6]
    good = deflate(plain)
    streams, want = [], []
    for k in range(60):   # a flipped bit somewhere: refused, or decoded to something whose CRC32 is wrong
        bad = bytearray(good)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        streams.append((bytes(bad), plain))
    streams.append((good[:len(good) // 2], plain))             # truncated
    streams.append((good, plain + b"!"))                       # recorded size too large
    streams.append((good, plain[:-1]))                         # ... too small
    streams.append((b"\x07" + good, plain))                    # reserved block type 3
    data, blocks = pack(streams)
    blocks.append((blocks[0][0], len(good), len(plain), 0x12345678))   # right stream, wrong CRC... on a damaged copy
    blocks.append((len(data) - len(b"\xEE" * 3) - len(good) - 1 + 1, len(good), len(plain), zlib.crc32(plain) & 0xFFFFFFFF))
    with capi.Context(device=0) as ctx:
        _out, status, _ = ctx.inflate_blocks(data, blocks)
        # whatever zlib accepts with the right size and CRC must be accepted here too, and nothing else
        for k, (off, n, isz, crc) in enumerate(blocks):
            try:
                got = zlib.decompress(data[off:off + n].tobytes(), -15)
                ok = len(got) == isz and (zlib.crc32(got) & 0xFFFFFFFF) == crc
            except zlib.error:
                ok = False
            assert (status[k] == 0) == ok, (k, int(status[k]), ok)
        assert (status != 0).sum() >= 55
        # the undamaged stream still decodes in the same call, after all that
        out, st, _ = ctx.inflate_blocks(*pack([(good, plain)]))
        assert st.tolist() == [0] and out.tobytes() == plain


def test_many_blocks_round_robin_over_the_waves(capi):
    """More blocks than resident waves, of very different sizes: every wave takes several in turn (LDS tables and the
    input window are set up again per block)."""
    rng = np.random.default_rng(11)
    kinds = payloads(rng)
    streams = []
    for k in range(9000):
        plain = kinds[k % len(kinds)]
        a = int(rng.integers(0, max(len(plain) - 10, 1)))
        plain = plain[a:a + int(rng.integers(1, 3000))]
        streams.append((deflate(plain, int(rng.integers(1, 10))), plain))
    data, blocks = pack(streams)
    with capi.Context(device=0) as ctx:
        out, status, ms = ctx.inflate_blocks(data, blocks)
    assert not status.any()
    assert out.tobytes() == b"".join(p for _c, p in streams)


def test_damage_fuzz_agrees_with_zlib_on_what_is_a_stream(capi):
    """A few thousand damaged streams of every kind (flipped bits, flipped bytes in the block headers, truncations, spliced
    and random bytes): the kernel accepts exactly what zlib inflates to the recorded size with the recorded CRC32, refuses
    the rest, and comes back (a table with holes, a distance past the output's start or a code that never ends must end the
    wave's walk, not spin it)."""
    rng = np.random.default_rng(20261004)
    kinds = payloads(rng)
    streams = []
    for k in range(3600):
        plain = kinds[k % len(kinds)]
        a = int(rng.integers(0, max(len(plain) - 10, 1)))
        plain = plain[a:a + int(rng.integers(1, 6000))]
        good = deflate(plain, int(rng.integers(1, 10)), int(rng.choice([0, 0, 0, 2, 3, 4])))
        bad = bytearray(good)
        how = k % 6
        if how == 0:      # one flipped bit anywhere
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:    # a flipped bit in the first bytes: block type, code-length code, the tables themselves
            bad[int(rng.integers(0, min(len(bad), 40)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 2:    # a few flipped bytes
            for _ in range(int(rng.integers(1, 5))):
                bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        elif how == 3:    # cut short
            del bad[int(rng.integers(0, len(bad))):]
            bad += b"\x00" if not bad else b""
        elif how == 4:    # another stream's tail spliced in
            other = deflate(kinds[(k + 3) % len(kinds)][:3000], 6)
            cut = int(rng.integers(1, len(bad) + 1))
            bad = bad[:cut] + other[int(rng.integers(0, len(other))):]
        else:             # noise behind a dynamic-block header (BTYPE 2) or all noise
            noise = bytes(rng.integers(0, 256, int(rng.integers(4, 400)), dtype=np.uint8))
            bad = bytearray((b"\x05" if k % 12 == 5 else b"") + noise)
        streams.append((bytes(bad), plain))
    data, blocks = pack(streams)
    want = []
    for off, n, isz, crc in blocks:
        try:
            got = zlib.decompress(data[off:off + n].tobytes(), -15)
            want.append(len(got) == isz and (zlib.crc32(got) & 0xFFFFFFFF) == crc)
        except zlib.error:
            want.append(False)
    with capi.Context(device=0) as ctx:
        _out, status, _ = ctx.inflate_blocks(data, blocks, want_out=False)
    got = status == 0
    wrong = [k for k in range(len(blocks)) if bool(got[k]) != want[k]]
    assert not wrong, (len(wrong), wrong[:10], [int(status[k]) for k in wrong[:10]])
    assert 0 < sum(want) < len(want) // 2   # (a flipped bit sometimes lands where it changes nothing that is checked... rarely)
