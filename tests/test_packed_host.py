"""The producer's side of conga_sample_reads_packed (include/conga_hip.h; the seam is count_reads_bam, bam_data.c:192-221): capi.encode_packed
turns sorted positions into differences of 4 to 16 bits plus an exception list.  Checked here without a GPU: a plain decoder of the
format as the header states it gives the positions back at every width, the width chosen by default keeps exceptions rare, and the
one-buffer form carries the exceptions where the header says."""
import numpy as np
import pytest

from conga_amd import capi


def decode(bits, width, esc_index, esc_pos, n):
    """difference i = bits [i * width, (i + 1) * width) of the stream, least significant bit first; all ones = the next exception"""
    if width == 16:
        v = bits.view("<u2")[:n].astype(np.int64)
    else:
        b = np.unpackbits(bits, bitorder="little")[:((n + 7) // 8 * 8) * width].reshape(-1, width)
        v = (b.astype(np.int64) << np.arange(width)).sum(1)[:n]
    top = (1 << width) - 1
    exc = dict(zip(esc_index.tolist(), esc_pos.tolist()))
    out = np.zeros(n, np.int64)
    for i in range(n):
        out[i] = exc[i] if v[i] == top else out[i - 1] + v[i]
    assert sorted(exc) == np.flatnonzero(v == top).tolist()   # every all-ones value has its entry, and nothing else has one
    return out


@pytest.mark.parametrize("mean_gap", [3, 20, 110, 900])
def test_every_width_gives_the_positions_back(mean_gap):
    rng = np.random.default_rng(mean_gap)
    n1, n2 = 3000, 2000
    a = np.cumsum(rng.geometric(1.0 / mean_gap, n1)).astype(np.int32)
    b = np.cumsum(rng.geometric(1.0 / mean_gap, n2)).astype(np.int32)   # a second chromosome: its first read is an exception
    b[700:] += 250_000                                                  # a gap no width holds
    pos = np.concatenate([a, b])
    off = np.array([0, n1, n1, n1 + n2], np.uint64)                     # (an empty chromosome in between)
    for width in list(range(4, 17)) + [None]:
        bits, w, ei, ep = capi.encode_packed(pos, off, width)
        assert w == (width or w) and 4 <= w <= 16
        assert len(bits) == ((len(pos) + 7) // 8 * w if w != 16 else 2 * len(pos))
        assert np.array_equal(decode(bits, w, ei, ep, len(pos)), pos), width
        assert {0, n1, n1 + 700} <= set(ei.tolist())
        if width is None:   # the producer's rule: the fewest bytes among the widths that keep exceptions rare
            assert len(ei) <= max(len(pos) // 1000, 64)
    # the exceptions behind the differences in one buffer: at the next multiple of 16 bytes, indexes then positions
    bits, w, ei, ep = capi.encode_packed(pos, off, 10)
    one = capi.pack_inline(bits, ei, ep)
    at = (len(bits) + 15) // 16 * 16
    assert np.array_equal(one[:len(bits)], bits)
    assert np.array_equal(one[at:at + 4 * len(ei)].view("<u4"), ei) and np.array_equal(one[at + 4 * len(ei):at + 8 * len(ei)].view("<i4"), ep)


def test_unsorted_positions_travel_as_exceptions():
    pos = np.array([100, 90, 95, 4000, 10], np.int32)   # (a position in front of its predecessor cannot be a difference)
    bits, w, ei, ep = capi.encode_packed(pos, np.array([0, 5], np.uint64), 8)
    assert ei.tolist() == [0, 1, 3, 4] and ep.tolist() == [100, 90, 4000, 10]
    assert np.array_equal(decode(bits, w, ei, ep, 5), pos)
