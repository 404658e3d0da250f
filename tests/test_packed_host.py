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


# ---- the library's own producer (conga_packer_*, conga_amd/csrc/pack_host.h): host threads, no device -------------------------------
def _case(rng, n_chrom, mean_gap, n_max):
    chunks, off = [], [0]
    for c in range(n_chrom):
        n = int(rng.integers(0, n_max)) if rng.random() > 0.15 else 0      # (some chromosomes are empty)
        p = np.cumsum(rng.geometric(1.0 / mean_gap, n)).astype(np.int64)
        if n > 50 and rng.random() < 0.5:
            k = int(rng.integers(1, n))
            p[k:] += int(rng.integers(1 << 15, 1 << 22))                    # a gap no width holds
        if n > 10 and rng.random() < 0.3:
            k = int(rng.integers(1, n))
            p[k] = p[k - 1] - int(rng.integers(1, 50))                      # a position in front of its predecessor
        chunks.append(p.astype(np.int32))
        off.append(off[-1] + n)
    pos = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
    return np.ascontiguousarray(pos, np.int32), np.array(off, np.uint64)


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_c_producer_writes_what_the_numpy_encoder_writes(threads):
    """Every width, ragged sample sizes (not multiples of 8 or of the packer's runs of 8 192 reads), empty chromosomes, chromosome
    borders inside a group of eight, gaps and reads out of order: the one-copy buffer is byte for byte capi.pack_inline(encode_packed)."""
    rng = np.random.default_rng(threads)
    with capi.Packer(threads) as pk:
        for trial in range(12):
            pos, off = _case(rng, int(rng.integers(1, 7)), float(rng.choice([3, 20, 110, 900])), int(rng.choice([9, 100, 8192, 30000])))
            n = len(pos)
            out = np.full(pk.bound(n, n) + 8, 0xAB, np.uint8)
            for width in list(range(4, 17)):
                pk.start(pos, off, out, width)
                w, n_esc, nbytes = pk.finish()
                bits, w2, ei, ep = capi.encode_packed(pos, off, width)
                want = capi.pack_inline(bits, ei, ep)
                assert (w, n_esc) == (w2, len(ei)) and nbytes == ((len(bits) + 15) & ~15) + 8 * len(ei), (trial, width)
                assert np.array_equal(out[:nbytes], want[:nbytes]), (trial, width)
                assert (out[pk.bound(n, n):] == 0xAB).all()               # nothing behind what the bound promises
                # conga_packer_start_v -- one array per chromosome, each an allocation of its own -- writes the same bytes
                parts = [pos[int(off[c]):int(off[c + 1])].copy() for c in range(len(off) - 1)]
                out_v = np.full(len(out), 0xAB, np.uint8)
                assert np.array_equal(pk.start_v(parts, out_v, width), off)
                assert pk.finish() == (w, n_esc, nbytes) and np.array_equal(out_v[:nbytes], out[:nbytes]), (trial, width)
            # width 0: the producer's rule on a sample of the differences -- exceptions stay rare, and the positions come back
            pk.start(pos, off, out, 0)
            w, n_esc, nbytes = pk.finish()
            assert 4 <= w <= 16
            at = ((n + 7) // 8 * w + 15) & ~15
            ei, ep = out[at:at + 4 * n_esc].view("<u4"), out[at + 4 * n_esc:at + 8 * n_esc].view("<i4")
            assert np.array_equal(decode(out[:(n + 7) // 8 * w].copy(), w, ei, ep, n), pos)
            out_v = np.full(len(out), 0xAB, np.uint8)
            pk.start_v([pos[int(off[c]):int(off[c + 1])].copy() for c in range(len(off) - 1)], out_v, 0)
            assert pk.finish() == (w, n_esc, nbytes) and np.array_equal(out_v[:nbytes], out[:nbytes])   # (the same width from the same reads)


def test_c_producer_into_a_buffer_at_any_address():
    """The producer's runs leave the core with non-temporal 16-byte stores when the destination allows it and with plain ones when it
    does not (pack_host.h: stream_copy): a buffer that starts 1, 3, 8 or 15 bytes behind a 16-byte boundary gets the same bytes, and
    nothing in front of it or behind what the bound promises is touched."""
    rng = np.random.default_rng(11)
    with capi.Packer(3) as pk:
        pos, off = _case(rng, 4, 110.0, 30000)
        n = len(pos)
        assert n > 3 * 8192                                              # (several whole runs and a ragged one)
        for width in (5, 10, 16, 0):
            base = np.full(pk.bound(n, n) + 64, 0xAB, np.uint8)
            a0 = (-base.ctypes.data) % 16                                 # first 16-byte boundary inside `base`
            pk.start(pos, off, base[a0:], width)
            w, n_esc, nbytes = pk.finish()
            want = base[a0:a0 + nbytes].copy()
            for shift in (1, 3, 8, 15):
                buf = np.full(len(base), 0xAB, np.uint8)
                out = buf[a0 + shift:]
                assert out.ctypes.data % 16 == shift
                pk.start(pos, off, out, w)
                assert pk.finish() == (w, n_esc, nbytes)
                assert np.array_equal(out[:nbytes], want), (width, shift)
                assert (buf[:a0 + shift] == 0xAB).all() and (out[pk.bound(n, n):] == 0xAB).all(), (width, shift)


def test_c_producer_picks_the_bench_widths_and_refuses_what_does_not_fit():
    rng = np.random.default_rng(7)
    with capi.Packer(2) as pk:
        for cov, want in ((1.0, 10), (5.0, 8), (30.0, 5)):               # (10 bits at 1x, 8 at 5x, 5 at 30x)
            n = 400_000
            pos = np.cumsum(rng.geometric(cov / 100.0, n)).astype(np.int32)
            off = np.array([0, n], np.uint64)
            out = np.zeros(pk.bound(n, n // 50), np.uint8)
            pk.start(pos, off, out, 0)
            w, n_esc, _ = pk.finish()
            assert abs(w - want) <= 1 and n_esc <= n // 500, (cov, w, n_esc)
        small = np.zeros(64, np.uint8)
        with pytest.raises(capi.CongaError):
            pk.start(pos, off, small, 10)                                   # (CONGA_ERR_NOMEM: the differences alone do not fit)
        pk.start(np.zeros(0, np.int32), np.array([0, 0, 0], np.uint64), small, 0)   # an empty sample is a sample
        assert pk.finish()[1:] == (0, 0)


def test_c_producer_under_thread_sanitizer(tmp_path):
    """pack_host.h is host-only C++: built here with g++ -fsanitize=thread and driven through several samples of different sizes on
    one pool (the hand-over between start() / the workers / finish() is what the sanitizer watches)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "%s/conga_amd/csrc/pack_host.h"
#include <cstdio>
int main()
{
	conga_pack::Packer pk(4);
	for (int rep = 0; rep < 6; rep++) {
		const uint64_t n = 1000 + 37000 * (uint64_t) rep;
		std::vector<int32_t> pos(n);
		int32_t p = 0;
		for (uint64_t i = 0; i < n; i++) { p += (int32_t) (i * 2654435761u %% 211u); if (i %% 9001 == 9000) p += 400000; pos[i] = p; }
		uint64_t off[4] = {0, n / 3, n / 3, n};
		std::vector<uint8_t> out(conga_pack::bound(n, n));
		std::vector<int32_t> a(pos.begin(), pos.begin() + n / 3), b(pos.begin() + n / 3, pos.end());   // (every third sample: one array per chromosome)
		const int32_t *parts[3] = {a.data(), nullptr, b.data()};
		if ((rep %% 3 == 2 ? pk.start_v(parts, off, 3, rep %% 2 ? 0 : 9, out.data(), out.size())
				: pk.start(pos.data(), off, 3, rep %% 2 ? 0 : 9, out.data(), out.size())) != 0) return 2;
		int w = 0; size_t ne = 0, nb = 0;
		if (pk.finish(&w, &ne, &nb) != 0 || ne < 2) return 3;
		// decode: the positions come back
		const size_t at = ((size_t) ((n + 7) / 8) * (size_t) w + 15) & ~(size_t) 15;
		const uint32_t *ei = (const uint32_t *) (out.data() + at);
		const int32_t *ep = (const int32_t *) (out.data() + at) + ne;
		size_t e = 0; int64_t cur = 0;
		for (uint64_t i = 0; i < n; i++) {
			uint32_t v = 0;
			for (int b = 0; b < w; b++) { const uint64_t bit = i * (uint64_t) w + (uint64_t) b; v |= (uint32_t) ((out[bit >> 3] >> (bit & 7)) & 1u) << b; }
			if (v == (1u << w) - 1u) { if (e >= ne || ei[e] != i) return 4; cur = ep[e++]; } else cur += v;
			if (cur != pos[i]) return 5;
		}
		if (e != ne) return 6;
	}
	puts("ok");
	return 0;
}
''' % root)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-o", str(exe), str(src), "-lpthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr[-3000:]
