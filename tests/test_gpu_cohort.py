"""Cohort mode (conga_sample_reads / conga_sample_begin + the staging ring / conga_sample_fetch): further samples behind
a layout that was handed over once give exactly what a fresh context gives, and what the oracle gives.

The reference runs one process per sample (svdepth.c:47-66); per sample its hot path is count_reads_bam ->
calc_mean_per_chr -> find_SVs (bam_data.c:192-221, read_distribution.c:49-84, likelihood.c:311-371), which is what each
case below is checked against.
"""
import numpy as np
import pytest

from conga_amd import synth
from test_gpu_parity import assert_records, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["tuple_space", "dense"])
def formulation(request, capi):
    capi.EXTRA_FLAGS = capi.FLAG_MATERIALIZE_DEPTH if request.param == "dense" else 0
    yield request.param
    capi.EXTRA_FLAGS = 0


def layout(with_map):
    """Three chromosomes with their intervals (and tracks); the reads of `chroms` are sample 0."""
    specs = (("7", 900_000, 40, 9), ("9", 400_123, 25, 4), ("11", 1_300_000, 50, 12))
    chroms = []
    for name, L, nd, nu in specs:
        c = synth.make_chrom(name, L, cov=1.0, n_dels=nd, n_dups=nu, mappability=with_map, gaps=(L > 1_000_000))
        ds, de = synth.kept_sorted(c.del_start, c.del_end)
        us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
        chroms.append((c, ds, de, us, ue))
    return chroms


def sample_reads_of(chroms, sample, cov, empty=None, mapq_floor=None):
    """Another individual's reads on the same chromosomes (same GC track, other seed and coverage)."""
    out = []
    for k, (c, *_rest) in enumerate(chroms):
        rng = np.random.default_rng([sample, k, 77])
        pos, mapq = synth.make_reads(c.length, c.gc, c.step, cov, 100, rng)
        if empty == k:
            pos, mapq = pos[:0], mapq[:0]
        if mapq_floor is not None:
            mapq = np.maximum(mapq, mapq_floor).astype(np.uint8)
        out.append((pos, mapq))
    return out


def open_layout(ctx, chroms, with_map):
    for c, ds, de, us, ue in chroms:
        ctx.chrom_begin(c.length, c.gc)
        if with_map:
            ctx.mappability(c.map_start, c.map_end, c.map_val)
        ctx.intervals("D", ds, de)
        ctx.intervals("E", us, ue)


def pinned_sample(ctx, reads):
    n = sum(len(p) for p, _ in reads)
    pos, mapq = ctx.host_alloc(max(n, 1), np.int32), ctx.host_alloc(max(n, 1), np.uint8)
    off = np.zeros(len(reads) + 1, np.uint64)
    at = 0
    for k, (p, m) in enumerate(reads):
        pos[at:at + len(p)] = p
        mapq[at:at + len(p)] = m
        at += len(p)
        off[k + 1] = at
    return pos, mapq, off


def check_against_oracle(oracle, chroms, reads, recs, E, with_map, mq=-1):
    at = 0
    for k, ((c, ds, de, us, ue), (pos, mapq)) in enumerate(zip(chroms, reads)):
        rows = (c.map_start, c.map_end, c.map_val) if with_map else None
        want = run_oracle(oracle, c.length, c.gc, pos, mapq, ds, de, us, ue, mq=mq, rows=rows)
        assert np.array_equal(E[k].view(np.uint32), want["E"].view(np.uint32)), "expected_read_depth of chromosome %d" % k
        assert_records(recs[at:at + len(ds)], want["dels"], with_map)
        assert_records(recs[at + len(ds):at + len(ds) + len(us)], want["dups"], with_map)
        at += len(ds) + len(us)
    assert at == len(recs)


@pytest.mark.parametrize("with_map", [False, True])
def test_samples_behind_one_layout(capi, oracle, with_map):
    chroms = layout(with_map)
    samples = [[(c.pos, c.mapq) for c, *_ in chroms],
               sample_reads_of(chroms, 1, 3.0),
               sample_reads_of(chroms, 2, 0.3, empty=1),      # one chromosome without a read
               sample_reads_of(chroms, 3, 1.0)]
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        open_layout(ctx, chroms, with_map)
        for si, reads in enumerate(samples):
            pos, mapq, off = pinned_sample(ctx, reads)
            ctx.sample_reads(pos, mapq, off)
            ctx.compute()
            recs, E, st = ctx.sample_fetch(want_stats=True)
            check_against_oracle(oracle, chroms, reads, recs, E, with_map)
            for k, (p, _m) in enumerate(reads):
                assert st[k].reads_committed == len(p) and st[k].reads_counted == len(p)
            # ... and exactly what a context that never saw another sample gives
            with capi.Context(device=0, flags=capi.FLAG_BATCH) as fresh:
                for (c, ds, de, us, ue), (p, m) in zip(chroms, reads):
                    fresh.chrom_begin(c.length, c.gc)
                    fresh.reads(p, m)
                    if with_map:
                        fresh.mappability(c.map_start, c.map_end, c.map_val)
                    fresh.intervals("D", ds, de)
                    fresh.intervals("E", us, ue)
                fresh.compute()
                want = b"".join(d.tobytes() + u.tobytes() for d, u, _e, _s in fresh.fetch_all())
            assert recs.tobytes() == want, "sample %d differs from a fresh context's records" % si
            # the one-chromosome fetch still works behind a sample
            ctx.select(1)
            d1, u1 = ctx.fetch()[:2]
            a = len(chroms[0][1]) + len(chroms[0][3])
            assert d1.tobytes() + u1.tobytes() == recs[a:a + len(d1) + len(u1)].tobytes()


def test_mapq_may_stay_at_home_under_the_default_threshold(capi, oracle):
    """`qual > -1` holds for every read (bam_data.c:205, cmdline.c:188-194): conga_sample_reads takes mapq = NULL then and sends
    4 bytes per read; with a real threshold the bytes are required."""
    chroms = layout(False)
    reads = sample_reads_of(chroms, 21, 1.5)
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        open_layout(ctx, chroms, False)
        pos, _mapq, off = pinned_sample(ctx, reads)
        ctx.sample_reads(pos, None, off)
        ctx.compute()
        recs, E, st = ctx.sample_fetch(want_stats=True)
        check_against_oracle(oracle, chroms, reads, recs, E, False)
        assert [int(x.reads_counted) for x in st] == [len(p) for p, _ in reads]
    with capi.Context(device=0, flags=capi.FLAG_BATCH, mq_threshold=0) as ctx:
        open_layout(ctx, chroms, False)
        pos, _mapq, off = pinned_sample(ctx, reads)
        with pytest.raises(capi.CongaError):
            ctx.sample_reads(pos, None, off)


def test_sample_through_the_staging_ring(capi, oracle):
    """conga_sample_begin + conga_sample_chrom: the decoder's route (one chromosome after the other through
    conga_reads_staging / conga_reads_commit), with a MAPQ threshold."""
    chroms = layout(False)
    reads = sample_reads_of(chroms, 5, 2.0, empty=0)
    with capi.Context(device=0, flags=capi.FLAG_BATCH, mq_threshold=10) as ctx:
        open_layout(ctx, chroms, False)
        ctx.sample_begin()
        for k, (p, m) in enumerate(reads):
            ctx.sample_chrom(k)
            ctx.reads(p, m)
        ctx.compute()
        recs, E, _ = ctx.sample_fetch()
        check_against_oracle(oracle, chroms, reads, recs, E, False, mq=10)
        # descending order is refused, and so is a wrong chromosome count
        with pytest.raises(capi.CongaError):
            ctx.sample_chrom(0)
        with pytest.raises(capi.CongaError):
            ctx.sample_reads(np.zeros(1, np.int32), np.zeros(1, np.uint8), np.zeros(2, np.uint64))


def test_sample_with_a_pile_up_wraps_like_a_short(capi, oracle, formulation):
    """40 000 reads on one base wrap read_depth's `short` (common.h:91, bam_data.c:213).  Reads that arrive through
    conga_sample_reads are only looked at on the device: the tuple pass flags the run and the fetch recomputes in the
    dense formulation."""
    chroms = layout(False)
    reads = sample_reads_of(chroms, 9, 1.0)
    c1 = chroms[1][0]
    hot = int(chroms[1][1][3]) + 17   # inside the fourth deletion of chromosome 1
    p, m = reads[1]
    p = np.sort(np.concatenate([p, np.full(40_000, hot, np.int32)])).astype(np.int32)
    m = np.full(len(p), 60, np.uint8)
    reads[1] = (p, m)
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        open_layout(ctx, chroms, False)
        pos, mapq, off = pinned_sample(ctx, reads)
        ctx.sample_reads(pos, mapq, off)
        ctx.compute()
        recs, E, st = ctx.sample_fetch(want_stats=True)
        assert st[1].depth_materialized == 1
        check_against_oracle(oracle, chroms, reads, recs, E, False)
        rd, _ = oracle.count_reads(c1.length, p, m, -1)
        assert rd[hot] < 0  # the case does wrap
        # the next, harmless sample goes back to tuple space (unless the dense formulation is forced)
        reads2 = sample_reads_of(chroms, 10, 1.0)
        pos2, mapq2, off2 = pinned_sample(ctx, reads2)
        ctx.sample_reads(pos2, mapq2, off2)
        ctx.compute()
        recs2, E2, st2 = ctx.sample_fetch(want_stats=True)
        assert st2[1].depth_materialized == (1 if formulation == "dense" else 0)
        check_against_oracle(oracle, chroms, reads2, recs2, E2, False)


def test_next_sample_handed_over_beside_the_compute_one_context(capi, oracle, formulation):
    """conga_sample_reads is double-buffered: sample k + 1 is handed over while sample k is computed and before it is fetched --
    its copy runs on a stream of its own into the other pair of buffers -- and every sample's records are still its own.
    One of the samples wraps a `short` (40 000 reads on one base): its fetch computes it again in the dense formulation out of
    the pair of buffers the NEXT sample's copy is not writing."""
    chroms = layout(False)
    samples = [sample_reads_of(chroms, 20 + k, [1.0, 2.0, 0.5, 1.5][k]) for k in range(4)]
    hot = int(chroms[1][1][3]) + 17
    p, m = samples[1][1]
    p = np.sort(np.concatenate([p, np.full(40_000, hot, np.int32)])).astype(np.int32)
    samples[1][1] = (p, np.full(len(p), 60, np.uint8))
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        open_layout(ctx, chroms, False)
        pinned = [pinned_sample(ctx, r) for r in samples]
        ctx.sample_reads(*pinned[0])
        ctx.compute()
        for k in range(len(samples)):
            if k + 1 < len(samples):
                ctx.sample_reads(*pinned[k + 1])      # beside compute k; before fetch k
            recs, E, st = ctx.sample_fetch(want_stats=True)
            check_against_oracle(oracle, chroms, samples[k], recs, E, False)
            assert [s.reads_committed for s in st] == [len(r[0]) for r in samples[k]]
            assert st[1].depth_materialized == (1 if (k == 1 or formulation == "dense") else 0)
            if k + 1 < len(samples):
                ctx.compute()
        # two samples handed over without a compute in between: the later one counts
        ctx.sample_reads(*pinned[2])
        ctx.sample_reads(*pinned[3])
        ctx.compute()
        recs, E, _ = ctx.sample_fetch()
        check_against_oracle(oracle, chroms, samples[3], recs, E, False)


def test_two_computes_in_flight_one_context(capi, oracle, formulation):
    """conga_chrom_compute_ahead (ABI v9): sample k + 1 is handed over AND computed before sample k is fetched -- the results of the
    compute before the latest one stay where they are (conga_sample_fetch_previous) -- and every sample's records are still its own,
    whether its positions came as 32-bit numbers or as packed differences.  Two of the samples wrap a `short` (40 000 reads on one
    base): the older compute's guard is settled with the sets changed over, out of the pair of tuple buffers no copy is writing,
    behind the launches of the compute ahead."""
    chroms = layout(False)
    covs = [1.0, 2.0, 0.5, 1.5, 0.8, 1.2]
    samples = [sample_reads_of(chroms, 40 + k, covs[k]) for k in range(len(covs))]
    hot = int(chroms[1][1][3]) + 17
    for k in (1, 4):
        p, m = samples[k][1]
        p = np.sort(np.concatenate([p, np.full(40_000, hot, np.int32)])).astype(np.int32)
        samples[k][1] = (p, np.full(len(p), 60, np.uint8))
    n = len(samples)
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        open_layout(ctx, chroms, False)
        pinned = [pinned_sample(ctx, r) for r in samples]
        packed = {}                                        # (the arrays of a hand-over stay as they are until its compute has been waited for)

        def hand_over(k):
            pos, mapq, off = pinned[k]
            if k % 2 == 0:
                ctx.sample_reads(pos, mapq, off)
            else:                                          # (the packed hand-over's expansion runs in the compute that goes ahead)
                if k not in packed:
                    bits, w, ei, ep = capi.encode_packed(pos[:int(off[-1])], off, None)
                    packed[k] = (capi.pack_inline(bits, ei, ep), w, len(ei))
                ctx.sample_reads_packed(packed[k][0], packed[k][1], packed[k][2], None, mapq, off)

        def check(k, got):
            recs, E, st = got
            check_against_oracle(oracle, chroms, samples[k], recs, E, False)
            assert [s.reads_committed for s in st] == [len(r[0]) for r in samples[k]], k
            assert st[1].depth_materialized == (1 if (k in (1, 4) or formulation == "dense") else 0), k

        with pytest.raises(capi.CongaError):
            ctx.sample_fetch_previous()                    # nothing computed, nothing kept
        hand_over(0)
        ctx.compute_ahead()                                # (nothing to keep yet: a plain compute)
        with pytest.raises(capi.CongaError):
            ctx.sample_fetch_previous()
        for k in range(n):
            if k + 1 < n:
                hand_over(k + 1)
                ctx.compute_ahead()                        # behind compute k, which nobody has waited for
                check(k, ctx.sample_fetch_previous(want_stats=True))
            else:
                check(k, ctx.sample_fetch(want_stats=True))
        # a caller that hands the NEXT sample over before it has fetched the older one: the hand-over settles that one's guard
        # first (it is sample 4's pair of buffers the copy goes into), the records are fetched afterwards all the same
        hand_over(4)
        ctx.compute()
        hand_over(2)
        ctx.compute_ahead()
        hand_over(3)                                       # sample 4 (it wraps) has not been fetched
        check(4, ctx.sample_fetch_previous(want_stats=True))
        ctx.compute_ahead()                                # keeps sample 2, computes sample 3
        check(2, ctx.sample_fetch_previous(want_stats=True))
        check(3, ctx.sample_fetch(want_stats=True))
        # both computes in flight wrap, and the LATEST one is fetched first: its guard sends it to the dense kernels and leaves the
        # context knowing that its sample wraps -- which says nothing about the older one (tests/soak.py --ahead, seed 901 case 11:
        # the older sample's records came back as tuple space had them)
        hand_over(1)
        ctx.compute()
        hand_over(4)
        ctx.compute_ahead()
        check(4, ctx.sample_fetch(want_stats=True))
        check(1, ctx.sample_fetch_previous(want_stats=True))
        check(4, ctx.sample_fetch(want_stats=True))
        # a plain compute gives the older results up
        hand_over(0)
        ctx.compute()
        with pytest.raises(capi.CongaError):
            ctx.sample_fetch_previous()
        check(0, ctx.sample_fetch(want_stats=True))


def test_records_of_the_compute_before_the_latest_on_the_device(capi, oracle):
    """conga_sync_previous + conga_results_copy_previous (the multi-GPU loop's half of ABI v9): with CONGA_FLAG_RESULTS_ON_DEVICE the
    older compute's records are copied device to device while the latest compute's launches are in the queues."""
    import ctypes
    chroms = layout(False)
    samples = [sample_reads_of(chroms, 50 + k, 1.0 + 0.5 * k) for k in range(3)]
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as plain:
        open_layout(plain, chroms, False)
        want = []
        for r in samples:
            plain.sample_reads(*pinned_sample(plain, r))
            plain.compute()
            want.append(plain.sample_fetch()[0].tobytes())
    # device buffers from the HIP runtime the library is linked against (torch ships another and wants a process of its own)
    paths = sorted({line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}, key=lambda q: ("/opt/rocm" not in q, q))
    hip = ctypes.CDLL(paths[0])
    nbytes = len(want[0])
    bufs = []
    for _ in samples:
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
        assert hip.hipMemset(p, 0, ctypes.c_size_t(nbytes)) == 0
        bufs.append(p)
    try:
        with capi.Context(device=0, flags=capi.FLAG_BATCH | capi.FLAG_RESULTS_ON_DEVICE) as ctx:
            open_layout(ctx, chroms, False)
            pinned = [pinned_sample(ctx, r) for r in samples]
            ctx.sample_reads(*pinned[0])
            ctx.compute()
            for k in range(len(samples)):
                if k + 1 < len(samples):
                    ctx.sample_reads(*pinned[k + 1])
                    ctx.compute_ahead()
                    ctx.sync_previous()
                    ctx.results_copy_previous(bufs[k].value, nbytes)
                else:
                    ctx.sync()
                    ctx.results_copy(bufs[k].value, nbytes)
            ctx.sync()
            for k in range(len(samples)):
                got = np.zeros(nbytes, np.uint8)
                assert hip.hipMemcpy(ctypes.c_void_p(got.ctypes.data), bufs[k], ctypes.c_size_t(nbytes), 2) == 0   # hipMemcpyDeviceToHost
                assert got.tobytes() == want[k], k
    finally:
        for p in bufs:
            hip.hipFree(p)


def test_positions_as_16_bit_differences_give_the_same_records(capi, oracle, formulation):
    """conga_sample_reads_d16: the same samples handed over as differences + exceptions (a gap of 65 535 bases and more, one of
    exactly 65 534 and 65 535, equal neighbours, a chromosome without reads, a chromosome of one read, chunk borders of the scan
    on chromosome borders) give byte for byte the records of the 32-bit hand-over; a position in front of its predecessor travels
    as an exception and is refused as unsorted, as it is there."""
    chroms = layout(False)
    for sample, empty in ((30, None), (31, 1), (32, 0)):
        reads = sample_reads_of(chroms, sample, [1.0, 3.0, 0.3][sample - 30], empty=empty)
        p, m = reads[2]
        keep = (p < 300_000) | (p > 300_000 + 65_534 + 70_000)          # a long gap ...
        p = np.sort(np.concatenate([p[keep], [300_000, 300_000 + 65_534, 300_000 + 65_534 + 65_535, 300_000 + 65_534 + 65_535]])).astype(np.int32)
        p = p[~((p > 300_000) & (p < 300_000 + 65_534))]                # ... with differences of exactly 65 534, 65 535 and 0 at its rim
        reads[2] = (p, np.full(len(p), 60, np.uint8))
        if sample == 32:
            reads[1] = (reads[1][0][:1], reads[1][1][:1])                # one read
        with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
            open_layout(ctx, chroms, False)
            pos, mapq, off = pinned_sample(ctx, reads)
            ctx.sample_reads(pos, mapq, off)
            ctx.compute()
            want, wantE, _ = ctx.sample_fetch()
            delta, ei, ep = capi.encode_d16(pos[:int(off[-1])], off)
            assert len(ei) >= 3 and int((delta == 0xFFFF).sum()) == len(ei)
            d_pin = ctx.host_alloc(max(len(delta), 1), np.uint16)
            d_pin[:len(delta)] = delta
            for _ in range(2):                                           # (the second time beside the first's compute: double-buffered)
                ctx.sample_reads_d16(d_pin, ei, ep, mapq, off)
                ctx.compute()
                got, gotE, _ = ctx.sample_fetch()
                assert got.tobytes() == want.tobytes() and gotE.tobytes() == wantE.tobytes()
            # ... and as differences of 4 to 16 bits (conga_sample_reads_packed): other exceptions, the same positions
            n_exc = []
            widths = (4, 5, 7, 8, 9, 10, 11, 12, 13, 15, 16, None)
            for width in widths:
                bits, w, ei2, ep2 = capi.encode_packed(pos[:int(off[-1])], off, width)
                assert w == (width or w) and len(bits) == ((int(off[-1]) + 7) // 8 * w if w != 16 else 2 * int(off[-1]))
                n_exc.append(len(ei2))
                ctx.sample_reads_packed(np.concatenate([bits, np.zeros(32, np.uint8)]), w, ei2, ep2, mapq, off)
                ctx.compute()
                got, gotE, _ = ctx.sample_fetch()
                assert got.tobytes() == want.tobytes() and gotE.tobytes() == wantE.tobytes(), width
            assert all(a >= b for a, b in zip(n_exc[:-2], n_exc[1:-1])) and n_exc[0] > n_exc[-2] == len(ei)   # (wider: fewer exceptions; 16 bits: conga_sample_reads_d16's)
            # the exceptions behind the differences in one buffer (one copy per sample)
            ctx.sample_reads_packed(capi.pack_inline(bits, ei2, ep2), w, len(ei2), None, mapq, off)
            ctx.compute()
            got, gotE, _ = ctx.sample_fetch()
            assert got.tobytes() == want.tobytes() and gotE.tobytes() == wantE.tobytes()
            for no_such_width in (3, 17):
                with pytest.raises(capi.CongaError):
                    ctx.sample_reads_packed(bits, no_such_width, ei2, ep2, mapq, off)
            check_against_oracle(oracle, chroms, reads, got, gotE, False)
    # unsorted input: the exception carries the position as it is, the engine's order check sees it
    if formulation == "tuple_space":
        reads = sample_reads_of(chroms, 33, 1.0)
        p = reads[0][0].copy()
        p[100], p[101] = p[101] + 5, p[100]
        reads[0] = (p, reads[0][1])
        with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
            open_layout(ctx, chroms, False)
            pos, mapq, off = pinned_sample(ctx, reads)
            delta, ei, ep = capi.encode_d16(pos[:int(off[-1])], off)
            ctx.sample_reads_d16(delta, ei, ep, mapq, off)
            ctx.compute()
            with pytest.raises(capi.CongaError) as err:
                ctx.sample_fetch()
            assert err.value.status == capi.CONGA_ERR_UNSORTED


# ---- conga_reads_bgzf called in process (the command-line tests run it in fresh subprocesses only) ---------------------
def bgzf_table(raw):
    """[(data_off, data_len, inflated_len, crc32)] of the non-empty blocks of a BGZF file + their inflated bytes."""
    import struct
    import zlib
    blocks, stream, at = [], bytearray(), 0
    while at < len(raw):
        xlen = struct.unpack_from("<H", raw, at + 10)[0]
        assert raw[at + 12:at + 16] == b"BC\x02\x00" and xlen == 6
        bsize = struct.unpack_from("<H", raw, at + 16)[0] + 1
        data_off, data_len = at + 18, bsize - 18 - 8
        crc, isize = struct.unpack_from("<II", raw, at + bsize - 8)
        if isize:
            blocks.append((data_off, data_len, isize, crc))
            stream += zlib.decompress(bytes(raw[data_off:data_off + data_len]), -15)
        at += bsize
    return blocks, bytes(stream)


@pytest.mark.parametrize("strategy", ["fixed", "default"])
def test_reads_bgzf_in_process_on_recycled_device_memory(capi, tmp_path, strategy, formulation):
    """The decoders' scratch is raw device memory: fixed-Huffman blocks (BTYPE 1) must decode right even when the
    allocator hands back memory full of 0xFF (the `fixed_ready` flag was read uninitialised)."""
    import ctypes
    import struct
    import zlib
    from conga_amd import formats
    # the HIP runtime the library itself is linked against (the one already mapped into this process; torch ships another):
    # blocks freed through it are what the library's next hipMalloc gets
    capi.load()
    paths = sorted({line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line},
                   key=lambda q: ("/opt/rocm" not in q, q))
    assert paths, "libconga_hip.so is loaded, so a HIP runtime must be"
    hip = ctypes.CDLL(paths[0])
    junk = []
    for n in (1 << 20, 3 << 20, 17 << 20, 64 << 20, 2 << 20):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)) == 0
        assert hip.hipMemset(p, 0xFF, ctypes.c_size_t(n)) == 0
        junk.append(p)
    assert hip.hipDeviceSynchronize() == 0
    for p in junk:
        assert hip.hipFree(p) == 0   # back to the allocator, contents intact

    cs = [synth.make_chrom(n, L, cov=2.0, n_dels=nd, gaps=False) for n, L, nd in (("1", 300_000, 15), ("2", 200_000, 10))]
    path = str(tmp_path / "r.bam")
    formats.write_bam(path, "S", [(c.name, c.length, c.pos, c.mapq) for c in cs], index=True, block_payload=20_000,
                      strategy=zlib.Z_FIXED if strategy == "fixed" else 0, unplaced=2)
    raw = np.fromfile(path, np.uint8)
    blocks, stream = bgzf_table(raw.tobytes())
    # the first record lies behind the header: magic, l_text, text, n_ref, then (l_name, name, l_ref) per reference
    l_text = struct.unpack_from("<i", stream, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, at)[0]
    at += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", stream, at)[0]
        at += 4 + l_name + 4
    segments = [(at, 0, cs[0].length, 0, 0), (at, 0, cs[1].length, 1, 1)]
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        for c in cs:
            ds, de = synth.kept_sorted(c.del_start, c.del_end)
            ctx.chrom_begin(c.length, c.gc)
            ctx.intervals("D", ds, de)
        per = ctx.reads_bgzf(raw, blocks, segments)
        assert per == [len(c.pos) for c in cs]
        ctx.compute()
        got = ctx.fetch_all()
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        for c in cs:
            ds, de = synth.kept_sorted(c.del_start, c.del_end)
            ctx.chrom_begin(c.length, c.gc)
            ctx.reads(c.pos, c.mapq)
            ctx.intervals("D", ds, de)
        ctx.compute()
        want = ctx.fetch_all()
    for (gd, _gu, gE, gs), (wd, _wu, wE, ws) in zip(got, want):
        assert gd.tobytes() == wd.tobytes() and gE.tobytes() == wE.tobytes() and gs.reads_counted == ws.reads_counted


def test_release_staging_between_two_overlapped_uploads(capi, tmp_path, monkeypatch):
    """conga_release_staging gives the pinned ring of the overlapped upload back; the next conga_reads_bgzf makes it again and
    decodes the same reads (the conga executable releases it beside the compute of its last sample)."""
    import struct
    from conga_amd import formats
    monkeypatch.setenv("CONGA_BGZF_OVERLAP", "1")
    monkeypatch.setenv("CONGA_BGZF_PIECE_KB", "8")
    cs = [synth.make_chrom(n, L, cov=2.0, n_dels=nd, gaps=False) for n, L, nd in (("1", 300_000, 15), ("2", 200_000, 10))]
    path = str(tmp_path / "r.bam")
    formats.write_bam(path, "S", [(c.name, c.length, c.pos, c.mapq) for c in cs], index=True, block_payload=20_000, unplaced=2)
    raw = np.fromfile(path, np.uint8)
    blocks, stream = bgzf_table(raw.tobytes())
    l_text = struct.unpack_from("<i", stream, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", stream, at)[0]
    at += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", stream, at)[0]
        at += 4 + l_name + 4
    segments = [(at, 0, cs[0].length, 0, 0), (at, 0, cs[1].length, 1, 1)]
    results = []
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        ctx.release_staging()   # nothing to give back yet: fine
        for c in cs:
            ds, de = synth.kept_sorted(c.del_start, c.del_end)
            ctx.chrom_begin(c.length, c.gc)
            ctx.intervals("D", ds, de)
        for _ in range(2):
            assert ctx.reads_bgzf(raw, blocks, segments) == [len(c.pos) for c in cs]
            ctx.release_staging()
            ctx.release_staging()
            ctx.compute()
            results.append(ctx.fetch_all())
            ctx.sample_begin()
    for (ad, _au, aE, ast), (bd, _bu, bE, bst) in zip(*results):
        assert ad.tobytes() == bd.tobytes() and aE.tobytes() == bE.tobytes() and ast.reads_counted == bst.reads_counted
        assert ast.reads_counted > 0


@pytest.mark.timeout(300, method="thread")
@pytest.mark.parametrize("named", ["between_calls_own_table", "between_calls_engine_table", "no_table_at_all"])
def test_bytes_named_ahead_with_the_callers_own_table_are_taken_up(capi, tmp_path, monkeypatch, capfd, named):
    """conga_reads_bgzf_next_fd + conga_reads_bgzf_next_blocks (include/conga_hip.h: "a caller that has read the table by itself hands
    it over ... to the same effect"): the bytes of the NEXT sample named from their descriptor alone, the block table brought by the
    caller, then the conga_reads_bgzf_fd call that takes them up -- it must return (round 3's last commit left a job whose table the
    caller brought out of the set that hands the spare output buffers on: its inflating thread slept for ever and the call with it,
    ADVICE round 3), find the stream inflated ahead, and give the reads of a call that was never named."""
    import os
    import struct
    from conga_amd import formats
    monkeypatch.setenv("CONGA_BGZF_OVERLAP", "1")
    monkeypatch.setenv("CONGA_BGZF_PIECE_KB", "8")
    monkeypatch.setenv("CONGA_TIMING", "1")
    cs = [synth.make_chrom(n, L, cov=2.0, n_dels=nd, gaps=False) for n, L, nd in (("1", 300_000, 15), ("2", 200_000, 10))]
    files = []
    for k in range(2):   # two samples on one layout
        rng = np.random.default_rng([k, 91])
        reads = [synth.make_reads(c.length, c.gc, c.step, 1.0 + k, 100, rng) for c in cs]
        path = str(tmp_path / ("s%d.bam" % k))
        formats.write_bam(path, "S%d" % k, [(c.name, c.length, p, m) for c, (p, m) in zip(cs, reads)], index=True, block_payload=20_000, unplaced=2)
        raw = np.fromfile(path, np.uint8)
        blocks, stream = bgzf_table(raw.tobytes())
        l_text = struct.unpack_from("<i", stream, 4)[0]
        at = 8 + l_text
        n_ref = struct.unpack_from("<i", stream, at)[0]
        at += 4
        for _ in range(n_ref):
            l_name = struct.unpack_from("<i", stream, at)[0]
            at += 4 + l_name + 4
        segments = [(at, 0, cs[0].length, 0, 0), (at, 0, cs[1].length, 1, 1)]
        files.append((path, len(raw), blocks, segments, [len(p) for p, _m in reads]))
    fds = [os.open(f[0], os.O_RDONLY) for f in files]
    try:
        with capi.Context(device=0, flags=capi.FLAG_BATCH | capi.FLAG_EXPECT_BGZF) as ctx:
            for c in cs:
                ds, de = synth.kept_sorted(c.del_start, c.del_end)
                ctx.chrom_begin(c.length, c.gc)
                ctx.intervals("D", ds, de)
            _p, size, blocks, segments, want = files[0]
            assert ctx.reads_bgzf_fd(fds[0], 0, size, blocks, segments) == want
            ctx.compute()
            first = ctx.fetch_all()
            # the next sample's bytes, named between two calls
            _p, size, blocks, segments, want = files[1]
            known = [b[0] - 18 for b in blocks[::3]] if named == "between_calls_engine_table" else []
            ticket = ctx.reads_bgzf_next_fd(fds[1], 0, size, known)
            assert ticket != 0
            if named == "between_calls_own_table":
                ctx.reads_bgzf_next_blocks(ticket, blocks)
            ctx.sample_begin()
            assert ctx.reads_bgzf_fd(fds[1], 0, size, blocks, segments) == want      # <- stood still for ever before the fix
            ctx.compute()
            second = ctx.fetch_all()
            # and the same sample through a call that was never named: the same records
            ctx.sample_begin()
            assert ctx.reads_bgzf_fd(fds[1], 0, size, blocks, segments) == want
            ctx.compute()
            again = ctx.fetch_all()
    finally:
        for fd in fds:
            os.close(fd)
    err = capfd.readouterr().err
    if named == "between_calls_own_table":
        # the stream WAS inflated ahead.  (Bytes named between two calls with known starts only go up when the next call begins, and
        # that call launches their inflates itself: it takes the job up before the job's own thread could.)
        assert "named ahead with its block table" in err, err[-2000:]
    for (ad, _au, aE, ast), (bd, _bu, bE, bst) in zip(second, again):
        assert ad.tobytes() == bd.tobytes() and aE.tobytes() == bE.tobytes() and ast.reads_counted == bst.reads_counted > 0
    assert first[0][0].tobytes() != second[0][0].tobytes()
