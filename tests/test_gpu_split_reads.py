"""Split-read evidence (--rp with --dups; SURVEY.md section 8 rows a15-a18, BASELINE configs[4]) through the C-ABI
against the oracle restatement of split_read.c / bam_data.c:29-154 / likelihood.c:41-94."""
import numpy as np
import pytest

from conga_amd import synth

pytestmark = pytest.mark.gpu

CODE = {ord("A"): 1, ord("C"): 2, ord("G"): 4, ord("T"): 8, ord("N"): 15}


def make_case(seed=3, L=400_000, n_normal=9000):
    rng = np.random.default_rng(seed)
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    ref[50_000:50_400] = ord("N")                                   # N block: 10-mers across it are not indexed
    ref[250_000:311_000] = ord("A")                                 # one 10-mer 61,000 times: bucket dropped (>= 50000)
    ref[120_000:120_300] = ref[20_000:20_300]                       # a repeat within 100 kb: two mappings, mapq 30
    lower = ref.copy()
    lower[300:900] |= 0x20                                          # soft-masked bases: upper-cased on load
    dels = [(30_000, 34_000), (100_000, 103_000), (150_000, 151_500)]
    dups = [(200_000, 204_000), (330_000, 345_000)]
    reads = []  # (pos, bases uint8[], mapq, flag, qual)

    def add(pos, bases, mapq=60, flag=0, q=30):
        quals = np.full(len(bases), q, np.uint8) if np.isscalar(q) else np.asarray(q, np.uint8)
        reads.append((int(pos), np.asarray(bases, np.uint8), mapq, flag, quals))

    for (s, e) in dels[1:]:                                         # reads across a deletion junction
        for k in range(25, 80, 3):
            add(s - k, np.concatenate([ref[s - k:s], ref[e:e + 100 - k]]))
    for (s, e) in dups:                                             # reads across a tandem-duplication junction
        for k in range(30, 75, 4):
            add(e - k, np.concatenate([ref[e - k:e], ref[s:s + 100 - k]]))
    for _ in range(n_normal):                                       # ordinary reads, some noisy
        l = int(rng.choice([100, 100, 100, 101, 76, 70, 59, 60, 61, 151]))
        p = int(rng.integers(0, L - 200))
        b = ref[p:p + l].copy()
        if rng.random() < 0.3:
            idx = rng.integers(0, l, int(rng.integers(1, 5)))
            b[idx] = rng.choice(np.frombuffer(b"ACGTN", np.uint8), len(idx))
        mapq = int(rng.choice([60, 60, 60, 0, 17, 40]))
        flag = int(rng.choice([0, 0, 0, 0, 0x400, 0x100, 0x800, 0x200, 16]))
        # base qualities: one value per read, or anything per base (the half-read means are sequential float sums, the
        # second half starting from the first half's mean, and they gate the element against the mapq threshold)
        add(p, b, mapq, flag, int(rng.choice([30, 30, 12, 2])) if rng.random() < 0.5 else rng.integers(0, 61, l))
    add(0, ref[0:100])                                              # pos == 0 is skipped
    add(20_000, ref[20_000:20_100])                                 # second half sits in the repeat: 2 mappings
    add(20_100, ref[20_100:20_200])
    reads.sort(key=lambda r: r[0])
    pos = np.array([r[0] for r in reads], np.int32)
    lq = np.array([len(r[1]) for r in reads], np.int32)
    off = np.concatenate([[0], np.cumsum(lq)[:-1]]).astype(np.uint64)
    lut = np.full(256, 15, np.uint8)
    for k, v in CODE.items():
        lut[k] = v
    codes = lut[np.concatenate([r[1] for r in reads])]
    qual = np.concatenate([r[4] for r in reads])
    mapq = np.array([r[2] for r in reads], np.uint8)
    flag = np.array([r[3] for r in reads], np.uint16)
    sat_s = np.array([60_000, 380_000, 61_000], np.int32)          # unsorted, overlapping on purpose
    sat_e = np.array([62_000, 381_000, 64_000], np.int32)
    return dict(L=L, ref=bytes(ref), ref_lower=bytes(lower), dels=dels, dups=dups, pos=pos, mapq=mapq, flag=flag, lq=lq,
                off=off, codes=codes, qual=qual, sat_s=sat_s, sat_e=sat_e)


def run_both(capi, oracle, c, mq=-1, min_read_length=60):
    ds, de = np.array([d[0] for d in c["dels"]], np.int32), np.array([d[1] for d in c["dels"]], np.int32)
    us, ue = np.array([d[0] for d in c["dups"]], np.int32), np.array([d[1] for d in c["dups"]], np.int32)
    rows, counts = oracle.split_read_rows(c["ref"], c["sat_s"], c["sat_e"], c["pos"], c["mapq"], c["flag"], c["lq"], c["off"],
                                          c["codes"], c["qual"], mq, min_read_length)
    od, ou = oracle.make_svs(ds, de), oracle.make_svs(us, ue)
    oracle.count_read_pairs(rows, od, ou)
    gc = np.full((c["L"] + 99) // 100, 40, np.uint8)
    with capi.Context(device=0, mq_threshold=mq, min_read_length=min_read_length) as ctx:
        ctx.chrom_begin(c["L"], gc)
        ctx.reads(c["pos"], c["mapq"])
        ctx.reference(c["ref_lower"])
        ctx.satellites(c["sat_s"], c["sat_e"])
        ctx.split_reads(c["pos"], c["mapq"], c["flag"], c["lq"], c["codes"], c["qual"], c["off"])
        ctx.intervals("D", ds, de)
        ctx.intervals("E", us, ue)
        dels, dups, _, st = ctx.finish()
        again = ctx.finish()
    assert np.array_equal(again[0]["border_rp"], dels["border_rp"]) and np.array_equal(again[1]["rp"], dups["rp"])
    return (dels, dups, st), (od, ou, rows, counts)


@pytest.mark.parametrize("mq,min_len", [(-1, 60), (20, 60), (35, 75), (-1, 10)])
def test_split_read_support_matches_the_oracle(capi, oracle, mq, min_len):
    c = make_case()
    (dels, dups, st), (od, ou, rows, counts) = run_both(capi, oracle, c, mq, min_len)
    assert (st.split_elements, st.split_mappings, st.split_del_rows, st.split_dup_rows) == tuple(int(x) for x in counts)
    assert np.array_equal(dels["border_rp"], od["border_rp"]) and np.all(dels["rp"] == 0)
    assert np.array_equal(dups["rp"], ou["rp"]) and np.all(dups["border_rp"] == 0)
    if mq == -1 and min_len == 60:
        # the planted junctions are found: every deletion / duplication with spanning reads gets support
        assert od["border_rp"][1] > 5 and od["border_rp"][2] > 5 and od["border_rp"][0] == 0
        assert ou["rp"][0] > 5 and ou["rp"][1] > 5
        assert counts[0] > 5_000 and counts[1] > 3_000


def test_reverse_complement_mappings_and_long_reads(capi, oracle):
    """Inverted repeats -- a half read's reverse complement maps -- at the chromosome's very start, within and beyond the
    look-ahead, and reads of every length up to the 1 022 bases a half-read buffer holds (the comparison goes by 56 bases)."""
    rng = np.random.default_rng(77)
    L = 300_000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    comp = np.arange(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    for a, b, n in ((40_000, 0, 1500), (90_000, 120_000, 3000), (10_000, 200_000, 2000), (150_000, 150_700, 600), (260_000, 3, 90)):
        ref[b:b + n] = comp[ref[a:a + n][::-1]]
    reads = []
    for a, n in ((40_000, 1500), (90_000, 3000), (10_000, 2000), (150_000, 600), (260_000, 90), (0, 1500), (120_000, 3000)):
        for l in (60, 61, 100, 111, 112, 113, 151, 224, 225, 250, 400, 1021, 1022):
            for p in (a, a + 7, a + max(0, n - l), a + max(0, n - l) // 2):
                b = ref[p:p + l].copy()
                if len(b) == l and p > 0:
                    if rng.random() < 0.5:
                        b[rng.integers(0, l, 2)] = rng.choice(np.frombuffer(b"ACGTN", np.uint8), 2)
                    reads.append((p, b))
    reads.sort(key=lambda r: r[0])
    lut = np.full(256, 15, np.uint8)
    for k, v in CODE.items():
        lut[k] = v
    lq = np.array([len(r[1]) for r in reads], np.int32)
    c = dict(L=L, ref=bytes(ref), ref_lower=bytes(ref), dels=[(50_000, 53_000)], dups=[(100_000, 104_000)],
             pos=np.array([r[0] for r in reads], np.int32), mapq=np.full(len(reads), 60, np.uint8), flag=np.zeros(len(reads), np.uint16), lq=lq,
             off=np.concatenate([[0], np.cumsum(lq)[:-1]]).astype(np.uint64), codes=lut[np.concatenate([r[1] for r in reads])],
             qual=np.full(int(lq.sum()), 30, np.uint8), sat_s=np.zeros(0, np.int32), sat_e=np.zeros(0, np.int32))
    (dels, dups, st), (od, ou, rows, counts) = run_both(capi, oracle, c, -1, 50)
    assert (st.split_elements, st.split_mappings, st.split_del_rows, st.split_dup_rows) == tuple(int(x) for x in counts)
    assert np.array_equal(dels["border_rp"], od["border_rp"]) and np.array_equal(dups["rp"], ou["rp"])
    assert counts[1] > 1.5 * counts[0] > 0   # (forward at the read's own place and reverse in the inverted copy: about two mappings per element)


def test_reverse_complement_hits_at_every_offset_of_the_chromosomes_first_bases(capi, oracle):
    """Round 3's last-day fault, as a case of its own (profiles/r03l_split_map_fault_seed82_case11.log, commit 4000c64): halves of
    57 .. 111 bases whose reverse complement maps at c = 0 .. 63 of the reference text -- the last 56-base step of the wide
    comparison would load from in front of the text there, so those windows must go base by base (split_geom.h: sr_rev_wide_ok;
    tests/test_split_geom.py walks the arithmetic on the host).  An inverted copy of [a, a + 400) lies on the chromosome's first
    400 bases; read (n, c) is 2 n bases long and placed so that its second half's reverse complement begins at base c."""
    rng = np.random.default_rng(82)
    L, a, N = 200_000, 30_000, 400
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    comp = np.arange(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    ref[0:N] = comp[ref[a:a + N][::-1]]
    reads = []
    for n in range(57, 112):
        for c in range(0, 64):
            p = a + N - 2 * n - c
            b = ref[p:p + 2 * n].copy()
            if (n + c) % 3 == 0:                                    # some with a mismatch or two (dist_max = 5 % of the half)
                b[rng.integers(0, 2 * n, 2)] = rng.choice(np.frombuffer(b"ACGTN", np.uint8), 2)
            reads.append((p, b))
    reads.sort(key=lambda r: r[0])
    lut = np.full(256, 15, np.uint8)
    for k, v in CODE.items():
        lut[k] = v
    lq = np.array([len(r[1]) for r in reads], np.int32)
    c = dict(L=L, ref=bytes(ref), ref_lower=bytes(ref), dels=[(50_000, 53_000)], dups=[(100_000, 104_000)],
             pos=np.array([r[0] for r in reads], np.int32), mapq=np.full(len(reads), 60, np.uint8), flag=np.zeros(len(reads), np.uint16), lq=lq,
             off=np.concatenate([[0], np.cumsum(lq)[:-1]]).astype(np.uint64), codes=lut[np.concatenate([r[1] for r in reads])],
             qual=np.full(int(lq.sum()), 30, np.uint8), sat_s=np.zeros(0, np.int32), sat_e=np.zeros(0, np.int32))
    (dels, dups, st), (od, ou, rows, counts) = run_both(capi, oracle, c, -1, 50)
    assert (st.split_elements, st.split_mappings, st.split_del_rows, st.split_dup_rows) == tuple(int(x) for x in counts)
    assert np.array_equal(dels["border_rp"], od["border_rp"]) and np.array_equal(dups["rp"], ou["rp"])
    assert counts[0] == 2 * len(reads) and counts[1] > 1.9 * counts[0]   # both halves of (nearly) every read map twice: at home and in the inverted copy (oracle: 7 040 elements, 13 600 mappings)


def test_without_reference_or_reads_nothing_is_counted(capi):
    c = make_case(n_normal=200)
    ds, de = np.array([100_000], np.int32), np.array([103_000], np.int32)
    gc = np.full((c["L"] + 99) // 100, 40, np.uint8)
    with capi.Context(device=0) as ctx:
        ctx.chrom_begin(c["L"], gc)
        ctx.reads(c["pos"], c["mapq"])
        ctx.split_reads(c["pos"], c["mapq"], c["flag"], c["lq"], c["codes"], c["qual"], c["off"])
        ctx.intervals("D", ds, de)
        dels, _, _, st = ctx.finish()
        assert st.split_elements == 0 and np.all(dels["border_rp"] == 0)
        with pytest.raises(capi.CongaError):
            ctx.reference(b"ACGT")  # wrong length


def test_records_that_cannot_be_are_refused_or_ignored(capi):
    """What a caller must not be able to do to the device: a record block that lies outside the committed bytes (also one
    whose offset makes the sum wrap) is refused at commit; records with a negative position, a position behind the
    chromosome's end or an absurd length are carried and never looked up."""
    import ctypes as C
    c = make_case(n_normal=300)
    ds, de = np.array([100_000], np.int32), np.array([103_000], np.int32)
    gc = np.full((c["L"] + 99) // 100, 40, np.uint8)
    with capi.Context(device=0) as ctx:
        ctx.chrom_begin(c["L"], gc)
        ctx.reads(c["pos"], c["mapq"])
        ctx.reference(c["ref"])
        ctx.satellites(np.zeros(0, np.int32), np.zeros(0, np.int32))
        for off in (1 << 40, (1 << 64) - 40, 1000):
            stg = capi.SplitStaging()
            ctx._check(ctx._lib.conga_split_reads_staging(ctx._h, C.byref(stg)))
            stg.pos[0], stg.mapq[0], stg.flag[0], stg.l_qseq[0] = 1000, 60, 0, 100
            stg.data_off[0] = off
            assert ctx._lib.conga_split_reads_commit(ctx._h, 1, 150) != 0   # 100 bases need 150 bytes from `off` on
        pos = c["pos"].copy()
        pos[0::7] = -5
        pos[1::7] = c["L"] + 17
        pos[2::7] = 2_000_000_000
        ctx.split_reads(pos, c["mapq"], c["flag"], c["lq"], c["codes"], c["qual"], c["off"])
        ctx.intervals("D", ds, de)
        dels, _, _, st = ctx.finish()
        assert st.split_elements > 0


def test_batch_with_a_chromosome_without_reference_in_the_middle(capi, oracle):
    """Three chromosomes in one batch context, the middle one with records but no conga_reference(): it takes no part in the
    split-read launch, the other two get exactly what they get alone (their records need not follow each other)."""
    cases = [make_case(seed=11, n_normal=1500), make_case(seed=12, n_normal=800), make_case(seed=13, n_normal=1200)]
    alone = []
    for c in (cases[0], cases[2]):
        (dels, dups, st), _ = run_both(capi, oracle, c)
        alone.append((dels["border_rp"].copy(), dups["rp"].copy(), st.split_elements, st.split_mappings))
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        for k, c in enumerate(cases):
            gc = np.full((c["L"] + 99) // 100, 40, np.uint8)
            ctx.chrom_begin(c["L"], gc)
            ctx.reads(c["pos"], c["mapq"])
            if k != 1:
                ctx.reference(c["ref_lower"])
                ctx.satellites(c["sat_s"], c["sat_e"])
            ctx.split_reads(c["pos"], c["mapq"], c["flag"], c["lq"], c["codes"], c["qual"], c["off"])
            ctx.intervals("D", np.array([d[0] for d in c["dels"]], np.int32), np.array([d[1] for d in c["dels"]], np.int32))
            ctx.intervals("E", np.array([d[0] for d in c["dups"]], np.int32), np.array([d[1] for d in c["dups"]], np.int32))
        ctx.compute()
        res = ctx.fetch_all()
    for k, want in ((0, alone[0]), (2, alone[1])):
        dels, dups, _E, st = res[k]
        assert np.array_equal(dels["border_rp"], want[0]) and np.array_equal(dups["rp"], want[1])
        assert (st.split_elements, st.split_mappings) == want[2:]
    dels, dups, _E, st = res[1]
    assert st.split_elements == 0 and np.all(dels["border_rp"] == 0) and np.all(dups["rp"] == 0)


def test_next_sample_of_a_cohort_brings_its_records_the_indexes_stay(capi, oracle):
    """conga_sample_begin with split reads: the reference sequences and the 10-mer indexes are the layout's, the records the
    sample's.  Sample B behind sample A in one context equals sample B alone; A again equals A."""
    a = make_case(seed=21, n_normal=2500)
    b = make_case(seed=21, n_normal=2500)
    keep = np.arange(len(b["pos"])) % 4 != 2
    per_base = np.repeat(keep, b["lq"])
    lq = b["lq"][keep]
    b.update(pos=b["pos"][keep], mapq=b["mapq"][keep], flag=b["flag"][keep], lq=lq, codes=b["codes"][per_base], qual=b["qual"][per_base],
             off=np.concatenate([[0], np.cumsum(lq)[:-1]]).astype(np.uint64))
    want = {}
    for name, c in (("a", a), ("b", b)):
        (dels, dups, st), _ = run_both(capi, oracle, c)
        want[name] = (dels["border_rp"].copy(), dups["rp"].copy(), st.split_elements, st.split_mappings, st.split_del_rows, st.split_dup_rows)
    assert want["a"][2] != want["b"][2]
    ds, de = np.array([d[0] for d in a["dels"]], np.int32), np.array([d[1] for d in a["dels"]], np.int32)
    us, ue = np.array([d[0] for d in a["dups"]], np.int32), np.array([d[1] for d in a["dups"]], np.int32)
    gc = np.full((a["L"] + 99) // 100, 40, np.uint8)
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        ctx.chrom_begin(a["L"], gc)
        ctx.reference(a["ref_lower"])
        ctx.satellites(a["sat_s"], a["sat_e"])
        ctx.intervals("D", ds, de)
        ctx.intervals("E", us, ue)
        for name, c in (("a", a), ("b", b), ("a", a)):
            ctx._check(ctx._lib.conga_sample_begin(ctx._h))
            ctx.reads(c["pos"], c["mapq"])
            ctx.split_reads(c["pos"], c["mapq"], c["flag"], c["lq"], c["codes"], c["qual"], c["off"])
            ctx.compute()
            dels, dups, _E, st = ctx.fetch_all()[0]
            got = (dels["border_rp"], dups["rp"], st.split_elements, st.split_mappings, st.split_del_rows, st.split_dup_rows)
            assert np.array_equal(got[0], want[name][0]) and np.array_equal(got[1], want[name][1]) and got[2:] == want[name][2:], name
