"""Parity of the HIP path (through the C-ABI of include/conga_hip.h) with the CPU oracle.

Bar (BASELINE.json north_star): integer results bit-exact (read_depth, observed, CN, genotype strings,
GC histogram), expected_rd float32 bit-exact, log-likelihoods / c-score / mappability within 1e-6.
"""
import os

import numpy as np
import pytest

from conga_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["tuple_space", "dense"])
def formulation(request, capi):
    """Every case runs twice: in tuple space (the default: no read_depth[] in HBM) and with
    CONGA_FLAG_MATERIALIZE_DEPTH (the reference's dense formulation).  Results must not differ."""
    capi.EXTRA_FLAGS = capi.FLAG_MATERIALIZE_DEPTH if request.param == "dense" else 0
    yield request.param
    capi.EXTRA_FLAGS = 0

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ATOL = 1e-6  # north_star: "log-likelihoods agree within 1e-6"
RTOL_SCORE = 1e-6


def genotype_strings(rec, c_score=np.float32(0.5)):
    """likelihood.c:184-195 for dels."""
    out = []
    inv = np.float32(1) / c_score
    for r in rec:
        called = "1/1" if r["cn"] == 2 else "0/1"
        if r["score"] < c_score:
            out.append(called)
        elif r["score"] <= inv:
            out.append("N/A")
        else:
            out.append("0/0")
    return out


def assert_records(got, want, has_map=True):
    assert len(got) == len(want)
    assert np.array_equal(got["observed"], want["observed"])
    assert np.array_equal(got["expected"].view(np.uint32), want["expected"].view(np.uint32)), "expected_rd bits"
    assert np.array_equal(got["cn"], want["cn"])
    for k in ("lhomo", "lhetero", "lnone", "score"):
        g, w = got[k], want[k]
        both_nan = np.isnan(g) & np.isnan(w)  # log of a negative lambda (wrapped depth counter only)
        # the three log-likelihoods: absolute 1e-6.  The c-score is trunc_max / lnone, so a log-likelihood
        # error of 1e-8 is amplified by 1/|lnone|: it gets 1e-6 absolute + 1e-6 relative (it is printed %.2f).
        tol = ATOL + (RTOL_SCORE * np.abs(w) if k == "score" else 0.0)
        with np.errstate(invalid="ignore"):
            assert np.all(both_nan | (np.abs(g - w) <= tol) | (g == w)), k
    ok = ~np.isnan(want["score"])
    assert np.array_equal(np.signbit(got["score"])[ok], np.signbit(want["score"])[ok])
    if has_map:
        # (0 / 0 for an empty interval, which only the C-ABI can be handed: the command line keeps >= min-sv-size rows)
        assert np.allclose(got["mappability"], want["mappability"], rtol=0, atol=ATOL, equal_nan=True)
    assert genotype_strings(got) == genotype_strings(want)


def run_oracle(O, length, gc, pos, mapq, ds, de, us, ue, mq=-1, step=100, rows=None, gc_like=None):
    rd, counted = O.count_reads(length, pos, mapq, mq)
    E, S, W = O.calc_mean_per_chr(rd, gc, step)
    m = O.paint_mappability(length, *rows) if rows is not None else None
    gl = gc if gc_like is None else gc_like
    dels = O.find_depths(rd, m, gl, E, "D", O.make_svs(ds, de), step)
    dups = O.find_depths(rd, m, gl, E, "E", O.make_svs(us, ue), step)
    return dict(rd=rd, counted=counted, E=E, S=S, W=W, dels=dels, dups=dups, map=m)


def run_gpu(capi, length, gc, pos, mapq, ds, de, us, ue, mq=-1, step=100, rows=None, gc_like=None, flags=0,
            want_tracks=True):
    with capi.Context(device=0, mq_threshold=mq, gc_step=step, flags=flags) as ctx:
        ctx.chrom_begin(length, gc, gc_like)
        ctx.reads(pos, mapq)
        if rows is not None:
            ctx.mappability(*rows)
        ctx.intervals("D", ds, de)
        ctx.intervals("E", us, ue)
        dels, dups, E, st = ctx.finish()
        out = dict(dels=dels, dups=dups, E=E, counted=st.reads_counted, oor=st.reads_out_of_range,
                   S=np.array(st.rd_per_gc[:]), W=np.array(st.window_per_gc[:]), mean=st.mean, rd_sum=st.rd_sum,
                   dense=bool(st.depth_materialized))
        if want_tracks:
            out["rd"] = ctx.read_depth()
            if rows is not None and (len(ds) + len(us)) > 0:
                out["map"] = ctx.mappability_track()
    return out


def compare(got, want, has_map):
    if "rd" in got:
        assert np.array_equal(got["rd"], want["rd"]), "read_depth"
    assert got["counted"] == want["counted"]
    assert np.array_equal(got["S"], want["S"])
    assert np.array_equal(got["W"], want["W"])
    assert np.array_equal(got["E"].view(np.uint32), want["E"].view(np.uint32)), "expected_read_depth bits"
    if has_map and "map" in got:
        assert np.array_equal(got["map"], want["map"]), "mappability track"
    assert_records(got["dels"], want["dels"], has_map)
    assert_records(got["dups"], want["dups"], has_map)


def chrom_case(name, length, **kw):
    c = synth.make_chrom(name, length, **kw)
    ds, de = synth.kept_sorted(c.del_start, c.del_end)
    us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
    return c, ds, de, us, ue


# ---------------------------------------------------------------------------------------------
def test_golden_fixture(capi):
    z = np.load(os.path.join(GOLDEN, "small_chr.npz"))
    dels_w, dups_w = z["dels"], z["dups"]
    got = run_gpu(capi, int(z["length"]), z["gc"], z["pos"], z["mapq"], dels_w["start"], dels_w["end"],
                  dups_w["start"], dups_w["end"], rows=(z["map_start"], z["map_end"], z["map_val"]))
    rd = np.zeros(int(z["length"]), np.int16)
    rd[z["rd_nonzero_idx"]] = z["rd_nonzero_val"]
    assert np.array_equal(got["rd"], rd)
    assert got["counted"] == int(z["counted"])
    assert np.array_equal(got["E"].view(np.uint32), z["E"].view(np.uint32))
    assert np.array_equal(got["S"], z["S"]) and np.array_equal(got["W"], z["W"])
    assert_records(got["dels"], dels_w)
    assert_records(got["dups"], dups_w)


@pytest.mark.parametrize("name,length,kw,mq,with_map", [
    ("3", 1_000_000, dict(cov=1.0, n_dels=60, n_dups=15), -1, False),
    ("4", 1_200_000, dict(cov=1.0, n_dels=60, n_dups=15, mappability=True), -1, True),
    ("5", 800_000, dict(cov=5.0, n_dels=40, n_dups=10), 20, False),
    ("6", 3_000_017, dict(cov=30.0, n_dels=50, n_dups=12, gaps=True), 0, False),
    ("8", 99_999, dict(cov=0.5, n_dels=10, n_dups=3), 59, False),
])
def test_random_chromosomes(capi, oracle, formulation, name, length, kw, mq, with_map):
    c, ds, de, us, ue = chrom_case(name, length, **kw)
    rows = (c.map_start, c.map_end, c.map_val) if with_map else None
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, mq=mq, rows=rows)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, mq=mq, rows=rows)
    compare(got, want, with_map)
    assert got["dense"] == (formulation == "dense")


@pytest.mark.parametrize("step", [1, 7, 64, 100, 128, 1000])
def test_other_gc_steps(capi, oracle, step):
    c, ds, de, us, ue = chrom_case("9", 400_000, cov=2.0, n_dels=25, n_dups=6, step=step, gaps=False)
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, step=step)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, step=step)
    compare(got, want, False)


def test_distinct_gc_arrays_for_histogram_and_likelihood(capi, oracle):
    """The boundary keeps loop B's and loop C's GC lookups separate (read_distribution.c:70 vs likelihood.c:117)."""
    c, ds, de, us, ue = chrom_case("10", 500_000, cov=2.0, n_dels=30, n_dups=8, gaps=False)
    gc_like = np.roll(c.gc, 1)
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, gc_like=gc_like)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, gc_like=gc_like)
    compare(got, want, False)


def test_edge_intervals(capi, oracle):
    """Chromosome ends, window-boundary starts, l = 1000 exactly, duplicates, nested, whole chromosome,
    an interval running past the chromosome end (reads/maps nothing there)."""
    c, _, _, _, _ = chrom_case("11", 700_123, cov=1.0, gaps=False)
    L = c.length
    s = np.array([0, 0, 100, 199, 200, 300_000, 300_000, 300_000, 300_050, L - 1000, L - 1500, 650_000, 0, 5], np.int32)
    e = np.array([1000, L, 1100, 1199, 1300, 301_000, 301_000, 400_000, 300_950 + 100, L, L + 777, 650_000 + 1000, L + 5, 1005], np.int32)
    ds, de = synth.kept_sorted(s, e)
    want = run_oracle(oracle, L, c.gc, c.pos, c.mapq, ds, de, ds[:5], de[:5])
    got = run_gpu(capi, L, c.gc, c.pos, c.mapq, ds, de, ds[:5], de[:5])
    compare(got, want, False)


def test_no_reads_and_no_intervals(capi, oracle):
    c, ds, de, us, ue = chrom_case("12", 250_000, cov=1.0, n_dels=12, n_dups=3, gaps=False)
    empty_i, empty_b = np.zeros(0, np.int32), np.zeros(0, np.uint8)
    want = run_oracle(oracle, c.length, c.gc, empty_i, empty_b, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, empty_i, empty_b, ds, de, us, ue)
    compare(got, want, False)
    assert np.all(got["dels"]["observed"] == 0) and np.all(np.signbit(got["dels"]["score"]))
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, empty_i, empty_i, empty_i, empty_i)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, empty_i, empty_i, empty_i, empty_i)
    compare(got, want, False)
    assert got["rd_sum"] == int(want["rd"].astype(np.int64).sum())


def test_all_gc_zero_gives_zero_expectation(capi, oracle):
    c, ds, de, us, ue = chrom_case("13", 200_000, cov=1.0, n_dels=12, n_dups=3, gaps=False)
    gc = np.zeros_like(c.gc)
    want = run_oracle(oracle, c.length, gc, c.pos, c.mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, gc, c.pos, c.mapq, ds, de, us, ue)
    compare(got, want, False)
    assert np.all(got["E"] == 0) and np.all(got["dels"]["expected"] == 0)


def test_pileup_wraps_like_a_short(capi, oracle):
    """70000 reads starting on one base: read_depth is a `short` (common.h:91) and wraps."""
    c, ds, de, us, ue = chrom_case("14", 120_000, cov=1.0, n_dels=10, n_dups=2, gaps=False)
    pile = np.full(70_000, 60_001, np.int32)
    pos = np.sort(np.concatenate([c.pos, pile, np.full(40_000, 60_002, np.int32)]), kind="stable")
    mapq = np.full(len(pos), 60, np.uint8)
    ds = np.concatenate([ds, [59_000]]).astype(np.int32)
    de = np.concatenate([de, [61_500]]).astype(np.int32)
    ds, de = synth.kept_sorted(ds, de)
    want = run_oracle(oracle, c.length, c.gc, pos, mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, pos, mapq, ds, de, us, ue)
    assert want["rd"][60_001] == np.int16(70_000 + int((c.pos == 60_001).sum()) - 65536)
    compare(got, want, False)
    assert got["dense"], "a possible `short` wrap must send the batch to the dense kernels"


def test_pileup_straddling_two_commits_is_still_detected(capi, oracle):
    """33000 reads on one base, 14304 of them at the end of the first 4M-tuple commit and the rest in the next:
    neither part alone looks like a wrap, the run carried across the commit does."""
    L = 6_000_000
    rng = np.random.default_rng(77)
    n_before = (1 << 22) - 14_304
    before = np.sort(rng.integers(0, 3_000_000, n_before)).astype(np.int32)
    after = np.sort(rng.integers(3_000_001, L, 500_000)).astype(np.int32)
    pos = np.concatenate([before, np.full(33_000, 3_000_000, np.int32), after])
    mapq = np.full(len(pos), 60, np.uint8)
    gc = synth.make_gc_track(L, rng, gaps=False)
    ds = np.array([2_990_000, 10_000, 2_999_999], np.int32)
    de = np.array([3_010_000, 2_000_000, 3_000_002], np.int32)
    ds, de = synth.kept_sorted(ds, de, min_sv_size=0)
    us, ue = np.zeros(0, np.int32), np.zeros(0, np.int32)
    want = run_oracle(oracle, L, gc, pos, mapq, ds, de, us, ue)
    got = run_gpu(capi, L, gc, pos, mapq, ds, de, us, ue)
    assert want["rd"][3_000_000] == np.int16(33_000 - 65536)
    compare(got, want, False)
    assert got["dense"]


def test_deep_pileup_below_the_wrap_stays_exact(capi, oracle, formulation):
    """30000 reads on one base (no wrap, below the detector's threshold): both formulations agree with the oracle."""
    c, ds, de, us, ue = chrom_case("14", 120_000, cov=2.0, n_dels=10, n_dups=2, gaps=False)
    pos = np.sort(np.concatenate([c.pos, np.full(30_000, 70_001, np.int32)]), kind="stable")
    mapq = np.full(len(pos), 60, np.uint8)
    ds = np.concatenate([ds, [69_000, 70_001, 70_002]]).astype(np.int32)
    de = np.concatenate([de, [71_500, 70_002, 70_500]]).astype(np.int32)
    ds, de = synth.kept_sorted(ds, de, min_sv_size=0)
    want = run_oracle(oracle, c.length, c.gc, pos, mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, pos, mapq, ds, de, us, ue)
    compare(got, want, False)
    assert got["dense"] == (formulation == "dense")


def test_out_of_range_reads_are_skipped_and_counted(capi, oracle):
    c, ds, de, us, ue = chrom_case("15", 150_000, cov=1.0, n_dels=10, n_dups=2, gaps=False)
    pos = np.concatenate([[-5, -1], c.pos, [c.length, c.length + 10, 2_000_000_000]]).astype(np.int32)
    mapq = np.concatenate([[60, 60], c.mapq, [60, 60, 60]]).astype(np.uint8)
    want = run_oracle(oracle, c.length, c.gc, pos, mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, pos, mapq, ds, de, us, ue)
    compare(got, want, False)
    assert got["oor"] == 5


def test_gc_above_100_is_refused(capi):
    """The reference indexes 101-entry tables with the rounded GC% (read_distribution.c:71-72, likelihood.c:118)."""
    gc = np.full(10, 40, np.uint8)
    gc[3] = 101
    with capi.Context(device=0) as ctx:
        for args in ((gc,), (np.full(10, 40, np.uint8), gc)):
            with pytest.raises(capi.CongaError) as err:
                ctx.chrom_begin(1000, *args)
            assert err.value.status == capi.CONGA_ERR_RANGE
        ctx.chrom_begin(1000, np.full(10, 100, np.uint8))


def test_unsorted_reads_fail_loudly_or_take_the_atomic_path(capi, oracle):
    c, ds, de, us, ue = chrom_case("16", 300_000, cov=3.0, n_dels=15, n_dups=4, gaps=False)
    perm = np.random.default_rng(3).permutation(len(c.pos))
    pos, mapq = c.pos[perm], c.mapq[perm]
    with pytest.raises(capi.CongaError) as err:
        run_gpu(capi, c.length, c.gc, pos, mapq, ds, de, us, ue, mq=10)
    assert err.value.status == capi.CONGA_ERR_UNSORTED
    want = run_oracle(oracle, c.length, c.gc, pos, mapq, ds, de, us, ue, mq=10)
    got = run_gpu(capi, c.length, c.gc, pos, mapq, ds, de, us, ue, mq=10, flags=capi.FLAG_READS_UNSORTED)
    compare(got, want, False)
    assert got["dense"]


def test_streaming_commits_larger_than_the_staging_ring(capi, oracle):
    """> 2 x 4M tuples: the pinned ring is reused and the HBM tuple buffer grows."""
    c, ds, de, us, ue = chrom_case("17", 30_000_000, cov=50.0, n_dels=300, n_dups=60, gaps=True)
    assert len(c.pos) > 9_000_000
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue)
    compare(got, want, False)


def test_mappability_general_row_order(capi, oracle):
    """Overlapping, nested, unsorted, empty (end < start) and out-of-range rows: file order decides (svs.c:363-371)."""
    c, ds, de, us, ue = chrom_case("18", 200_000, cov=1.0, n_dels=20, n_dups=5, gaps=False)
    rng = np.random.default_rng(9)
    n = 3000
    s = rng.integers(0, c.length, n).astype(np.int32)
    e = (s + rng.integers(-20, 900, n)).astype(np.int32)
    e[-1] = c.length + 500
    v = rng.choice(np.array([1, 0.5, 0.333333, 0.25, 0.1], np.float32), n)
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, rows=(s, e, v))
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, rows=(s, e, v))
    compare(got, want, True)


def test_mappability_is_exact_for_kmer_track_values(capi, oracle):
    """With the usual few-bit values every partial sum is exact, so the parallel double sum equals the
    reference's serial one bit for bit."""
    c, ds, de, us, ue = chrom_case("19", 2_000_000, cov=1.0, n_dels=80, n_dups=20, gaps=False)
    rng = np.random.default_rng(2)
    ms, me, _ = synth.make_mappability(c.length, rng)
    mv = rng.choice(np.array([1, 0.5, 0.25, 0.125, 0.75], np.float32), len(ms))
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, rows=(ms, me, mv))
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, rows=(ms, me, mv))
    compare(got, want, True)
    assert np.array_equal(got["dels"]["mappability"], want["dels"]["mappability"])
    assert np.array_equal(got["dups"]["mappability"], want["dups"]["mappability"])


def test_mappability_sorted_rows_edge_cases(capi, oracle):
    """Rows that qualify for the row-space sum (sorted, next start >= previous end) in their awkward forms: abutting
    rows that share their endpoint (the later row owns the shared base: svs.c:368 paints end inclusive, in file
    order), single-base rows, repeated starts, gaps, a row that begins before base 0 and one that ends past the
    chromosome.  Intervals start / end on those very bases.  Exact values, so the sums must be bit-identical."""
    c, _, _, _, _ = chrom_case("19", 300_000, cov=1.0, n_dels=5, n_dups=2, gaps=False)
    L = c.length
    rows = [(-50, 10, 0.5), (10, 10, 0.25), (10, 10, 0.75), (10, 500, 1.0), (500, 900, 0.5), (900, 900, 0.125),
            (1_200, 1_200, 0.25), (1_200, 40_000, 1.0), (40_000, 40_001, 0.5), (40_001, 150_000, 0.75),
            (150_000, 150_000, 0.125), (200_000, 250_000, 0.5), (250_000, L + 400, 0.25)]
    ms = np.array([r[0] for r in rows], np.int32)
    me = np.array([r[1] for r in rows], np.int32)
    mv = np.array([r[2] for r in rows], np.float32)
    s = np.array([0, 9, 10, 11, 400, 899, 900, 1_000, 1_199, 39_000, 40_000, 40_001, 149_000, 150_000, 199_000,
                  249_000, L - 1_500, 5], np.int32)
    e = np.array([1_500, 1_010, 1_011, 2_000, 1_400, 1_900, 1_901, 2_500, 41_000, 41_000, 41_001, 42_000, 151_000,
                  151_001, 251_000, L, L, L], np.int32)
    ds, de = synth.kept_sorted(s, e)
    us, ue = ds[::3], de[::3]
    want = run_oracle(oracle, L, c.gc, c.pos, c.mapq, ds, de, us, ue, rows=(ms, me, mv))
    got = run_gpu(capi, L, c.gc, c.pos, c.mapq, ds, de, us, ue, rows=(ms, me, mv))
    compare(got, want, True)
    assert np.array_equal(got["dels"]["mappability"], want["dels"]["mappability"])
    assert np.array_equal(got["dups"]["mappability"], want["dups"]["mappability"])


def test_replay_is_idempotent_and_split_support_is_copied_through(capi, oracle):
    c, ds, de, us, ue = chrom_case("20", 600_000, cov=1.0, n_dels=30, n_dups=8, gaps=False)
    with capi.Context(device=0, flags=capi.FLAG_PROFILE) as ctx:
        ctx.chrom_begin(c.length, c.gc)
        ctx.reads(c.pos, c.mapq)
        ctx.intervals("D", ds, de)
        ctx.intervals("E", us, ue)
        ctx.split_support("D", np.arange(len(ds)))
        ctx.split_support("E", np.arange(len(us)) * 3)
        first = ctx.finish()
        ctx.compute()
        ctx.compute()
        second = ctx.fetch()
        assert first[0].tobytes() == second[0].tobytes() and first[1].tobytes() == second[1].tobytes()
        assert np.array_equal(first[0]["border_rp"], np.arange(len(ds))) and np.all(first[0]["rp"] == 0)
        assert np.array_equal(first[1]["rp"], np.arange(len(us)) * 3) and np.all(first[1]["border_rp"] == 0)
        k = second[3].kernel_ms
        if capi.EXTRA_FLAGS & capi.FLAG_MATERIALIZE_DEPTH:
            assert k[1] > 0 and k[4] > 0 and k[7] == 0  # depth_tile and interval_reduce were timed
        else:
            assert k[0] > 0 and k[7] > 0 and k[1] == 0  # ingest (tuples) and interval_count were timed
        # a second chromosome on the same context starts from clean state
        ctx.chrom_begin(c.length, c.gc)
        ctx.reads(c.pos[:1000], c.mapq[:1000])
        d2 = ctx.finish()
        assert len(d2[0]) == 0 and d2[3].reads_counted == 1000


def test_score_kernel_known_answers(capi):
    """SURVEY.md App. D through the device: one interval whose depths are forced to obs=519 is not
    constructible directly, so check lpoisson's pieces via E=0 and obs=0 paths instead."""
    L = 10_000
    gc = np.full(100, 40, np.uint8)
    got = run_gpu(capi, L, gc, np.zeros(0, np.int32), np.zeros(0, np.uint8), [1000], [3000], [1000], [3000])
    d, u = got["dels"][0], got["dups"][0]
    assert d["observed"] == 0 and d["expected"] == 0
    assert d["lhomo"] == d["lhetero"] == d["lnone"] == -0.01     # lpoisson(0, 0.0) (KAT)
    assert d["score"] == 0 and np.signbit(d["score"]) and d["cn"] == 1
    assert u["lhomo"] == -0.01 and u["cn"] == 1


# ------------------------------------------------------------------------------------------------
# BASELINE.json sizes
# ------------------------------------------------------------------------------------------------
def test_config0_chr21_full_size_against_oracle(capi, oracle):
    """configs[0]: chr21 only, ~2k deletions, 0.5x."""
    c, ds, de, us, ue = chrom_case("21", 48_129_895, cov=0.5, n_dels=2000, n_dups=0)
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue)
    compare(got, want, False)


def test_chr1_size_properties(capi):
    """Largest chromosome of configs[1] (L = 249,250,621, 1x): size-independent properties only."""
    c, ds, de, us, ue = chrom_case("1", 249_250_621, cov=1.0, n_dels=3634, n_dups=519)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, want_tracks=False)
    assert got["counted"] == len(c.pos) and got["oor"] == 0
    assert got["rd_sum"] == len(c.pos)                       # checksum of the depth track
    assert int(got["W"].sum()) == c.length                   # every base lands in exactly one GC bin
    # observed == number of read starts inside the interval (searchsorted on the sorted tuples)
    for rec, s, e in ((got["dels"], ds, de), (got["dups"], us, ue)):
        cnt = np.searchsorted(c.pos, e, "left") - np.searchsorted(c.pos, s, "left")
        assert np.array_equal(rec["observed"], cnt)
    # additivity: observed of a split interval is the sum of the halves; doubling the reads doubles it
    mid = ((ds.astype(np.int64) + de) // 2).astype(np.int32)
    s2 = np.concatenate([ds, mid]).astype(np.int32)
    e2 = np.concatenate([mid, de]).astype(np.int32)
    o = np.lexsort((e2, s2))
    halves = run_gpu(capi, c.length, c.gc, np.repeat(c.pos, 2), np.repeat(c.mapq, 2), s2[o], e2[o],
                     np.zeros(0, np.int32), np.zeros(0, np.int32), want_tracks=False)
    tot = np.zeros(len(ds), np.int64)
    inv = np.empty_like(o)
    inv[o] = np.arange(len(o))
    ob = halves["dels"]["observed"][inv]
    tot = ob[:len(ds)].astype(np.int64) + ob[len(ds):]
    assert np.array_equal(tot, 2 * got["dels"]["observed"].astype(np.int64))
    assert np.array_equal(halves["S"], 2 * got["S"])


def test_long_chains_exact_ties_and_stuck_accumulator(capi, oracle):
    """One read on every base -> E = 1.0 exactly.  The float accumulator of a 20 Mb interval counts
    exactly up to 2^24, after which every add is an exact tie that rounds back to even: it stays at
    16777216 (likelihood.c:119 in float32).  Exercises the wave-cooperative chain and its tie path."""
    L = 20_000_000
    gc = np.full((L + 99) // 100, 40, np.uint8)
    pos = np.arange(L, dtype=np.int32)
    mapq = np.full(L, 60, np.uint8)
    s = np.array([0, 1_000_000, 3_000_017, 5, 100], np.int32)
    e = np.array([L, 19_000_000, 3_900_000, 17_000_000, 16_777_500], np.int32)
    ds, de = synth.kept_sorted(s, e)
    want = run_oracle(oracle, L, gc, pos, mapq, ds, de, ds, de)
    got = run_gpu(capi, L, gc, pos, mapq, ds, de, ds, de, want_tracks=False)
    assert want["E"][40] == 1.0
    assert want["dels"]["expected"].max() == 16777216.0
    compare(got, want, False)


def test_long_chains_random_tables(capi, oracle):
    """Long intervals over depth tables with awkward mantissas (wave path vs the literal per-base loop)."""
    for seed, cov in ((1, 0.3), (2, 7.0), (3, 40.0)):
        c = synth.make_chrom("2", 6_000_000, cov=cov, seed=seed, gaps=True)
        rng = np.random.default_rng(seed)
        s = rng.integers(0, 3_000_000, 40).astype(np.int32)
        e = (s + rng.integers(20_000, 3_000_000, 40)).astype(np.int32)
        ds, de = synth.kept_sorted(s, e)
        want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, ds[::2], de[::2])
        got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, ds[::2], de[::2], want_tracks=False)
        compare(got, want, False)


@pytest.mark.parametrize("knobs", [
    dict(CONGA_CHAIN_BLOCK_WINDOWS="40", CONGA_CHAIN_LONG_WINDOWS="30", CONGA_CHAIN_SERIAL_WINDOWS="8"),
    dict(CONGA_CHAIN_BLOCK_WINDOWS="1000000", CONGA_CHAIN_LONG_WINDOWS="100", CONGA_CHAIN_SERIAL_WINDOWS="0"),
    dict(CONGA_CHAIN_BLOCK_WINDOWS="1000000", CONGA_CHAIN_LONG_WINDOWS="1000000", CONGA_CHAIN_SERIAL_WINDOWS="1000000"),
    dict(CONGA_CHAIN_BLOCK_WINDOWS="1", CONGA_CHAIN_LONG_WINDOWS="1", CONGA_CHAIN_SERIAL_WINDOWS="0"),
])
def test_every_chain_class_gives_the_same_sums(capi, oracle, monkeypatch, knobs):
    """The chain kernel sorts intervals into four classes by length (workgroup / wave / 16-lane group / lane per
    interval).  The thresholds are tuning knobs: whichever class an interval lands in, its float32 sum is the oracle's."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    c = synth.make_chrom("9", 4_000_000, cov=3.0, seed=11, gaps=True, n_dels=150, n_dups=30)
    rng = np.random.default_rng(12)
    s = rng.integers(0, 3_000_000, 60).astype(np.int32)
    e = (s + rng.integers(1_000, 900_000, 60)).astype(np.int32)
    ds, de = synth.kept_sorted(np.concatenate([c.del_start, s]), np.concatenate([c.del_end, e]))
    us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, want_tracks=False)
    compare(got, want, False)


def test_graph_replay_gives_the_same_records(capi, oracle, monkeypatch):
    """CONGA_GRAPH=1: from the third compute of an unchanged layout the step is replayed from a captured hipGraph."""
    c, ds, de, us, ue = chrom_case("12", 700_000, cov=2.0, n_dels=40, n_dups=10)
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue)
    monkeypatch.setenv("CONGA_GRAPH", "1")
    with capi.Context(device=0) as ctx:
        ctx.chrom_begin(c.length, c.gc)
        ctx.reads(c.pos, c.mapq)
        ctx.intervals("D", ds, de)
        ctx.intervals("E", us, ue)
        runs = []
        for _ in range(5):
            ctx.compute()
            dels, dups, E, st = ctx.fetch()
            runs.append((dels.tobytes(), dups.tobytes(), E.tobytes(), st.reads_counted))
        assert all(r == runs[0] for r in runs)
        assert_records(dels, want["dels"], False)
        assert_records(dups, want["dups"], False)


def test_results_copy_into_a_torch_tensor_with_stream_events():
    """What every rank of the multi-GPU bench does per step, without the collective (tools/check_torch_interop.py).
    In a process of its own: torch brings its own HIP runtime along and has to be imported before libconga_hip.so is
    loaded, as bench.py does."""
    import subprocess
    import sys
    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_torch_interop.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_results_on_device_are_fetched_on_demand(capi, oracle):
    """CONGA_FLAG_RESULTS_ON_DEVICE: the compute sends no records over PCIe; fetch copies them when asked, and the
    device copy is what conga_results_copy hands to a gather."""
    c, ds, de, us, ue = chrom_case("11", 900_000, cov=2.0, n_dels=50, n_dups=12)
    want = run_oracle(oracle, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue)
    got = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, flags=capi.FLAG_RESULTS_ON_DEVICE, want_tracks=False)
    compare(got, want, False)
    plain = run_gpu(capi, c.length, c.gc, c.pos, c.mapq, ds, de, us, ue, want_tracks=False)
    assert got["dels"].tobytes() == plain["dels"].tobytes() and got["dups"].tobytes() == plain["dups"].tobytes()


def test_batch_mode_matches_per_chromosome_results(capi, oracle):
    """CONGA_FLAG_BATCH: several chromosomes resident at once, every kernel launched once over the batch.
    Mixed bag on purpose: with/without mappability, no intervals, no reads, a distinct likelihood GC array."""
    specs = [("1", 900_000, dict(cov=1.0, n_dels=40, n_dups=10, mappability=True)),
             ("2", 50_007, dict(cov=3.0, n_dels=6, n_dups=0, gaps=False)),
             ("3", 2_500_000, dict(cov=8.0, n_dels=80, n_dups=30)),
             ("4", 300_000, dict(cov=1.0, n_dels=0, n_dups=0, gaps=False)),
             ("5", 400_000, dict(cov=1.0, n_dels=20, n_dups=5, gaps=False)),
             ("6", 1_000_001, dict(cov=0.5, n_dels=30, n_dups=8, mappability=True))]
    cases = []
    with capi.Context(device=0, mq_threshold=5, flags=capi.FLAG_BATCH) as ctx:
        for i, (name, length, kw) in enumerate(specs):
            c, ds, de, us, ue = chrom_case(name, length, **kw)
            pos, mapq = (c.pos[:0], c.mapq[:0]) if name == "5" else (c.pos, c.mapq)
            gc_like = np.roll(c.gc, 3) if name == "3" else None
            rows = (c.map_start, c.map_end, c.map_val) if c.map_start is not None else None
            assert ctx.chrom_begin(c.length, c.gc, gc_like) == i
            ctx.reads(pos, mapq)
            cases.append((c, pos, mapq, ds, de, us, ue, rows, gc_like))
        # intervals / tracks may be attached after all reads are in (chromosome selected by index)
        for i, (c, pos, mapq, ds, de, us, ue, rows, gc_like) in enumerate(cases):
            ctx.select(i)
            if rows is not None:
                ctx.mappability(*rows)
            ctx.intervals("D", ds, de)
            ctx.intervals("E", us, ue)
        assert ctx.chrom_count() == len(specs)
        ctx.compute()
        ctx.compute()  # replay
        for i, (c, pos, mapq, ds, de, us, ue, rows, gc_like) in enumerate(cases):
            ctx.select(i)
            dels, dups, E, st = ctx.fetch()
            got = dict(dels=dels, dups=dups, E=E, counted=st.reads_counted, S=np.array(st.rd_per_gc[:]),
                       W=np.array(st.window_per_gc[:]), rd=ctx.read_depth())
            if rows is not None and len(ds) + len(us):
                got["map"] = ctx.mappability_track()
            want = run_oracle(oracle, c.length, c.gc, pos, mapq, ds, de, us, ue, mq=5, rows=rows, gc_like=gc_like)
            compare(got, want, rows is not None)
        # the packed device records are the per-chromosome records in begin order
        ptr, n = ctx.results_device()
        assert n == sum(len(x[3]) + len(x[5]) for x in cases) and ptr
        ctx.reset()
        assert ctx.chrom_count() == 0
        c = cases[1][0]
        ctx.chrom_begin(c.length, c.gc)
        ctx.reads(cases[1][1], cases[1][2])
        ctx.intervals("D", cases[1][3], cases[1][4])
        dels = ctx.finish()[0]
        want = run_oracle(oracle, c.length, c.gc, cases[1][1], cases[1][2], cases[1][3], cases[1][4],
                          cases[1][5], cases[1][6], mq=5)
        assert_records(dels, want["dels"], False)


@pytest.mark.parametrize("total_reads", [None, 4096, 5 * 1024 + 1, 1023])
def test_batch_of_many_tiny_chromosomes(capi, oracle, total_reads):
    """Sixty contigs of a few kb with 0 .. 900 reads each in one batch: one 1024-tuple chunk of the tuple pass spans
    several chromosomes (and empty ones), the batch ends exactly on / just past / before a chunk boundary, intervals
    of one base, intervals that reach the contig end."""
    rng = np.random.default_rng(31337)
    contigs = []
    for i in range(60):
        L = int(rng.integers(1_500, 60_000))
        n = int(rng.integers(0, 900)) if i % 7 else 0
        contigs.append([L, n])
    if total_reads is not None:  # trim / pad the last non-empty contigs so that the batch has exactly this many tuples
        have = sum(n for _, n in contigs)
        i = len(contigs) - 1
        while have != total_reads:
            d = total_reads - have
            n_new = max(0, min(5000, contigs[i][1] + d))
            have += n_new - contigs[i][1]
            contigs[i][1] = n_new
            i = (i - 1) % len(contigs)
    cases = []
    with capi.Context(device=0, mq_threshold=9, flags=capi.FLAG_BATCH) as ctx:
        for L, n in contigs:
            gc = synth.make_gc_track(L, rng, gaps=False)
            pos = np.sort(rng.integers(0, L, n)).astype(np.int32)
            mapq = rng.choice(np.array([0, 5, 9, 10, 30, 60], np.uint8), n)
            k = int(rng.integers(0, 6))
            s = rng.integers(0, max(L - 1, 1), k).astype(np.int32)
            e = np.minimum(s + rng.integers(1, L, k), L).astype(np.int32)
            ds, de = synth.kept_sorted(s, e, min_sv_size=1)
            us, ue = ds[::2], de[::2]
            ctx.chrom_begin(L, gc)
            ctx.reads(pos, mapq)
            ctx.intervals("D", ds, de)
            ctx.intervals("E", us, ue)
            cases.append((L, gc, pos, mapq, ds, de, us, ue))
        ctx.compute()
        for i, (L, gc, pos, mapq, ds, de, us, ue) in enumerate(cases):
            ctx.select(i)
            dels, dups, E, st = ctx.fetch()
            got = dict(dels=dels, dups=dups, E=E, counted=st.reads_counted, S=np.array(st.rd_per_gc[:]),
                       W=np.array(st.window_per_gc[:]))
            want = run_oracle(oracle, L, gc, pos, mapq, ds, de, us, ue, mq=9)
            compare(got, want, False)


def test_batch_unsorted_chromosome_is_reported_per_chromosome(capi):
    c1, ds, de, us, ue = chrom_case("7", 200_000, cov=1.0, n_dels=10, gaps=False)
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        ctx.chrom_begin(c1.length, c1.gc)
        ctx.reads(c1.pos, c1.mapq)
        ctx.intervals("D", ds, de)
        ctx.chrom_begin(c1.length, c1.gc)
        ctx.reads(c1.pos[::-1].copy(), c1.mapq)
        ctx.intervals("D", ds, de)
        ctx.compute()
        ctx.select(0)
        assert len(ctx.fetch()[0]) == len(ds)
        ctx.select(1)
        with pytest.raises(capi.CongaError) as err:
            ctx.fetch()
        assert err.value.status == capi.CONGA_ERR_UNSORTED
