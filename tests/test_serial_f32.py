"""conga_repeat_add_f32 (conga_amd/csrc/serial_f32.h) must equal k literal float32 adds, bit for bit.
It is what lets the GPU reproduce the reference's serial accumulation `expected_rd += E[gc]`
(likelihood.c:111,119) without doing one dependent add per base.  The host build of the same inline
function is exported as conga_host_repeat_add_f32, so this runs without a GPU."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


def naive(s, c, k):
    s = np.float32(s)
    c = np.float32(c)
    with np.errstate(over="ignore"):
        for _ in range(k):
            s = np.float32(s + c)
    return s


def bits(x):
    return np.float32(x).view(np.uint32)


def f32(b):
    return np.uint32(b).view(np.float32)


finite_pos = st.integers(min_value=0, max_value=0x7F7FFFFF).map(f32)
finite_any = st.builds(lambda b, neg: f32(b | (0x80000000 if neg else 0)), st.integers(0, 0x7F7FFFFF), st.booleans())


@settings(max_examples=300, deadline=None)
@given(s=finite_pos, c=finite_pos, k=st.integers(min_value=0, max_value=400))
def test_any_nonnegative_operands(capi, s, c, k):
    assert bits(capi.host_repeat_add_f32(s, c, k)) == bits(naive(s, c, k))


@settings(max_examples=400, deadline=None)
@given(ec=st.integers(100, 140), mc=st.integers(0, 0x7FFFFF), de=st.integers(0, 26), ms=st.integers(0, 0x7FFFFF),
       tz=st.integers(0, 23), k=st.integers(1, 300))
def test_near_ties_and_binade_edges(capi, ec, mc, de, ms, tz, k):
    """Addends with trailing zeros (exact ties at some binade) and accumulators next to a binade top."""
    mc &= ~((1 << tz) - 1)
    c = f32((ec << 23) | mc)
    for m in (ms, 0x7FFFFF - (ms % 300), 0):
        s = f32(((ec + de) << 23) | m)
        assert bits(capi.host_repeat_add_f32(s, c, k)) == bits(naive(s, c, k))


@settings(max_examples=300, deadline=None)
@given(s=finite_any, c=finite_any, k=st.integers(min_value=0, max_value=300))
def test_signed_operands(capi, s, c, k):
    """Negative addends only arise from a wrapped `short` depth counter, but must still be exact."""
    assert bits(capi.host_repeat_add_f32(s, c, k)) == bits(naive(s, c, k))


@settings(max_examples=200, deadline=None)
@given(e=st.integers(110, 130), ms=st.integers(0, 0x7FFFFF), mc=st.integers(0, 0x7FFFFF), de=st.integers(-3, 12),
       k=st.integers(1, 300), sn=st.booleans(), cn=st.booleans())
def test_signed_operands_of_similar_magnitude(capi, e, ms, mc, de, k, sn, cn):
    s = f32(((e + de) << 23) | ms | (0x80000000 if sn else 0))
    c = f32((e << 23) | mc | (0x80000000 if cn else 0))
    assert bits(capi.host_repeat_add_f32(s, c, k)) == bits(naive(s, c, k))


@settings(max_examples=600, deadline=None)
@given(ec=st.integers(110, 125), mc=st.integers(0, 0x7FFFFF), de=st.integers(2, 24), back=st.integers(0, 1100),
       k=st.integers(1, 1024))
def test_one_crossing_inside_a_window(capi, ec, mc, de, back, k):
    """The chain kernels' common irregular case: the accumulator sits `back` steps below a binade top and one GC
    window (k <= 1024 adds) carries it across -- conga_window_add_f32 resolves that without its general loop."""
    c = f32((ec << 23) | mc)
    top = f32((ec + de + 1) << 23)
    s = np.float32(top)
    for _ in range(3):
        s = np.nextafter(s, np.float32(0), dtype=np.float32)
    s = np.float32(s - np.float32(back) * c)
    if not np.isfinite(s) or s <= 0:
        s = np.float32(top / 2)
    assert bits(capi.host_repeat_add_f32(s, c, k)) == bits(naive(s, c, k))


def test_chains_from_zero_like_the_kernels_walk_them(capi):
    """What interval_chain does: start at 0, one call per GC window (k <= step bases of one table value), left to
    right.  The early windows cross several binades inside one call; compare with the literal per-base loop."""
    rng = np.random.default_rng(20221124)
    for case in range(40):
        table = (rng.random(12).astype(np.float32) * np.float32(10.0 ** rng.integers(-4, 2))).astype(np.float32)
        table[rng.integers(0, 12)] = np.float32(0)
        s_fast = np.float32(0)
        s_ref = np.float32(0)
        for w in range(int(rng.integers(5, 160))):
            c = table[rng.integers(0, 12)]
            k = int(rng.choice([1, 7, 37, 100, 100, 100, 1000, 1024]))
            s_fast = capi.host_repeat_add_f32(s_fast, c, k)
            with np.errstate(over="ignore"):
                acc = np.full(k, c, np.float32)
                s_ref = np.add.accumulate(np.concatenate([[s_ref], acc]).astype(np.float32), dtype=np.float32)[-1]
            assert bits(s_fast) == bits(s_ref), (case, w, float(c), k)


def test_subnormals_and_zero(capi):
    rng = np.random.default_rng(5)
    for _ in range(300):
        c = f32(int(rng.integers(0, 0x800000)))
        s = f32(int(rng.integers(0, 0x1000000)))
        k = int(rng.integers(1, 200))
        assert bits(capi.host_repeat_add_f32(s, c, k)) == bits(naive(s, c, k))
    assert bits(capi.host_repeat_add_f32(0.0, 0.0, 100)) == bits(0.0)
    assert bits(capi.host_repeat_add_f32(3.5, 0.0, 100)) == bits(3.5)
    for s0, c0 in [(0.0, -0.0), (-0.0, 0.0), (-0.0, -0.0), (0.0, 0.0), (-2.5, 0.0), (-2.5, -0.0)]:
        assert bits(capi.host_repeat_add_f32(s0, c0, 9)) == bits(naive(s0, c0, 9))


def test_survey_probe_float_vs_double(capi):
    """the float-serial sum differs from a double sum at the 1e-5 level, so the
    rounding sequence matters.  A whole interval walked window by window equals the per-base loop."""
    rng = np.random.default_rng(11)
    E = (rng.random(101) * 0.02).astype(np.float32)
    gcs = rng.integers(20, 75, 1000)
    s_fast = np.float32(0)
    s_ref = np.float32(0)
    for g in gcs:
        s_fast = capi.host_repeat_add_f32(s_fast, E[g], 100)
        s_ref = naive(s_ref, E[g], 100)
    assert bits(s_fast) == bits(s_ref)
    as_double = float(np.sum(E[gcs].astype(np.float64)) * 100)
    assert abs(float(s_ref) - as_double) / as_double > 1e-7


def test_long_runs(capi):
    for s, c, k in [(0.0, 0.0097, 5_000_000), (1.0, 1e-3, 2_000_000), (16777216.0, 1.0, 1000), (0.5, 2.0 ** -25, 10_000)]:
        # closed form is not available; compare against a vectorised literal loop in chunks
        ref = np.float32(s)
        cc = np.float32(c)
        for _ in range(k):
            ref = np.float32(ref + cc)
            if ref == np.float32(ref + cc):  # stuck: no further change
                break
        assert bits(capi.host_repeat_add_f32(s, c, k)) == bits(ref)
