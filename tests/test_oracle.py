"""The oracle against the only external anchors that exist for this path, and against an independent
numpy restatement of the same arithmetic.  PARITY UNPINNED: the reference has no tests or fixtures
(SURVEY.md section 4); the known answers below were recorded in SURVEY.md App. D."""
import json
import os

import numpy as np
import pytest

from conga_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_lpoisson_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "survey_appendix_d_kat.json")))
    for k in kat["lpoisson"]:
        assert oracle.lpoisson(k["observed"], k["lambda"]) == pytest.approx(k["value"], rel=0, abs=1e-12)


def test_deletion_known_answer(oracle):
    """DEL 800000-900000, obs=519, exp=967.484009f."""
    kat = json.load(open(os.path.join(GOLDEN, "survey_appendix_d_kat.json")))["deletion_example"]
    r = oracle.score(kat["observed"], np.float32(kat["expected"]), "D")
    assert r["lhomo"] == pytest.approx(kat["lhomo"], abs=1e-3)
    assert r["lhetero"] == pytest.approx(kat["lhetero"], abs=1e-4)
    assert r["lnone"] == pytest.approx(kat["lnone"], abs=1e-3)
    # the int-truncated max: (double)(-5) / lnone, not max/lnone = 0.040989
    assert r["score"] == pytest.approx(kat["score"], abs=1e-9)
    assert r["score"] == -5.0 / r["lnone"]


def test_zero_reads_gives_negative_zero(oracle):
    """obs=0 -> (int)(-0.01) = 0 -> score = -0.0 (printed -0.00), CN = 2."""
    r = oracle.score(0, np.float32(96.68), "D")
    assert r["lhomo"] == -0.01
    assert r["score"] == 0.0 and np.signbit(r["score"])
    assert r["cn"] == 2
    assert "%.2f" % r["score"] == "-0.00"


def test_expected_zero_edge(oracle):
    """E == 0 -> all three lambdas become 0.01 (likelihood.c:101-102) -> equal log-likelihoods."""
    for t in "DE":
        r = oracle.score(7, np.float32(0.0), t)
        assert r["lhomo"] == r["lhetero"] == r["lnone"]
    r = oracle.score(0, np.float32(0.0), "D")
    assert r["score"] == 0.0 and np.signbit(r["score"])


def test_dup_two_times_expected_is_float_multiply(oracle):
    ex = np.float32(3.3e38)  # 2 * ex overflows in float (inf) but not in double
    r = oracle.score(1, ex, "E")
    assert np.isinf(r["lhomo"]) or np.isnan(r["lhomo"])


def _numpy_restatement(c, start, end, sv_type, mq=-1, mappability=None):
    """Independent restatement with numpy primitives (bincount, sequential float32 cumsum)."""
    keep = c.mapq.astype(np.int64) > mq
    rd = np.bincount(c.pos[keep], minlength=c.length).astype(np.int64)
    gc_base = np.repeat(c.gc, c.step)[:c.length]
    S = np.bincount(gc_base, weights=rd, minlength=101).astype(np.int64)
    W = np.bincount(gc_base, minlength=101)
    with np.errstate(all="ignore"):
        E = S.astype(np.float32) / W.astype(np.float32)
    E[~np.isfinite(E)] = 0
    E[0] = 0
    out = []
    for s, e in zip(start, end):
        ex = np.add.accumulate(E[gc_base[s:e]], dtype=np.float32)[-1]  # strictly left to right
        ob = int(rd[s:e].sum())
        mp = float(np.add.accumulate(mappability[s:e].astype(np.float64))[-1]) / (e - s) if mappability is not None else 0.0
        out.append((ob, ex, mp))
    return rd, E, out


def test_oracle_matches_numpy_restatement(oracle):
    c = synth.make_chrom("7", 300_000, cov=2.0, n_dels=30, n_dups=8, mappability=True, gaps=False)
    ds, de = synth.kept_sorted(c.del_start, c.del_end)
    rd, counted = oracle.count_reads(c.length, c.pos, c.mapq, 10)
    E, S, W = oracle.calc_mean_per_chr(rd, c.gc)
    m = oracle.paint_mappability(c.length, c.map_start, c.map_end, c.map_val)
    got = oracle.find_depths(rd, m, c.gc, E, "D", oracle.make_svs(ds, de))
    rd_np, E_np, rows = _numpy_restatement(c, ds, de, "D", mq=10, mappability=m)
    assert np.array_equal(rd, rd_np)
    assert counted == int((c.mapq > 10).sum())
    assert np.array_equal(E.view(np.uint32), E_np.astype(np.float32).view(np.uint32))
    for g, (ob, ex, mp) in zip(got, rows):
        assert g["observed"] == ob
        assert np.float32(g["expected"]).view(np.uint32) == np.float32(ex).view(np.uint32)
        assert g["mappability"] == mp


def test_mappability_paint_inclusive_and_file_order(oracle):
    m = oracle.paint_mappability(20, [2, 5, 4, 18], [5, 8, 4, 40], [0.5, 0.25, 1.0, 0.2])
    want = np.zeros(20, np.float32)
    want[2:6] = 0.5
    want[5:9] = 0.25     # shares base 5 with the first row; later row wins
    want[4] = 1.0        # later single-base row overwrites
    want[18:20] = 0.2    # clamped at L - 1
    assert np.array_equal(m, want)


def test_bed_loader_filters_and_sorts(oracle, tmp_path):
    p = tmp_path / "dels.bed"
    p.write_text("#chr\tstart\tend\n21\t5000\t7000\n21\t100\t1099\n\n   \n21 300 1300 extra\nX\t1\t5000\n"
                 "21\t300\t1250\n2\t10\t5000\n21\t9000\t10000")
    svs = oracle.load_known_SVs(str(p), "21", 1000)
    assert list(zip(svs["start"], svs["end"])) == [(5000, 7000), (300, 1300), (9000, 10000)]
    oracle.sort_svs(svs)
    assert list(zip(svs["start"], svs["end"])) == [(300, 1300), (5000, 7000), (9000, 10000)]


def test_output_format(oracle, tmp_path):
    """Byte-level contract of SURVEY.md App. B (likelihood.c:172-288, bam_data.c:235-249)."""
    dels = oracle.make_svs([100, 5000, 9000], [2100, 7000, 12000])
    for i, (ob, ex) in enumerate([(3, 20.5), (0, 10.0), (40, 20.0)]):
        r = oracle.score(ob, np.float32(ex), "D")
        for k in ("observed", "expected", "lhomo", "lhetero", "lnone", "score", "cn"):
            dels[i][k] = r[k]
    dups = oracle.make_svs([300], [4300])
    r = oracle.score(90, np.float32(40.0), "E")
    for k in ("observed", "expected", "lhomo", "lhetero", "lnone", "score", "cn"):
        dups[0][k] = r[k]
    fs, fd, fu = (str(tmp_path / n) for n in ("o_svs.bed", "o_dels.bed", "o_dups.bed"))
    oracle.output_svs("21", dels, dups, fs, fd, fu, have_mappability=False, write_headers=True)
    svs_txt, del_txt, dup_txt = (open(f).read() for f in (fs, fd, fu))
    assert svs_txt.splitlines()[0] == "#CHR\tSTART_SV\tEND_SV\tSV_TYPE\tCOPY_NUMBER\tLIKELIHOOD\tREAD_PAIR\tMAPPABILITY"
    assert del_txt.splitlines()[0] == ("#CHR\tSTART_SV\tEND_SV\tCOPY_NUMBER\tLIKELIHOOD\tREAD_PAIR\tMAPPABILITY"
                                       "\tOBSERVED_READS\tEXPECTED_READS")
    rows = [l.split("\t") for l in del_txt.splitlines()[1:]]
    assert [r[3] for r in rows] == ["1/1", "1/1", "0/0"]
    assert rows[1][4] == "-0.00" and rows[1][6] == "N/A" and rows[1][7] == "0" and rows[1][8] == "10.0"
    # _svs.bed DEL rows without mappability have seven columns
    assert all(len(l.split("\t")) == 7 for l in svs_txt.splitlines()[1:] if "\tDEL\t" in l)
    assert dup_txt.splitlines()[1].split("\t")[3] in ("1/1", "0/1")
