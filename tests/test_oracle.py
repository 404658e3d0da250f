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


def test_oracle_reproduces_the_committed_golden_fixture(oracle, tmp_path):
    """tests/golden/small_chr.npz + the six .bed files were generated by tests/golden/make_golden.py; the oracle must
    keep producing them bit for bit / byte for byte (they pin the checker against drift; they are not reference outputs)."""
    z = np.load(os.path.join(GOLDEN, "small_chr.npz"))
    L, step = int(z["length"]), int(z["step"])
    rd, counted = oracle.count_reads(L, z["pos"], z["mapq"], -1)
    want_rd = np.zeros(L, np.int16)
    want_rd[z["rd_nonzero_idx"]] = z["rd_nonzero_val"]
    assert np.array_equal(rd, want_rd) and counted == int(z["counted"])
    E, S, W = oracle.calc_mean_per_chr(rd, z["gc"], step)
    assert np.array_equal(E.view(np.uint32), z["E"].view(np.uint32))
    assert np.array_equal(S, z["S"]) and np.array_equal(W, z["W"])
    m = oracle.paint_mappability(L, z["map_start"], z["map_end"], z["map_val"])
    dels = oracle.find_depths(rd, m, z["gc"], E, "D", oracle.make_svs(z["dels"]["start"], z["dels"]["end"]), step)
    dups = oracle.find_depths(rd, m, z["gc"], E, "E", oracle.make_svs(z["dups"]["start"], z["dups"]["end"]), step)
    assert dels.tobytes() == z["dels"].tobytes() and dups.tobytes() == z["dups"].tobytes()
    for tag, have_map in (("map", True), ("nomap", False)):
        paths = [str(tmp_path / ("%s_%s.bed" % (tag, k))) for k in ("svs", "dels", "dups")]
        oracle.output_svs("21", dels, dups, *paths, have_mappability=have_map, write_headers=True)
        for k, p in zip(("svs", "dels", "dups"), paths):
            assert open(p, "rb").read() == open(os.path.join(GOLDEN, "small_chr_%s_%s.bed" % (tag, k)), "rb").read(), (tag, k)


def test_split_read_oracle_on_planted_junctions(oracle):
    """split_read.c / bam_data.c:29-154 semantics on hand-made reads: a deletion junction, a tandem-duplication
    junction, an ordinary read, the quality-mean carry-over between the two half reads, and pos == 0."""
    rng = np.random.default_rng(2)
    L = 30_000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    code = np.full(256, 15, np.uint8)
    for k, v in {65: 1, 67: 2, 71: 4, 84: 8}.items():
        code[k] = v
    reads = [
        (9_950, np.concatenate([ref[9_950:10_000], ref[12_000:12_050]]), np.full(100, 30)),   # deletion 10000-12000
        (17_950, np.concatenate([ref[17_950:18_000], ref[15_000:15_050]]), np.full(100, 30)), # tandem dup 15000-18000
        (20_000, ref[20_000:20_100], np.full(100, 30)),                                        # ordinary
        (0, ref[0:100], np.full(100, 30)),                                                     # pos == 0: skipped
        # first half qualities sum to 499: a fresh mean would be 9 (< threshold 10), the carried-over mean is 10
        (22_000, ref[22_000:22_100], np.concatenate([np.full(49, 10), [9], np.full(50, 40)])),
    ]
    reads.sort(key=lambda r: r[0])
    pos = np.array([r[0] for r in reads], np.int32)
    lq = np.full(len(reads), 100, np.int32)
    off = (np.arange(len(reads)) * 100).astype(np.uint64)
    seq = code[np.concatenate([r[1] for r in reads])]
    qual = np.concatenate([r[2] for r in reads]).astype(np.uint8)
    mapq = np.full(len(reads), 60, np.uint8)
    flag = np.zeros(len(reads), np.uint16)
    rows, counts = oracle.split_read_rows(bytes(ref), [], [], pos, mapq, flag, lq, off, seq, qual, mq_threshold=10)
    # every read but the one at pos 0 yields two elements, including the carry-over read's second element
    assert counts[0] == 8
    got = sorted((r["sv_type"].decode(), int(r["left_end"]), int(r["right_start"])) for r in rows)
    # deletion: anchor 9950 (first half), second half maps at 12000 -> left end 9950+50-50, right start 12000+50
    # duplication: anchor 17950, second half maps back at 15000 -> left end 15000+50-50, right start 17950+50
    assert got == [("D", 9_950, 12_050), ("E", 15_000, 18_000)]
    dels, dups = oracle.make_svs([10_000], [12_000]), oracle.make_svs([15_000], [18_000])
    oracle.count_read_pairs(rows, dels, dups)
    assert dels["border_rp"][0] == 1 and dups["rp"][0] == 1
    # with the threshold at 11 the carry-over read loses its second element only
    _, counts11 = oracle.split_read_rows(bytes(ref), [], [], pos, mapq, flag, lq, off, seq, qual, mq_threshold=11)
    assert counts11[0] == 7


@pytest.mark.parametrize("mq,min_len", [(-1, 60), (20, 60), (35, 75)])
def test_split_read_oracle_matches_the_python_restatement(oracle, mq, min_len):
    """oracle/conga_oracle_sr.c against tests/sr_restatement.py (written independently from the reference's text) on a
    case with junction reads, a repeat, an N block, a dropped 61 000-hit bucket, satellites, every flag, short and odd
    read lengths and per-base qualities: same rows (as a multiset), same counters, same support columns."""
    import sr_restatement as R
    from test_gpu_split_reads import make_case
    c = make_case(seed=11, L=330_000, n_normal=700)
    rows, counts = oracle.split_read_rows(c["ref"], c["sat_s"], c["sat_e"], c["pos"], c["mapq"], c["flag"], c["lq"], c["off"],
                                          c["codes"], c["qual"], mq, min_len)
    reads = []
    for i in range(len(c["pos"])):
        o, l = int(c["off"][i]), int(c["lq"][i])
        reads.append((int(c["pos"][i]), int(c["mapq"][i]), int(c["flag"][i]), c["codes"][o:o + l], c["qual"][o:o + l]))
    sats = list(zip(c["sat_s"].tolist(), c["sat_e"].tolist()))
    want_rows, want_counts = R.split_read_rows(c["ref"].decode(), sats, reads, mq, min_len)
    assert tuple(int(x) for x in counts) == want_counts
    got = sorted((r["sv_type"].decode(), int(r["left_end"]), int(r["right_start"])) for r in rows)
    assert got == sorted(want_rows)
    od = oracle.make_svs([d[0] for d in c["dels"]], [d[1] for d in c["dels"]])
    ou = oracle.make_svs([d[0] for d in c["dups"]], [d[1] for d in c["dups"]])
    oracle.count_read_pairs(rows, od, ou)
    border, rp = R.count_read_pairs(want_rows, c["dels"], c["dups"])
    assert od["border_rp"].tolist() == border and ou["rp"].tolist() == rp
    if mq == -1:
        assert want_counts[0] > 500 and len(want_rows) > 10 and sum(border) > 5 and sum(rp) > 5
