"""BASELINE.json's configurations at FULL size through the HIP path, every record compared with the oracle.

configs[1]  22 autosomes, 42 000 deletion rows, 1x sample                    (likelihood.c:108-169, read_distribution.c:49-84)
configs[2]  + 6 000 duplication rows and the 100-mer mappability track       (svs.c:317-377, likelihood.c:121-128)
configs[4]  5x with --rp: a chromosome-21-sized case, >= 100 000 records against oracle/conga_oracle_sr.c
            (split_read.c:75-354, bam_data.c:29-154, likelihood.c:41-94)
The smaller parity cases live in test_gpu_parity.py / test_gpu_split_reads.py; these are the sizes the bench line is
quoted on.
"""
import numpy as np
import pytest

from conga_amd import synth
from test_gpu_parity import assert_records

pytestmark = pytest.mark.gpu


def genome(config):
    n_dups = synth.N_DUPS_GENOME if config != "dels" else 0
    out = []
    for name, length, nd, nu in synth.genome_plan(synth.GRCH37_AUTOSOMES, synth.N_DELS_GENOME, n_dups):
        c = synth.make_chrom(name, length, cov=1.0, n_dels=nd, n_dups=nu, mappability=(config != "dels"))
        ds, de = synth.kept_sorted(c.del_start, c.del_end)
        us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
        out.append((c, ds, de, us, ue))
    return out


def oracle_chrom(O, c, ds, de, us, ue, with_map):
    rd, counted = O.count_reads(c.length, c.pos, c.mapq, -1)
    E, S, W = O.calc_mean_per_chr(rd, c.gc)
    m = O.paint_mappability(c.length, c.map_start, c.map_end, c.map_val) if with_map else None
    od = O.find_depths(rd, m, c.gc, E, "D", O.make_svs(ds, de))
    ou = O.find_depths(rd, m, c.gc, E, "E", O.make_svs(us, ue))
    return counted, E, S, W, od, ou


def test_configs1_whole_genome_every_record(capi, oracle):
    """The bench line's workload through the cohort route (conga_sample_reads from pinned memory)."""
    chroms = genome("dels")
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        for c, ds, de, us, ue in chroms:
            ctx.chrom_begin(c.length, c.gc)
            ctx.intervals("D", ds, de)
        n = sum(len(c.pos) for c, *_ in chroms)
        pos, mapq, off = ctx.host_alloc(n, np.int32), ctx.host_alloc(n, np.uint8), np.zeros(len(chroms) + 1, np.uint64)
        at = 0
        for k, (c, *_r) in enumerate(chroms):
            pos[at:at + len(c.pos)] = c.pos
            mapq[at:at + len(c.pos)] = c.mapq
            at += len(c.pos)
            off[k + 1] = at
        ctx.sample_reads(pos, mapq, off)
        ctx.compute()
        recs, E, st = ctx.sample_fetch(want_stats=True)
    assert len(recs) > 39_000
    at = 0
    for k, (c, ds, de, us, ue) in enumerate(chroms):
        counted, Eo, S, W, od, _ou = oracle_chrom(oracle, c, ds, de, us, ue, False)
        assert st[k].reads_counted == counted and st[k].depth_materialized == 0
        assert np.array_equal(np.array(st[k].rd_per_gc[:]), S) and np.array_equal(np.array(st[k].window_per_gc[:]), W)
        assert np.array_equal(E[k].view(np.uint32), Eo.view(np.uint32)), "expected_read_depth, chromosome " + c.name
        assert_records(recs[at:at + len(ds)], od, has_map=False)
        at += len(ds)
    assert at == len(recs)


def test_configs2_whole_genome_every_record(capi, oracle):
    """dels + dups + the mappability track (19 M rows), through the staging ring."""
    chroms = genome("dels+dups+map")
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        for c, ds, de, us, ue in chroms:
            ctx.chrom_begin(c.length, c.gc)
            ctx.reads(c.pos, c.mapq)
            ctx.mappability(c.map_start, c.map_end, c.map_val)
            ctx.intervals("D", ds, de)
            ctx.intervals("E", us, ue)
        ctx.compute()
        res = ctx.fetch_all()
    total = 0
    for (c, ds, de, us, ue), (gd, gu, gE, st) in zip(chroms, res):
        counted, Eo, _S, _W, od, ou = oracle_chrom(oracle, c, ds, de, us, ue, True)
        assert st.reads_counted == counted
        assert np.array_equal(gE.view(np.uint32), Eo.view(np.uint32))
        assert_records(gd, od, has_map=True)
        assert_records(gu, ou, has_map=True)
        total += len(gd) + len(gu)
    assert total > 45_000


def test_configs4_chromosome_sized_split_reads(capi, oracle):
    """A chromosome-21-sized 5x sample with --rp: the first 120 000 records (planted junction reads among them) through
    the HIP split-read stage and through oracle/conga_oracle_sr.c -- element / mapping / row counts and both support
    columns."""
    from conga_amd import rp_bench
    name, length, nd, nu = [p for p in synth.genome_plan(synth.GRCH37_AUTOSOMES, synth.N_DELS_GENOME, synth.N_DUPS_GENOME) if p[0] == "21"][0]
    ch = rp_bench.make_rp_chrom(name, length, nd, nu, 5.0)
    k = 120_000
    first = int(np.searchsorted(ch["pos"], 14_000_000))   # past the leading gap, into the SVs
    sl = slice(first, first + k)
    pos, mapq, flag = ch["pos"][sl], ch["mapq"][sl], ch["flag"][sl]
    codes, qual = ch["codes"][sl], ch["qual"][sl]
    lq = np.full(k, rp_bench.READ_LEN, np.int32)
    off = np.arange(k, dtype=np.uint64) * rp_bench.READ_LEN
    rows, counts = oracle.split_read_rows(ch["ref"].tobytes(), ch["sat_s"], ch["sat_e"], pos, mapq, flag, lq, off,
                                          codes.reshape(-1), qual.reshape(-1), -1, 60)
    od, ou = oracle.make_svs(ch["ds"], ch["de"]), oracle.make_svs(ch["us"], ch["ue"])
    oracle.count_read_pairs(rows, od, ou)
    with capi.Context(device=0, flags=capi.FLAG_BATCH) as ctx:
        ctx.chrom_begin(ch["L"], ch["gc"])
        ctx.reads(pos, mapq)
        ctx.reference(ch["ref"].tobytes())
        ctx.satellites(ch["sat_s"], ch["sat_e"])
        rp_bench.stage_uniform(ctx, pos, mapq, flag, codes, qual)
        ctx.intervals("D", ch["ds"], ch["de"])
        ctx.intervals("E", ch["us"], ch["ue"])
        dels, dups, _E, st = ctx.finish()
    assert (st.split_elements, st.split_mappings, st.split_del_rows, st.split_dup_rows) == tuple(int(x) for x in counts)
    assert counts[0] > 200_000 and counts[2] + counts[3] > 10           # the junction reads pair up
    assert np.array_equal(dels["border_rp"], od["border_rp"]) and np.all(dels["rp"] == 0)
    assert np.array_equal(dups["rp"], ou["rp"]) and np.all(dups["border_rp"] == 0)
    assert int((dels["border_rp"] > 0).sum()) + int((dups["rp"] > 0).sum()) > 5


def test_configs4_chromosome_1_records_in_place_equal_the_staged_ones(capi):
    """configs[4] at the size of the longest chromosome (249 Mb, 5x: 11.4 M records with sequences): the `conga --rp` run that
    decodes on the GPU and maps the records where the inflate left them, the run on the host decoders (records through the
    pinned staging) and the C-ABI's staged route give the same three files / the same support for every interval
    (conga_amd/rp_bench.py checks all three against each other; bam_data.c:201-216, split_read.c:206-354, likelihood.c:41-94)."""
    import argparse
    from conga_amd import rp_bench
    out, _sample = rp_bench.leg(argparse.Namespace(rp_chroms="1", chroms="", steps=3, rp_cli=True), dict(local_rank=0))
    assert "checked" in out, out.get("regime")
    assert out["records"] > 11_000_000 and out["split_rows"] > 1000 and out["supported_dels"] > 500
    assert out["end_to_end"]["gpu_decode_calls_per_sample"] == 1
