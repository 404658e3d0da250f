"""BASELINE configs[4] as a leg of bench.py: a 5x sample with the split-read path enabled (`--rp` with `--dups`).

The reference runs find_split_reads on every record that passes the gate of count_reads_bam (bam_data.c:205-207), maps
both halves of the read against the chromosome's 10-mer index (split_read.c:75-204,206-354), pairs the mappings
(bam_data.c:29-154) and counts them against the chromosome's known SVs (likelihood.c:41-94); the depth / likelihood
path then runs as in the other configurations and copies the support into READ_PAIR.

Inputs (synthetic): a random ACGT reference per chromosome with a leading N gap, ~1 % of bases in satellite
blocks, reads of 100 bases at 5x copied from the reference with 0.3 % substitutions, reads across the junctions of the
deletions / duplications the truth genotype carries, 42 000 x L/genome deletions and 6 000 x L/genome duplications.
The whole genome is 144 M records = 21 GB of sequences and qualities; the default leg takes the chromosomes named by
`--rp-chroms` (19-22: 11 M records) so that the default bench run stays within minutes, and says so in its workload.
"""
import ctypes as C
import time

import numpy as np

from . import capi, synth

CODE = np.full(256, 15, np.uint8)
for _ch, _v in ((b"A", 1), (b"C", 2), (b"G", 4), (b"T", 8)):
    CODE[_ch[0]] = _v
ACGT = np.frombuffer(b"ACGT", np.uint8)
READ_LEN = 100


def stage_uniform(ctx, pos, mapq, flag, codes2d, qual2d):
    """conga_split_reads_staging / commit for reads of one length, whole staging buffers at a time."""
    lib, h = ctx._lib, ctx._h
    n, l = codes2d.shape
    half = (l + 1) // 2
    need = half + l
    stg = capi.SplitStaging()
    i = 0
    while i < n:
        ctx._check(lib.conga_split_reads_staging(h, C.byref(stg)))
        k = int(min(n - i, stg.capacity_reads, stg.capacity_bytes // need))
        c = codes2d[i:i + k]
        if l % 2:
            c = np.concatenate([c, np.zeros((k, 1), np.uint8)], axis=1)
        data = np.ctypeslib.as_array(stg.data, shape=(stg.capacity_bytes,))[:k * need].reshape(k, need)
        data[:, :half] = (c[:, 0::2] << 4) | c[:, 1::2]
        data[:, half:] = qual2d[i:i + k]
        np.ctypeslib.as_array(stg.data_off, shape=(stg.capacity_reads,))[:k] = np.arange(k, dtype=np.uint64) * need
        np.ctypeslib.as_array(stg.pos, shape=(stg.capacity_reads,))[:k] = pos[i:i + k]
        np.ctypeslib.as_array(stg.mapq, shape=(stg.capacity_reads,))[:k] = mapq[i:i + k]
        np.ctypeslib.as_array(stg.flag, shape=(stg.capacity_reads,))[:k] = flag[i:i + k]
        np.ctypeslib.as_array(stg.l_qseq, shape=(stg.capacity_reads,))[:k] = l
        ctx._check(lib.conga_split_reads_commit(h, k, k * need))
        i += k


def make_rp_chrom(name, length, n_dels, n_dups, cov, seed=synth.BASE_SEED):
    """One chromosome of the configs[4] workload: layout (GC track, intervals), reference, satellites, records."""
    c = synth.make_chrom(name, length, cov=cov, n_dels=n_dels, n_dups=n_dups, seed=seed)
    rng = np.random.default_rng([seed, int(name), 4])
    ref = ACGT[rng.integers(0, 4, length, dtype=np.uint8)]
    ref[c.gc.repeat(c.step)[:length] == 0] = ord("N")            # assembly gaps: the windows the GC track has at 0
    ds, de = synth.kept_sorted(c.del_start, c.del_end)
    us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
    # satellites: ~1 % of the bases in blocks of 1-50 kb
    n_sat = max(1, int(0.01 * length / 25_000))
    ss = np.sort(rng.integers(0, max(length - 60_000, 1), n_sat)).astype(np.int32)
    se = (ss + rng.integers(1_000, 50_000, n_sat)).astype(np.int32)
    # ordinary reads from the depth model, whole reads inside the chromosome, plus reads across SV junctions
    keep = (c.pos > 0) & (c.pos.astype(np.int64) + READ_LEN < length)
    pos, mapq = c.pos[keep], c.mapq[keep]
    jpos, jleft, jright = [], [], []   # a junction read: `k` bases that end at `left`, then READ_LEN - k from `right` on
    for (s_arr, e_arr, is_dup) in ((ds, de, False), (us, ue, True)):
        pick = rng.random(len(s_arr)) < 0.5                       # the individuals that carry the variant
        for s, e in zip(s_arr[pick], e_arr[pick]):
            for k in rng.integers(25, 76, 3):
                a, b = (int(e), int(s)) if is_dup else (int(s), int(e))  # dup: ...end | start...; del: ...start | end...
                if a - k > 0 and a - k + READ_LEN < length and b + READ_LEN - k < length:
                    jpos.append(a - k)
                    jleft.append(a)
                    jright.append(b)
    n0, nj = len(pos), len(jpos)
    pos = np.concatenate([pos, np.array(jpos, np.int32)])
    mapq = np.concatenate([mapq, np.full(nj, 60, np.uint8)])
    order = np.argsort(pos, kind="stable")
    pos, mapq = pos[order], mapq[order]
    bases = np.lib.stride_tricks.sliding_window_view(ref, READ_LEN)[pos]   # (row copies, no index array per base)
    where = np.empty(n0 + nj, np.int64)
    where[order] = np.arange(n0 + nj)
    for j in range(nj):                                           # the junction reads: two pieces of the reference
        k = jleft[j] - jpos[j]
        bases[where[n0 + j], k:] = ref[jright[j]:jright[j] + READ_LEN - k]
    n_err = int(rng.binomial(bases.size, 0.003))                  # substitutions (a few land on the same base: fine)
    bases.reshape(-1)[rng.integers(0, bases.size, n_err)] = ACGT[rng.integers(0, 4, n_err, dtype=np.uint8)]
    flag = np.where(rng.random(len(pos)) < 0.02, 0x400, 0).astype(np.uint16)
    pool = np.frombuffer(rng.bytes(1 << 22), np.uint8) % 21 + 20  # Phred 20..40, every read a window of a random pool
    qual = np.lib.stride_tricks.sliding_window_view(pool, READ_LEN)[rng.integers(0, len(pool) - READ_LEN, len(pos))]
    return dict(name=name, L=length, gc=c.gc, ref=ref, ds=ds, de=de, us=us, ue=ue, sat_s=ss, sat_e=se, pos=pos, mapq=mapq,
                flag=flag, codes=CODE[bases], qual=qual, n_junction=nj)


def open_chrom(ctx, ch):
    ctx.chrom_begin(ch["L"], ch["gc"])
    ctx.reads(ch["pos"], ch["mapq"])
    ctx.reference(ch["ref"].tobytes())
    ctx.satellites(ch["sat_s"], ch["sat_e"])
    stage_uniform(ctx, ch["pos"], ch["mapq"], ch["flag"], ch["codes"], ch["qual"])
    ctx.intervals("D", ch["ds"], ch["de"])
    ctx.intervals("E", ch["us"], ch["ue"])


def bucket_bytes_per_read(chroms):
    """Bytes of 10-mer index a read's four bucket scans touch on average.  A seed falls into a bucket with probability
    proportional to its size; over a random ACGT reference with n indexed bases the bucket sizes are Poisson(n / 4^10),
    whose size-biased mean is n / 4^10 + 1 (checked against the exact histogram of chromosome 21: 591 vs 586 bytes)."""
    tot, w = 0.0, 0.0
    for ch in chroms:
        n = float(np.count_nonzero(ch["ref"] != ord("N")))
        reads = len(ch["pos"])
        tot += reads * 4 * 4.0 * (n / (1 << 20) + 1.0)
        w += reads
    return tot / max(w, 1.0)


def leg(args, env):
    """-> (dict for bench.py's `configs["configs[4]"]`, a bounded sample of the records for the caller's CPU baseline)."""
    names = [n for n, _ in synth.GRCH37_AUTOSOMES] if args.rp_chroms == "all" else args.rp_chroms.split(",")
    if args.chroms:
        names = [n for n in names if n in set(args.chroms.split(","))] or args.chroms.split(",")[-1:]
    plan = {n: (l, nd, nu) for n, l, nd, nu in synth.genome_plan(synth.GRCH37_AUTOSOMES, synth.N_DELS_GENOME, synth.N_DUPS_GENOME)}
    cov = 5.0
    t0 = time.perf_counter()
    chroms = [make_rp_chrom(n, plan[n][0], plan[n][1], plan[n][2], cov) for n in names]
    t_gen = time.perf_counter() - t0
    n_reads = int(sum(len(ch["pos"]) for ch in chroms))
    n_iv = int(sum(len(ch["ds"]) + len(ch["us"]) for ch in chroms))
    rec_bytes = n_reads * (4 + 1 + 2 + 4 + 8 + READ_LEN // 2 + READ_LEN)
    out = dict(workload="BASELINE configs[4] on chromosomes %s (%d Mb; `--rp-chroms all` = the whole genome): 5x synthetic sample, "
                        "%d records of %d bases with sequences and qualities, --rp split-read path on, %d intervals (dels + dups "
                        ">= 1000 bp)" % (",".join(names), sum(ch["L"] for ch in chroms) // 1_000_000, n_reads, READ_LEN, n_iv),
               records=n_reads, intervals_per_step=n_iv, junction_reads=int(sum(ch["n_junction"] for ch in chroms)),
               generate_s=round(t_gen, 1))
    steps = max(3, min(args.steps, 5))

    def timed(ctx):
        ctx.compute()
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(steps):
            ctx.compute()
            ctx.sync()
        return (time.perf_counter() - t1) / steps

    with capi.Context(device=env["local_rank"], flags=capi.FLAG_BATCH) as ctx:
        t0 = time.perf_counter()
        for ch in chroms:
            open_chrom(ctx, ch)
        ctx.sync()
        out["handover_s"] = round(time.perf_counter() - t0, 2)     # python packing + PCIe, not the step
        t0 = time.perf_counter()
        ctx.compute()
        ctx.sync()
        out["first_compute_ms"] = round(1e3 * (time.perf_counter() - t0), 1)  # layout + the 10-mer indexes, once per layout
        t_with = timed(ctx)
        res = ctx.fetch_all()
        st = [r[3] for r in res]
    with capi.Context(device=env["local_rank"], flags=capi.FLAG_BATCH) as ctx:  # the same without the records: depth path only
        for ch in chroms:
            ctx.chrom_begin(ch["L"], ch["gc"])
            ctx.reads(ch["pos"], ch["mapq"])
            ctx.intervals("D", ch["ds"], ch["de"])
            ctx.intervals("E", ch["us"], ch["ue"])
        t_without = timed(ctx)
        plain = ctx.fetch_all()
    for (d1, u1, _e1, _s1), (d0, u0, _e0, _s0) in zip(res, plain):   # the support columns are all the split reads add
        for k in ("observed", "expected", "cn", "score"):
            assert np.array_equal(d1[k], d0[k], equal_nan=True) and np.array_equal(u1[k], u0[k], equal_nan=True), k
    sr_ms = 1e3 * (t_with - t_without)
    bucket = bucket_bytes_per_read(chroms)
    alg = n_reads * (19 + READ_LEN // 2 + READ_LEN + 16 + bucket + 2 * READ_LEN)   # record, buckets, ~2 reference compares
    out.update(ms_per_step=round(1e3 * t_with, 3), value=round(n_iv / t_with, 1), unit="intervals/s",
               records_per_s=round(n_reads / t_with, 1), regime="records resident in HBM (kernels only; the hand-over of "
               "%.1f GB of records is PCIe: %.0f ms at 54 GB/s)" % (rec_bytes / 1e9, rec_bytes / 54e6),
               depth_path_ms=round(1e3 * t_without, 3), split_read_stage_ms=round(sr_ms, 3),
               split_elements=int(sum(s.split_elements for s in st)), split_mappings=int(sum(s.split_mappings for s in st)),
               split_rows=int(sum(s.split_del_rows + s.split_dup_rows for s in st)),
               supported_dups=int(sum(int((r[1]["rp"] > 0).sum()) for r in res)),
               supported_dels=int(sum(int((r[0]["border_rp"] > 0).sum()) for r in res)),
               roofline=dict(bound="hbm", kernel="split_read_kernel", algorithmic_bytes_per_launch=int(alg),
                             avg_launch_ms=round(sr_ms, 3), achieved=round(alg / max(sr_ms, 1e-6) / 1e6, 1), peak=8000.0, unit="GB/s",
                             frac=round(alg / max(sr_ms, 1e-6) / 1e6 / 8000.0, 4), traffic=None,
                             note="instruction-bound: ~770 vector + 650 scalar instructions and ~5 dependent trips to HBM per read "
                                  "(fields -> qualities + sequence + reference -> bucket bounds -> buckets); launch time = step with "
                                  "records - step without"))
    # a bounded sample for the caller's CPU baseline (bench.py runs the oracle; nothing in this package does)
    ch = min(chroms, key=lambda c: c["L"])
    k = min(20_000, len(ch["pos"]))
    sample = dict(name=ch["name"], ref=ch["ref"].tobytes(), sat_s=ch["sat_s"], sat_e=ch["sat_e"], pos=ch["pos"][:k], mapq=ch["mapq"][:k],
                  flag=ch["flag"][:k], lq=np.full(k, READ_LEN, np.int32), off=np.arange(k, dtype=np.uint64) * READ_LEN,
                  codes=ch["codes"][:k].reshape(-1).copy(), qual=ch["qual"][:k].reshape(-1).copy())
    return out, sample
