"""BASELINE configs[4] as a leg of bench.py: a 5x sample with the split-read path enabled (`--rp` with `--dups`).

The reference runs find_split_reads on every record that passes the gate of count_reads_bam (bam_data.c:205-207), maps
both halves of the read against the chromosome's 10-mer index (split_read.c:75-204,206-354), pairs the mappings
(bam_data.c:29-154) and counts them against the chromosome's known SVs (likelihood.c:41-94); the depth / likelihood
path then runs as in the other configurations and copies the support into READ_PAIR.

Inputs (synthetic): a random ACGT reference per chromosome with a leading N gap, ~1 % of bases in satellite
blocks, reads of 100 bases at 5x copied from the reference with 0.3 % substitutions, reads across the junctions of the
deletions / duplications the truth genotype carries, 42 000 x L/genome deletions and 6 000 x L/genome duplications.
The whole genome is 131.5 M records = 20 GB of sequences and qualities and a 12.7 GB BAM; `bench.py` takes it when the host has
the memory and the scratch space to spare (`--rp-chroms auto`), otherwise chromosomes 20-22 (6.56 M records), and the leg says
which in its workload.
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


def write_fasta_fast(path, chroms, width=60):
    """[(name, uint8[L])] -> FASTA + .fai, whole lines at a time"""
    fai = []
    with open(path, "wb") as f:
        for name, seq in chroms:
            f.write(b">" + name.encode() + b"\n")
            fai.append((name, len(seq), f.tell(), width, width + 1))
            full = len(seq) // width * width
            lines = np.empty((full // width, width + 1), np.uint8)
            lines[:, :width] = seq[:full].reshape(-1, width)
            lines[:, width] = 10
            f.write(lines.tobytes())
            if full < len(seq):
                f.write(seq[full:].tobytes() + b"\n")
    with open(path + ".fai", "w") as f:
        for r in fai:
            f.write("\t".join(str(x) for x in r) + "\n")


def split_map_bytes(chroms):
    """Algorithmic bytes of one split_map_kernel launch over these records: per read the record's place
    (8) and header fields (16), its packed sequence (l / 2; the qualities are not read with a threshold <= 0), per half read two
    bucket bounds (2 x 8), two binary searches (4 bytes per probe, log2 of the mean kept bucket) and the reference under ~1.2
    candidates (l / 4 bytes each, packed codes)."""
    tot = 0.0
    for ch in chroms:
        n = float(np.count_nonzero(ch["ref"] != ord("N")))
        probes = np.ceil(np.log2(n / (1 << 20) + 2.0))
        tot += len(ch["pos"]) * (8 + 16 + READ_LEN // 2 + 2 * (2 * 8 + 2 * 4 * probes + 1.2 * READ_LEN // 4))
    return tot


def leg(args, env):
    """-> (dict for bench.py's `configs["configs[4]"]`, a bounded sample of the records for the caller's CPU baseline).

    `value` is the configuration's own path: BGZF bytes of a 5x sample -> the three output files, per further sample of a
    `conga --cohort --rp` run, the decode on the GPU and the split-read stage reading the records where the inflate left them
    (bam_data.c:201-216 with find_split_reads inside the BAM loop).  Beside it: the same with the host decoders, and the
    kernels alone on records resident in HBM (handed over through the C-ABI's pinned staging)."""
    import json
    import os
    import shutil
    import tempfile
    from . import e2e_bench, formats
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

    # ---- the C-ABI's staged route, and the kernels alone on what it left in HBM
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
        first = ctx.fetch_all()   # (the first launch on an index runs without the solo / echo bits: every seed asks its bucket)
        t_with = timed(ctx)
        res = ctx.fetch_all()
        st = [r[3] for r in res]
        # the launches behind the first one use the bits: the same rows, the same support, at this leg's full size
        for (d1, u1, _e1, s1), (d0, u0, _e0, s0) in zip(res, first):
            assert np.array_equal(d1["border_rp"], d0["border_rp"]) and np.array_equal(u1["rp"], u0["rp"]), "support differs with the solo / echo bits"
            assert (s1.split_elements, s1.split_mappings, s1.split_del_rows, s1.split_dup_rows) == \
                (s0.split_elements, s0.split_mappings, s0.split_del_rows, s0.split_dup_rows), "split-read counters differ with the solo / echo bits"
        out["bits_checked"] = "support columns and element / mapping / row counters of every chromosome equal between the first launch (no solo / echo bits) and the later ones"
        # the split-read launch by itself: HIP events around it on the context's stream (CONGA_FLAG_PROFILE)
        ctx.set_profile(True)
        k_ms = []
        for _ in range(steps):
            ctx.compute()
            ctx.select(0)
            k_ms.append(float(ctx.fetch()[3].kernel_ms[capi.KERNEL_NAMES.index("split_map")]))
        ctx.set_profile(False)
    sr_ms = float(np.mean(k_ms))
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
    alg = split_map_bytes(chroms)
    out.update(kernel_only=dict(ms_per_step=round(1e3 * t_with, 3), value=round(n_iv / t_with, 1), records_per_s=round(n_reads / t_with, 1),
                                regime="records resident in HBM (staged route), compute + sync per step; depth path alone %.3f ms"
                                       % (1e3 * t_without)),
               split_elements=int(sum(s.split_elements for s in st)), split_mappings=int(sum(s.split_mappings for s in st)),
               split_rows=int(sum(s.split_del_rows + s.split_dup_rows for s in st)),
               supported_dups=int(sum(int((r[1]["rp"] > 0).sum()) for r in res)),
               supported_dels=int(sum(int((r[0]["border_rp"] > 0).sum()) for r in res)),
               roofline=dict(bound="hbm", kernel="split_map_kernel", algorithmic_bytes_per_launch=int(alg),
                             avg_launch_ms=round(sr_ms, 4), achieved=round(alg / max(sr_ms, 1e-6) / 1e6, 1), peak=8000.0, unit="GB/s",
                             frac=round(alg / max(sr_ms, 1e-6) / 1e6 / 8000.0, 4), traffic=None,
                             records_per_s=round(n_reads / (sr_ms * 1e-3), 1),
                             note="HIP events around the launch on the context's stream (CONGA_FLAG_PROFILE); a lane per half-read "
                                  "element.  Four seeds in five are settled by two bits read next to the read's own place (split_map.hip.h: "
                                  "solo / echo); the fifth probes its 10-mer's bucket, each probe a 128-byte line from beyond L2 -- "
                                  "`random_fetch` prices the launch against the machine's rate for such fetches, `frac` against streamed bytes"))
    # HBM traffic and the bound that holds for this kernel, from the PMC campaign in profiles/ (per element, scaled to this run)
    try:
        with open(os.path.join(e2e_bench.ROOT, "profiles", "split_map_traffic.json")) as f:
            tr = json.load(f)
        n_el = out["split_elements"]
        raw = tr["fetch_bytes_raw_per_element"] * n_el
        cal = tr["random_probe_calibration"]
        out["roofline"].update(
            traffic=round(tr["hbm_bytes_per_launch"] / tr["elements"] * n_el),
            traffic_source="profiles/split_map_traffic.json (rocprofv3 --pmc campaign %s on the whole genome, per half-read element; not measured "
                           "in this run)" % tr["campaign"],
            random_fetch=dict(unit="TB/s of FETCH_SIZE (raw: a 128-byte line counts 63 bytes)", achieved=round(raw / (sr_ms * 1e-3) / 1e12, 3),
                              peak=round(cal["fetch_rate_raw_bytes_per_s"] / 1e12, 3),
                              frac=round(raw / (sr_ms * 1e-3) / cal["fetch_rate_raw_bytes_per_s"], 3),
                              line_fetches_per_element=round(tr["fetch_bytes_raw_per_element"] / cal["FETCH_SIZE_bytes_raw_per_probe_beyond_L2"], 2),
                              peak_source="tools/gathercal.hip: %.1f G independent 4-byte probes a second into a 2 GiB table"
                                          % (cal["probes_per_s_2GiB_table"] / 1e9)))
    except (OSError, KeyError, ValueError):
        pass

    # ---- the configuration's path: BAM files in, three files out
    d = tempfile.mkdtemp(prefix="conga_bench_rp_", dir=os.environ.get("CONGA_BENCH_TMP", "/tmp"))
    try:
        if not (os.path.exists(e2e_bench.CONGA) and os.path.exists(e2e_bench.BAMWRITE)):
            raise OSError("conga / tools/bamwrite are not built")
        if not getattr(args, "rp_cli", True):
            raise OSError("--no-cli")
        half = (READ_LEN + 1) // 2
        bam_in = []
        for ch in chroms:
            c = ch["codes"]
            bam_in.append((ch["name"], ch["L"], ch["pos"], ch["mapq"], ch["flag"], (c[:, 0::2] << 4) | c[:, 1::2], ch["qual"]))
            assert bam_in[-1][5].shape[1] == half
        bam, t_write = e2e_bench.write_bam(d, "rp", bam_in, l_seq=READ_LEN)
        formats.write_annotation(os.path.join(d, "a.cga"), [(ch["name"], ch["L"], ch["gc"], ch["sat_s"], ch["sat_e"]) for ch in chroms])
        write_fasta_fast(os.path.join(d, "ref.fa"), [(ch["name"], ch["ref"]) for ch in chroms])
        synth.write_bed(os.path.join(d, "dels.bed"), [(ch["name"], s, e) for ch in chroms for s, e in zip(ch["ds"], ch["de"])])
        synth.write_bed(os.path.join(d, "dups.bed"), [(ch["name"], s, e) for ch in chroms for s, e in zip(ch["us"], ch["ue"])])
        common = ["--ref", "ref.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed", "--rp", "10"]
        e2e = {}
        k_gpu = int(os.environ.get("CONGA_BENCH_RP_K", "8"))   # (the switch: a longer or shorter cohort)
        for decode, envx, k in (("gpu", dict(CONGA_GPU_BAM="1"), k_gpu), ("host", dict(CONGA_GPU_BAM="0"), 2)):
            t1, per, t_k, err = e2e_bench.cohort_times(d, [bam], k, common, dict(envx, CONGA_TIMING="1"), decode, repeats=2 if decode == "gpu" else 1)
            # (one call per sample; one per chromosome when a sample's stretch of the file is beyond the engine's piece limit)
            n_calls = err.count("conga_reads_bgzf:")
            assert "decoding on the host" not in err and (n_calls >= k if decode == "gpu" else n_calls == 0), err[-1500:]
            e2e[decode] = dict(decode=decode, first_sample_s=round(t1, 3), per_further_sample_ms=round(per, 1), samples=k, wall_s=round(t_k, 3),
                               intervals_per_s=round(n_iv / (per * 1e-3), 1), records_per_s=round(n_reads / (per * 1e-3), 1),
                               gpu_decode_calls_per_sample=n_calls // k)
            if decode == "gpu":   # what the engine says about the last sample's BAM stage (CONGA_TIMING)
                e2e[decode]["bam_stage_of_the_last_sample"] = [ln.split("] ", 1)[-1] for ln in err.splitlines()
                                                               if "conga_reads_bgzf:" in ln or "overlapped upload:" in ln][-2:]
        for kind in ("svs", "dels", "dups"):
            a = open(os.path.join(d, "gpu_s1_%s.bed" % kind), "rb").read()
            assert a == open(os.path.join(d, "host_s1_%s.bed" % kind), "rb").read() and len(a) > 100, kind
        # READ_PAIR of every row = the support the staged route of the C-ABI counted
        at = {"dels": 0, "dups": 1}
        for kind, col in (("dels", "border_rp"), ("dups", "rp")):
            rows = open(os.path.join(d, "gpu_s1_%s.bed" % kind)).read().splitlines()[1:]
            want = np.concatenate([r[at[kind]][col] for r in res])
            assert np.array_equal(np.array([int(x.split("\t")[5]) for x in rows], np.int32), want), kind
        per = e2e["gpu"]["per_further_sample_ms"] * 1e-3
        out.update(ms_per_step=round(1e3 * per, 2), value=round(n_iv / per, 1), unit="intervals/s", records_per_s=round(n_reads / per, 1),
                   regime="BGZF bytes -> three output files, per further sample of `conga --cohort --rp 10` (%.2f GB BAM written by "
                          "tools/bamwrite in %.1f s, in the page cache): upload, inflate, record walk, split-read stage on the records "
                          "in place, depth / likelihood path, output" % (os.path.getsize(bam) / 1e9, t_write),
                   end_to_end=e2e["gpu"], end_to_end_host_decoders=e2e["host"],
                   checked="three files byte-identical between the decoders; READ_PAIR of every row equal to the staged route's support")
    except (OSError, MemoryError) as e:   # (no room for the scratch BAM, say)
        out.update(ms_per_step=round(1e3 * t_with, 3), value=round(n_iv / t_with, 1), unit="intervals/s",
                   regime="records resident in HBM (the BAM route could not run: %s: %s)" % (type(e).__name__, e))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    # a bounded sample for the caller's CPU baseline (bench.py runs the oracle; nothing in this package does)
    ch = min(chroms, key=lambda c: c["L"])
    k = min(20_000, len(ch["pos"]))
    sample = dict(name=ch["name"], ref=ch["ref"].tobytes(), sat_s=ch["sat_s"], sat_e=ch["sat_e"], pos=ch["pos"][:k], mapq=ch["mapq"][:k],
                  flag=ch["flag"][:k], lq=np.full(k, READ_LEN, np.int32), off=np.arange(k, dtype=np.uint64) * READ_LEN,
                  codes=ch["codes"][:k].reshape(-1).copy(), qual=ch["qual"][:k].reshape(-1).copy())
    return out, sample
