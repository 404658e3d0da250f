"""Randomised differential soak: HIP path (through the C-ABI) against the CPU oracle on adversarial small cases.

Not collected by pytest (run it by hand on a GPU box: `python tests/soak.py --cases 300 --seed 1`).  The cases are
not the bench generator's: chromosome lengths that are not multiples of the window, GC bytes over the whole 0..100
range, reads piled up on single bases / on the first and last base / out of range, any mapq threshold, intervals of
any length (zero, one base, nested, identical, whole chromosome, past the end), mappability rows sorted, abutting,
overlapping or shuffled, several chromosomes per batch, both formulations.  Checks are the parity tests' own
(`tests/test_gpu_parity.py: compare`): integers and expected_rd bit-exact, log-likelihoods within 1e-6.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_case(rng, step=None, mq=None):
    if step is None:
        step = int(rng.choice([100, 100, 100, 100, 64, 7, 1000]))
    L = int(rng.choice([rng.integers(1, 300), rng.integers(300, 20_000), rng.integers(20_000, 400_000),
                        rng.integers(400_000, 3_000_000)], p=[0.1, 0.3, 0.4, 0.2]))
    n_win = (L + step - 1) // step
    kind = rng.integers(0, 5)
    if kind <= 1:
        gc = rng.integers(0, 101, n_win).astype(np.uint8)
    elif kind == 2:
        gc = np.full(n_win, int(rng.integers(0, 101)), np.uint8)       # one bin only
    else:
        walk = np.cumsum(rng.normal(0, 1.5, n_win)) + rng.integers(25, 60)
        gc = np.clip(np.round(walk), 0, 100).astype(np.uint8)
        if n_win > 20 and rng.random() < 0.5:                          # a gap
            a = int(rng.integers(0, n_win - 10))
            gc[a:a + int(rng.integers(1, n_win - a))] = 0
    gc_like = None
    if rng.random() < 0.15:
        gc_like = gc.copy()
        gc_like[-1] = rng.integers(0, 101)

    # reads
    n = int(rng.choice([0, rng.integers(1, 50), rng.integers(50, 5000), rng.integers(5000, 400_000)], p=[0.05, 0.15, 0.4, 0.4]))
    mode = rng.integers(0, 4)
    if mode == 0:
        pos = rng.integers(0, L, n)
    elif mode == 1:                                                    # clustered: heavy pile-ups
        centers = rng.integers(0, L, max(1, n // 500 + 1))
        pos = np.clip(rng.choice(centers, n) + rng.integers(-3, 4, n), 0, L - 1)
    elif mode == 2:                                                    # everything on a few bases, ends included
        pos = rng.choice(np.array([0, L - 1, L // 2, min(L - 1, step), max(0, L - step)]), n)
    else:
        pos = (rng.random(n) ** 2 * L).astype(np.int64)
    pos = np.sort(pos).astype(np.int32)
    mapq = rng.choice([rng.integers(0, 256, n), np.full(n, 60), rng.choice([0, 60], n)]).astype(np.uint8) if n else np.zeros(0, np.uint8)
    if mq is None:
        mq = int(rng.choice([-1, -1, 0, 1, 20, 59, 60, 254, 255]))

    def intervals(k):
        s = rng.integers(0, max(L, 1), k)
        ln = rng.choice([rng.integers(0, 3, k), rng.integers(1, 2 * step + 2, k), rng.integers(1000, 20_000, k),
                         rng.integers(1, max(2, L), k), np.full(k, L)])
        if ln.sum() > 3e7:                                             # the oracle walks every base
            ln = ln % 20_000
        e = s + ln
        if rng.random() < 0.5:
            e = np.minimum(e, L + int(rng.integers(0, 3 * step)))
        e = np.minimum(e, 2**31 - 2 - 2 * step)
        if k > 3 and rng.random() < 0.5:                               # identical and nested rows
            s[1], e[1] = s[0], e[0]
            s[2], e[2] = s[0], max(s[0], e[0] - 1)
        s[rng.random(k) < 0.1] = 0
        order = np.lexsort((e, s))
        return s[order].astype(np.int32), e[order].astype(np.int32)

    ds, de = intervals(int(rng.choice([0, 1, 5, 40, 300])))
    us, ue = intervals(int(rng.choice([0, 0, 1, 8, 60])))

    rows = None
    if rng.random() < 0.5:
        m = int(rng.choice([0, 1, 10, 500, 5000]))
        vals = rng.choice(np.array([1, 0.5, 0.333333, 0.25, 0.2, 0.1, 0.0, 0.7731], np.float32), m)
        layout = rng.integers(0, 4)
        if layout == 0 and m:                                          # bedGraph: sorted, abutting (shared end points)
            cuts = np.sort(rng.integers(0, L, m + 1))
            ms, me = cuts[:-1], cuts[1:]
        elif layout == 1 and m:                                        # sorted, disjoint with holes
            cuts = np.sort(rng.integers(0, L, 2 * m))
            ms, me = cuts[0::2], np.maximum(cuts[0::2], cuts[1::2] - 1)
        else:                                                          # any order, overlapping
            ms = rng.integers(0, L, m)
            me = np.minimum(ms + rng.integers(0, max(2, L // 4), m), L - 1)
        rows = (ms.astype(np.int32), me.astype(np.int32), vals.astype(np.float32))
    return dict(L=L, step=step, gc=gc, gc_like=gc_like, pos=pos, mapq=mapq, mq=mq, ds=ds, de=de, us=us, ue=ue, rows=rows)


CODE = {65: 1, 67: 2, 71: 4, 84: 8}  # BAM 4-bit codes of A C G T; everything else 15 (N)


def random_split_case(rng):
    """Reference + whole BAM records for the --rp path (split_read.c:31-466, bam_data.c:29-154, likelihood.c:41-94)."""
    L = int(rng.integers(30_000, 500_000))
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    for _ in range(int(rng.integers(0, 4))):                        # low-complexity stretches: crowded 10-mer buckets
        a = int(rng.integers(0, L - 2000))
        n = int(rng.integers(200, 20_000))
        ref[a:a + n] = rng.choice(np.frombuffer(rng.choice([b"AC", b"AT", b"A", b"ACG"]), np.uint8), len(ref[a:a + n]))
    for _ in range(int(rng.integers(0, 6))):                        # repeats (within and beyond the 100 kb look-ahead)
        n = int(rng.integers(50, 3000))
        a, b = int(rng.integers(0, L - n)), int(rng.integers(0, L - n))
        ref[b:b + n] = ref[a:a + n]
    comp = np.arange(256, dtype=np.uint8)
    comp[[65, 67, 71, 84]] = [84, 71, 67, 65]
    for _ in range(int(rng.integers(0, 5))):                        # inverted repeats: mappings of the reverse complement
        n = int(rng.integers(50, 2000))
        a = int(rng.integers(0, L - n))
        b = int(np.clip(a + rng.integers(-150_000, 150_000), 0, L - n)) if rng.random() < 0.8 else int(rng.integers(0, min(L - n, 300)))
        ref[b:b + n] = comp[ref[a:a + n][::-1]]
    for _ in range(int(rng.integers(0, 3))):
        a = int(rng.integers(0, L - 500))
        ref[a:a + int(rng.integers(1, 400))] = ord("N")
    lower = ref.copy()
    a = int(rng.integers(0, L - 100))
    lower[a:a + int(rng.integers(1, 5000))] |= 0x20

    def ivs(k, lo, hi):
        s = np.sort(rng.integers(100, max(101, L - hi - 200), k))
        return [(int(x), int(x + rng.integers(lo, hi))) for x in s]
    dels = ivs(int(rng.integers(0, 6)), 1000, 8000)
    dups = ivs(int(rng.integers(1, 6)), 1000, 20_000)
    reads = []

    def add(pos, bases, mapq=60, flag=0, q=30):
        if 0 <= pos and len(bases) > 0:
            # base qualities: one value, or anything per base (the half-read means are sequential float sums whose
            # accumulator is not reset between the halves, and they are compared with the mapq threshold)
            quals = np.full(len(bases), q, np.uint8) if rng.random() < 0.4 else rng.integers(0, int(rng.choice([42, 61, 94])), len(bases)).astype(np.uint8)
            reads.append((int(pos), np.asarray(bases, np.uint8), mapq, flag, quals))
    for (s0, e0) in dels:
        for k in range(int(rng.integers(20, 40)), 80, int(rng.integers(2, 9))):
            add(s0 - k, np.concatenate([ref[s0 - k:s0], ref[e0:e0 + 100 - k]]))
    for (s0, e0) in dups:
        for k in range(int(rng.integers(25, 45)), 75, int(rng.integers(2, 9))):
            add(e0 - k, np.concatenate([ref[e0 - k:e0], ref[s0:s0 + 100 - k]]))
    for _ in range(int(rng.integers(0, 6000))):
        l = int(rng.choice([100, 100, 101, 76, 70, 59, 60, 61, 151, 36, 250, 111, 113, 225, 400, 1021, 1022]))
        p = int(rng.integers(0, max(1, L - l - 1)))
        b = ref[p:p + l].copy()
        if rng.random() < 0.3:
            idx = rng.integers(0, l, int(rng.integers(1, 6)))
            b[idx] = rng.choice(np.frombuffer(b"ACGTN", np.uint8), len(idx))
        add(p, b, int(rng.choice([60, 60, 60, 0, 17, 40, 255])), int(rng.choice([0, 0, 0, 0, 0x400, 0x100, 0x800, 0x200, 16, 4])),
            int(rng.choice([30, 30, 12, 2, 40])))
    add(0, ref[0:100])
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
    ns = int(rng.integers(0, 5))
    sat_s = rng.integers(0, L - 10, ns).astype(np.int32)
    sat_e = (sat_s + rng.integers(1, 5000, ns)).astype(np.int32)
    return dict(L=L, ref=bytes(ref), ref_lower=bytes(lower), dels=dels, dups=dups, pos=pos, mapq=mapq, flag=flag, lq=lq,
                off=off, codes=codes, qual=qual, sat_s=sat_s, sat_e=sat_e)


def describe(c):
    return ("L=%d step=%d reads=%d mq=%d dels=%d dups=%d rows=%s gc_like=%s" % (
        c["L"], c["step"], len(c["pos"]), c["mq"], len(c["ds"]), len(c["us"]),
        None if c["rows"] is None else len(c["rows"][0]), c["gc_like"] is not None))


def bam_case(rng, d):
    """One random BAM case of `--bam` written into directory d (r.bam + .bai, a.cga, dels.bed, dups.bed) -> the command line's
    arguments.  The draws are the sequence's: np.random.default_rng([seed, 11_000_000 + case]) gives case `case` of seed `seed` again
    (tests/test_host_cli.py replays seed 81, case 38 -- the cohort that stood still on round 3's last day)."""
    import zlib
    from conga_amd import formats, synth
    n_chr = int(rng.integers(1, 6))
    chroms, reads = [], []
    for k in range(n_chr):
        L = int(rng.choice([rng.integers(20_000, 100_000), rng.integers(100_000, 2_000_000)]))
        c = synth.make_chrom(str(k + 1), L, cov=float(rng.choice([0.0, 0.3, 1.0, 4.0])), n_dels=int(rng.integers(0, 30)),
                             n_dups=int(rng.integers(0, 8)), gaps=bool(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)))
        pos = c.pos
        if len(pos) and rng.random() < 0.5:   # pile-ups on and around window boundaries of the index
            w = int(rng.integers(1, max(2, L >> 14))) << 14
            extra = np.concatenate([np.full(int(rng.integers(1, 500)), min(w, L - 1)), np.full(int(rng.integers(1, 500)), max(w - 1, 0))])
            pos = np.sort(np.concatenate([pos, extra])).astype(np.int32)
        if len(pos) and rng.random() < 0.4:   # a stretch without reads
            lo = int(rng.integers(0, L))
            pos = pos[(pos < lo) | (pos > lo + int(rng.integers(1, 200_000)))]
        chroms.append(c)
        reads.append((c.name, L, pos, rng.integers(0, 61, len(pos)).astype(np.uint8)))
    formats.write_bam(os.path.join(d, "r.bam"), "S", reads, index=True, unplaced=int(rng.integers(0, 20)),
                      block_payload=int(rng.choice([257, 1500, 9000, 40000, 65280])), level=int(rng.integers(0, 10)),
                      strategy=int(rng.choice([0, 0, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])))
    order = rng.permutation(n_chr)              # the annotation lists the chromosomes in its own order
    formats.write_annotation(os.path.join(d, "a.cga"), [(chroms[j].name, chroms[j].length, chroms[j].gc, [], []) for j in order])
    synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s_, e_) for c in chroms for s_, e_ in zip(c.del_start, c.del_end)])
    synth.write_bed(os.path.join(d, "dups.bed"), [(c.name, s_, e_) for c in chroms for s_, e_ in zip(c.dup_start, c.dup_end)])
    args = ["-i", "r.bam", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed"]
    if rng.random() < 0.5:
        args += ["--min-mapq", str(int(rng.integers(0, 60)))]
    if n_chr > 1 and rng.random() < 0.3:
        args += ["--first-chr", str(int(rng.integers(0, n_chr))), "--last-chr", str(n_chr - 1)]
    if rng.random() < 0.3:
        args += ["--gpus", str(int(rng.integers(2, 4)))]
    return args


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--first-case", type=int, default=0, help="--bam / --bam-rp: begin with this case of the seed's sequence (a failure names its case)")
    ap.add_argument("--seconds", type=float, default=0, help="stop after this long (0: run all cases)")
    ap.add_argument("--bam", action="store_true",
                    help="the command line on random BAM files (+ .bai): decode on the GPU (conga_reads_bgzf) against the host "
                         "decoders -- same three output files, same read counts")
    ap.add_argument("--split-reads", action="store_true", help="the --rp path: random references and whole BAM records")
    ap.add_argument("--packed", action="store_true",
                    help="cohort hand-overs: random samples behind random layouts through conga_sample_reads and through "
                         "conga_sample_reads_packed at every width (exceptions separate and inline, beside the previous compute): same records")
    ap.add_argument("--ahead", action="store_true",
                    help="two computes in flight (ABI v9): random layouts, random sequences of samples (pile-ups that wrap a `short`, empty "
                         "chromosomes, 32-bit and packed hand-overs) through conga_chrom_compute_ahead / conga_sample_fetch_previous in "
                         "random call orders: every sample's records == the records of that sample computed alone")
    ap.add_argument("--bam-rp", action="store_true",
                    help="`conga --rp` on random BAMs with sequences: records mapped in place after the decode on the GPU (one call, and "
                         "one per chromosome) against the host decoders handing them over -- same files, same split-read counts")
    ap.add_argument("--chroms-per-batch", type=int, default=0, help="with --batch: this many chromosomes in every batch (default: 1-12)")
    ap.add_argument("--batch", action="store_true",
                    help="CONGA_FLAG_BATCH: 1..12 random chromosomes per context (one launch per kernel over all of them), "
                         "computed twice, with random chain-class thresholds")
    a = ap.parse_args()
    import torch  # noqa: F401  (before the library: it ships its own HIP runtime)
    from conga_amd import capi
    from oracle import oracle as O
    import test_gpu_parity as T

    t0 = time.time()
    done = 0
    if a.bam:
        ahead_total = [0, 0]
        import re
        import subprocess
        import tempfile
        import zlib
        from conga_amd import formats, synth
        conga = os.path.join(ROOT, "conga_amd", "host", "conga")
        for i in range(a.first_case, a.first_case + a.cases):
            rng = np.random.default_rng([a.seed, 11_000_000 + i])
            d = tempfile.mkdtemp(prefix="conga_soak_bam_")
            args = bam_case(rng, d)
            outs = {}
            for tag, env in (("gpu", {"CONGA_GPU_BAM": "1"}), ("host", {"CONGA_GPU_BAM": "0", "CONGA_BAM_SEGMENTS": str(int(rng.integers(1, 9)))})):
                r = subprocess.run([conga] + args + ["--out", tag], cwd=d, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
                if r.returncode != 0 or (tag == "gpu" and "decoding on the host" in r.stderr):
                    print("FAILED bam case %d (seed %d, %s) in %s:\n%s" % (i, a.seed, tag, d, r.stderr[-1500:]), flush=True)
                    raise SystemExit(1)
                files = [open(os.path.join(d, "%s_%s.bed" % (tag, k)), "rb").read() for k in ("svs", "dels", "dups")]
                outs[tag] = (files, re.findall(r"\((\d+) reads, 0 split-reads\)", r.stderr))
            if outs["gpu"] != outs["host"]:
                print("FAILED bam case %d (seed %d): GPU and host decode differ, files in %s" % (i, a.seed, d), flush=True)
                raise SystemExit(1)
            if "--gpus" not in args:
                # ... and as a cohort of four: the engine kept, every further sample's bytes named ahead, its block table read off
                # them by the engine (checked against the file's: CONGA_BGZF_CHECK_TABLE) and inflated ahead; pieces of any size,
                # also far smaller than a BGZF block
                with open(os.path.join(d, "list.txt"), "w") as f:
                    f.write("".join("r.bam\tc%d\n" % k for k in range(4)))
                env = dict(os.environ, CONGA_GPU_BAM="1", CONGA_BGZF_OVERLAP="1", CONGA_BGZF_CHECK_TABLE="1", CONGA_TIMING="1",
                           CONGA_BGZF_PIECE_KB=str(int(rng.choice([4, 6, 16, 64, 512, 8192]))))
                print("bam case %d: cohort of four, pieces of %s KB" % (i, env["CONGA_BGZF_PIECE_KB"]), flush=True)
                try:
                    r = subprocess.run([conga, "--cohort", "list.txt", "--out", "co"] + args[2:], cwd=d, capture_output=True, text=True, timeout=90, env=env)
                except subprocess.TimeoutExpired as e:
                    print("FAILED bam case %d (seed %d, cohort): no end after 90 s, files in %s\n%s" % (
                        i, a.seed, d, (e.stderr or b"")[-3000:].decode("utf-8", "replace") if isinstance(e.stderr, bytes) else str(e.stderr)[-3000:]), flush=True)
                    raise SystemExit(1)
                if r.returncode != 0 or "decoding on the host" in r.stderr:
                    print("FAILED bam case %d (seed %d, cohort) in %s:\n%s" % (i, a.seed, d, r.stderr[-2500:]), flush=True)
                    raise SystemExit(1)
                for k in range(4):
                    files = [open(os.path.join(d, "c%d_%s.bed" % (k, kind)), "rb").read() for kind in ("svs", "dels", "dups")]
                    if files != outs["gpu"][0]:
                        print("FAILED bam case %d (seed %d): sample %d of the cohort differs from the single run, files in %s" % (i, a.seed, k, d), flush=True)
                        raise SystemExit(1)
                ahead_total[0] += r.stderr.count("named ahead with its block table")
                ahead_total[1] += r.stderr.count("the engine's and the file's agree")
            import shutil
            shutil.rmtree(d)
            done += 1
            if i % 10 == 9:
                print("%d bam cases ok, %.0f s" % (done, time.time() - t0), flush=True)
            if a.seconds and time.time() - t0 > a.seconds:
                break
        print("soak: %d random BAMs: decode on the GPU == host decoders == every sample of a cohort of four (%d samples inflated ahead, %d block "
              "tables read by the engine and checked) (seed %d, %.0f s)" % (done, ahead_total[0], ahead_total[1], a.seed, time.time() - t0))
        return
    if a.packed:
        from conga_amd import synth
        for i in range(a.cases):
            rng = np.random.default_rng([a.seed, 15_000_000 + i])
            n_chr = int(rng.integers(1, 6))
            chroms = []
            for k in range(n_chr):
                L = int(rng.choice([rng.integers(5_000, 100_000), rng.integers(100_000, 3_000_000)]))
                c = synth.make_chrom(str(k + 1), L, cov=0.0, n_dels=int(rng.integers(1, 40)), n_dups=int(rng.integers(0, 10)), gaps=bool(rng.integers(0, 2)),
                                     seed=int(rng.integers(1, 1 << 30)))
                chroms.append((c, *synth.kept_sorted(c.del_start, c.del_end), *synth.kept_sorted(c.dup_start, c.dup_end)))
            mq = int(rng.choice([-1, -1, 0, 30]))
            with capi.Context(device=0, mq_threshold=mq, flags=capi.FLAG_BATCH) as ctx:
                for c, ds, de, us, ue in chroms:
                    ctx.chrom_begin(c.length, c.gc)
                    ctx.intervals("D", ds, de)
                    ctx.intervals("E", us, ue)
                for smp in range(int(rng.integers(1, 4))):
                    reads = []
                    for c, *_r in chroms:
                        cov = float(rng.choice([0.0, 0.02, 0.5, 1.0, 8.0]))
                        p, m = synth.make_reads(c.length, c.gc, c.step, cov, 100, rng) if cov else (np.zeros(0, np.int32), np.zeros(0, np.uint8))
                        if len(p) and rng.random() < 0.3:       # pile-ups and long gaps
                            p = np.sort(np.concatenate([p, np.full(int(rng.integers(1, 300)), int(rng.integers(0, c.length)))])).astype(np.int32)
                            lo = int(rng.integers(0, c.length))
                            p = p[(p < lo) | (p > lo + int(rng.integers(1, 300_000)))]
                            m = rng.integers(0, 61, len(p)).astype(np.uint8)
                        reads.append((p, m))
                    n = sum(len(p) for p, _ in reads)
                    pos = np.concatenate([p for p, _ in reads]).astype(np.int32) if n else np.zeros(0, np.int32)
                    mapq = np.concatenate([m for _, m in reads]).astype(np.uint8) if n else np.zeros(0, np.uint8)
                    off = np.concatenate([[0], np.cumsum([len(p) for p, _ in reads])]).astype(np.uint64)
                    pos_a, mapq_a = np.concatenate([pos, np.zeros(1, np.int32)]), np.concatenate([mapq, np.zeros(1, np.uint8)])
                    ctx.sample_reads(pos_a, mapq_a, off)
                    ctx.compute()
                    want, wantE, _ = ctx.sample_fetch()
                    for width in (5, 7, 8, 9, 10, 11, 12, 14, 16, None):
                        bits, w, ei, ep = capi.encode_packed(pos, off, width)
                        if rng.random() < 0.5:
                            ctx.sample_reads_packed(capi.pack_inline(bits, ei, ep), w, len(ei), None, mapq_a, off)
                        else:
                            ctx.sample_reads_packed(np.concatenate([bits, np.zeros(32, np.uint8)]), w, ei, ep, mapq_a, off)
                        ctx.compute()
                        got, gotE, _ = ctx.sample_fetch()
                        if got.tobytes() != want.tobytes() or gotE.tobytes() != wantE.tobytes():
                            print("FAILED packed case %d (seed %d): sample %d width %s, %d chromosomes, %d reads" % (i, a.seed, smp, width, n_chr, n), flush=True)
                            raise SystemExit(1)
            done += 1
            if i % 10 == 9:
                print("%d packed cases ok, %.0f s" % (done, time.time() - t0), flush=True)
            if a.seconds and time.time() - t0 > a.seconds:
                break
        print("soak: %d layouts: conga_sample_reads_packed at every width == conga_sample_reads (seed %d, %.0f s)" % (done, a.seed, time.time() - t0))
        return
    if a.ahead:
        from conga_amd import synth
        wraps = 0
        for i in range(a.first_case, a.first_case + a.cases):
            rng = np.random.default_rng([a.seed, 16_000_000 + i])
            n_chr = int(rng.integers(1, 5))
            chroms = []
            for k in range(n_chr):
                L = int(rng.choice([rng.integers(5_000, 100_000), rng.integers(100_000, 2_000_000)]))
                c = synth.make_chrom(str(k + 1), L, cov=0.0, n_dels=int(rng.integers(1, 40)), n_dups=int(rng.integers(0, 10)), gaps=bool(rng.integers(0, 2)),
                                     seed=int(rng.integers(1, 1 << 30)))
                chroms.append((c, *synth.kept_sorted(c.del_start, c.del_end), *synth.kept_sorted(c.dup_start, c.dup_end)))
            n_smp = int(rng.integers(2, 8))
            samples = []
            for smp in range(n_smp):
                reads = []
                for c, *_r in chroms:
                    cov = float(rng.choice([0.0, 0.02, 0.5, 1.0, 4.0]))
                    p, m = synth.make_reads(c.length, c.gc, c.step, cov, 100, rng) if cov else (np.zeros(0, np.int32), np.zeros(0, np.uint8))
                    if rng.random() < 0.25:                   # a pile-up that wraps the reference's `short` (or stays just below the guard)
                        p = np.sort(np.concatenate([p, np.full(int(rng.choice([20_000, 33_000, 40_000, 70_000])), int(rng.integers(0, c.length)), np.int32)])).astype(np.int32)
                        m = np.full(len(p), 60, np.uint8)
                    reads.append((p, m))
                n = sum(len(p) for p, _ in reads)
                pos = np.concatenate([p for p, _ in reads] + [np.zeros(1, np.int32)]).astype(np.int32)
                mapq = np.concatenate([m for _, m in reads] + [np.zeros(1, np.uint8)]).astype(np.uint8)
                off = np.concatenate([[0], np.cumsum([len(p) for p, _ in reads])]).astype(np.uint64)
                kind = str(rng.choice(["int32", "packed", "inline"]))
                bits, w, ei, ep = capi.encode_packed(pos[:n], off, None if rng.random() < 0.5 else int(rng.integers(5, 17)))
                samples.append(dict(pos=pos, mapq=mapq, off=off, kind=kind, bits=np.concatenate([bits, np.zeros(32, np.uint8)]), w=w, ei=ei, ep=ep,
                                    inline=capi.pack_inline(bits, ei, ep)))

            def lay_out(ctx):
                for c, ds, de, us, ue in chroms:
                    ctx.chrom_begin(c.length, c.gc)
                    ctx.intervals("D", ds, de)
                    ctx.intervals("E", us, ue)

            def hand_over(ctx, sm):
                if sm["kind"] == "int32":
                    ctx.sample_reads(sm["pos"], sm["mapq"], sm["off"])
                elif sm["kind"] == "packed":
                    ctx.sample_reads_packed(sm["bits"], sm["w"], sm["ei"], sm["ep"], sm["mapq"], sm["off"])
                else:
                    ctx.sample_reads_packed(sm["inline"], sm["w"], len(sm["ei"]), None, sm["mapq"], sm["off"])

            mq = int(rng.choice([-1, -1, 0, 30]))
            want, wrapping = [], []
            with capi.Context(device=0, mq_threshold=mq, flags=capi.FLAG_BATCH) as ctx:   # every sample alone, one after the other
                lay_out(ctx)
                for sm in samples:
                    ctx.sample_reads(sm["pos"], sm["mapq"], sm["off"])
                    ctx.compute()
                    r, E, st = ctx.sample_fetch(want_stats=True)
                    wraps += int(any(x.depth_materialized for x in st))
                    wrapping.append(int(any(x.depth_materialized for x in st)))
                    want.append((r.tobytes(), E.tobytes()))
            with capi.Context(device=0, mq_threshold=mq, flags=capi.FLAG_BATCH) as ctx:
                lay_out(ctx)
                got = {}
                calls = []

                def take(k, previous):
                    r, E, _ = ctx.sample_fetch_previous() if previous else ctx.sample_fetch()
                    got.setdefault(k, []).append((r.tobytes(), E.tobytes()))
                    calls.append("fetch%s(%d)" % ("_previous" if previous else "", k))

                hand_over(ctx, samples[0])
                ctx.compute()
                for k in range(n_smp):
                    if k + 1 == n_smp:
                        take(k, False)
                        break
                    order = int(rng.integers(0, 5))
                    calls.append("[order %d]" % order)
                    if order == 4:                            # round 4's order now and then: hand over, fetch, compute
                        hand_over(ctx, samples[k + 1])
                        take(k, False)
                        ctx.compute()
                        continue
                    hand_over(ctx, samples[k + 1])
                    ctx.compute_ahead()
                    if order == 1:
                        take(k + 1, False)                    # the latest one first (it is fetched again in its own turn)
                    if order == 2:
                        ctx.sync_previous()
                    if order == 3 and k + 2 < n_smp:
                        # the next sample handed over BEFORE the older one is fetched: the hand-over settles its guard; the sample
                        # handed over is dropped again by the hand-over of the next turn (two without a compute: the later one counts)
                        hand_over(ctx, samples[k + 2])
                        take(k, True)
                        hand_over(ctx, samples[k + 1])
                        ctx.compute()                         # (sample k + 1 once more: a plain compute, the older results are given up)
                        continue
                    take(k, True)
                for k in range(n_smp):
                    for g in got.get(k, []):
                        if g != want[k]:
                            print("FAILED ahead case %d (seed %d): sample %d of %d (%s), %d chromosomes; records %s, tables %s; samples that wrap: %s; %s"
                                  % (i, a.seed, k, n_smp, samples[k]["kind"], n_chr, "differ" if g[0] != want[k][0] else "equal", "differ" if g[1] != want[k][1] else "equal",
                                     wrapping, " ".join(calls)), flush=True)
                            raise SystemExit(1)
                    if k not in got:
                        print("FAILED ahead case %d (seed %d): sample %d was never fetched" % (i, a.seed, k), flush=True)
                        raise SystemExit(1)
            done += 1
            if i % 10 == 9:
                print("%d ahead cases ok, %.0f s" % (done, time.time() - t0), flush=True)
            if a.seconds and time.time() - t0 > a.seconds:
                break
        print("soak: %d layouts, two computes in flight in random call orders == every sample computed alone (%d samples needed the dense kernels) (seed %d, %.0f s)"
              % (done, wraps, a.seed, time.time() - t0))
        return
    if a.bam_rp:
        import re
        import shutil
        import subprocess
        import tempfile
        from conga_amd import formats, synth
        conga = os.path.join(ROOT, "conga_amd", "host", "conga")
        for i in range(a.first_case, a.first_case + a.cases):
            rng = np.random.default_rng([a.seed, 13_000_000 + i])
            d = tempfile.mkdtemp(prefix="conga_soak_bamrp_")
            n_chr = int(rng.integers(1, 4))
            cases = [random_split_case(rng) for _ in range(n_chr)]
            reads, recs, annot, fasta, dels, dups = [], {}, [], [], [], []
            for k, c in enumerate(cases):
                name = str(k + 1)
                inside = c["pos"].astype(np.int64) < c["L"]     # (a record behind the annotation's chromosome end leaves the file to the host decoders)
                keep = np.flatnonzero(inside)
                lq = c["lq"][keep]
                per_base = np.repeat(inside, c["lq"])
                off = np.concatenate([[0], np.cumsum(lq)[:-1]]).astype(np.uint64) if len(lq) else np.zeros(0, np.uint64)
                reads.append((name, c["L"], c["pos"][keep], c["mapq"][keep], c["flag"][keep]))
                recs[name] = (lq, c["codes"][per_base], c["qual"][per_base], off)
                annot.append((name, c["L"], np.full((c["L"] + 99) // 100, int(rng.integers(30, 60)), np.uint8), c["sat_s"], c["sat_e"]))
                fasta.append((name, c["ref_lower"]))
                dels += [(name, s_, e_) for s_, e_ in c["dels"]]
                dups += [(name, s_, e_) for s_, e_ in c["dups"]]
            formats.write_bam(os.path.join(d, "r.bam"), "S", reads, records=recs, index=True, unplaced=int(rng.integers(0, 6)),
                              block_payload=int(rng.choice([1500, 9000, 40000, 65280])), level=int(rng.integers(0, 10)))
            formats.write_annotation(os.path.join(d, "a.cga"), annot)
            formats.write_fasta(os.path.join(d, "ref.fa"), fasta)
            synth.write_bed(os.path.join(d, "dels.bed"), dels)
            synth.write_bed(os.path.join(d, "dups.bed"), dups if dups else [("1", 10, 2000)])
            args = ["-i", "r.bam", "--ref", "ref.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--dups", "dups.bed", "--rp", str(int(rng.integers(1, 12)))]
            if rng.random() < 0.4:
                args += ["--min-mapq", str(int(rng.choice([0, 15, 25, 35])))]
            if rng.random() < 0.3:
                args += ["--min-read-length", str(int(rng.choice([10, 75, 100])))]
            size = os.path.getsize(os.path.join(d, "r.bam"))
            outs = {}
            for tag, env in (("gpu", {"CONGA_GPU_BAM": "1"}), ("each", {"CONGA_GPU_BAM": "1", "CONGA_GPU_BAM_MAX_MB": "%.4f" % (size * 0.6 / 1048576)}),
                             ("host", {"CONGA_GPU_BAM": "0"})):
                r = subprocess.run([conga] + args + ["--out", tag], cwd=d, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
                if r.returncode != 0 or (tag == "gpu" and "decoding on the host" in r.stderr):
                    print("FAILED bam-rp case %d (seed %d, %s) in %s: %s\n%s" % (i, a.seed, tag, d, " ".join(args), r.stderr[-3000:]), flush=True)
                    raise SystemExit(1)
                files = [open(os.path.join(d, "%s_%s.bed" % (tag, k)), "rb").read() for k in ("svs", "dels", "dups")]
                outs[tag] = (files, re.findall(r"\((\d+) reads, (\d+) split-reads\)", r.stderr), re.findall(r"CONGA paired (\d+)", r.stderr))
            if not (outs["gpu"] == outs["host"] == outs["each"]):
                print("FAILED bam-rp case %d (seed %d): the routes differ, files in %s" % (i, a.seed, d), flush=True)
                raise SystemExit(1)
            shutil.rmtree(d)
            done += 1
            if i % 10 == 9:
                print("%d bam-rp cases ok, %.0f s" % (done, time.time() - t0), flush=True)
            if a.seconds and time.time() - t0 > a.seconds:
                break
        print("soak: %d random --rp BAMs: records in place (one call / per chromosome) == host decoders (seed %d, %.0f s)" % (done, a.seed, time.time() - t0))
        return
    if a.split_reads:
        import test_gpu_split_reads as S
        for i in range(a.cases):
            rng = np.random.default_rng([a.seed, 9_000_000 + i])
            c = random_split_case(rng)
            mq, min_len = int(rng.choice([-1, 0, 15, 20, 25, 30, 35, 45])), int(rng.choice([60, 60, 10, 75, 100]))
            try:
                (dels, dups, st), (od, ou, rows, counts) = S.run_both(capi, O, c, mq, min_len)
                assert (st.split_elements, st.split_mappings, st.split_del_rows, st.split_dup_rows) == tuple(int(x) for x in counts)
                assert np.array_equal(dels["border_rp"], od["border_rp"]) and np.all(dels["rp"] == 0)
                assert np.array_equal(dups["rp"], ou["rp"]) and np.all(dups["border_rp"] == 0)
            except Exception:
                print("FAILED split-read case %d (seed %d): L=%d reads=%d dels=%d dups=%d mq=%d min_len=%d" % (
                    i, a.seed, c["L"], len(c["pos"]), len(c["dels"]), len(c["dups"]), mq, min_len), flush=True)
                raise
            done += 1
            if i % 10 == 9:
                print("%d split-read cases ok, %.0f s" % (done, time.time() - t0), flush=True)
            if a.seconds and time.time() - t0 > a.seconds:
                break
        print("soak: %d split-read cases agree with the oracle (seed %d, %.0f s)" % (done, a.seed, time.time() - t0))
        return
    for i in range(a.cases if a.batch else 0):
        rng = np.random.default_rng([a.seed, 7_000_000 + i])
        step = int(rng.choice([100, 100, 100, 64, 1000]))
        mq = int(rng.choice([-1, -1, 0, 20, 60]))
        cases = [random_case(rng, step, mq) for _ in range(a.chroms_per_batch or int(rng.integers(1, 13)))]
        knobs = {}
        if rng.random() < 0.7:  # class boundaries of the chain kernel (conga_api.hip: prepare)
            knobs = dict(CONGA_CHAIN_SERIAL_WINDOWS=str(int(rng.choice([0, 3, 20, 56, 200]))),
                         CONGA_CHAIN_LONG_WINDOWS=str(int(rng.choice([4, 64, 512, 4000]))),
                         CONGA_CHAIN_BLOCK_WINDOWS=str(int(rng.choice([64, 700, 2048, 100000]))))
        os.environ.update(knobs)
        try:
            for dense in (0, 1):
                capi.EXTRA_FLAGS = capi.FLAG_MATERIALIZE_DEPTH if dense else 0
                with capi.Context(device=0, mq_threshold=mq, gc_step=step, flags=capi.FLAG_BATCH) as ctx:
                    for c in cases:
                        ctx.chrom_begin(c["L"], c["gc"], c["gc_like"])
                        ctx.reads(c["pos"], c["mapq"])
                        if c["rows"] is not None:
                            ctx.mappability(*c["rows"])
                        ctx.intervals("D", c["ds"], c["de"])
                        ctx.intervals("E", c["us"], c["ue"])
                    ctx.compute()
                    ctx.compute()
                    for j, c in enumerate(cases):
                        ctx.select(j)
                        dels, dups, E, st = ctx.fetch()
                        got = dict(dels=dels, dups=dups, E=E, counted=st.reads_counted, S=np.array(st.rd_per_gc[:]),
                                   W=np.array(st.window_per_gc[:]))
                        want = T.run_oracle(O, c["L"], c["gc"], c["pos"], c["mapq"], c["ds"], c["de"], c["us"], c["ue"], mq=mq,
                                            step=step, rows=c["rows"], gc_like=c["gc_like"])
                        try:
                            T.compare(got, want, c["rows"] is not None)
                        except Exception:
                            print("FAILED batch %d chromosome %d (seed %d, %s, knobs %s): %s" % (
                                i, j, a.seed, "dense" if dense else "tuple space", knobs, describe(c)), flush=True)
                            raise
        finally:
            for k in knobs:
                os.environ.pop(k, None)
        done += 1
        if i % 10 == 9:
            print("%d batches ok, %.0f s" % (done, time.time() - t0), flush=True)
        if a.seconds and time.time() - t0 > a.seconds:
            break
    for i in range(0 if a.batch else a.cases):
        rng = np.random.default_rng([a.seed, i])
        c = random_case(rng)
        want = T.run_oracle(O, c["L"], c["gc"], c["pos"], c["mapq"], c["ds"], c["de"], c["us"], c["ue"], mq=c["mq"],
                            step=c["step"], rows=c["rows"], gc_like=c["gc_like"])
        for dense in (0, 1):
            capi.EXTRA_FLAGS = capi.FLAG_MATERIALIZE_DEPTH if dense else 0
            try:
                got = T.run_gpu(capi, c["L"], c["gc"], c["pos"], c["mapq"], c["ds"], c["de"], c["us"], c["ue"], mq=c["mq"],
                                step=c["step"], rows=c["rows"], gc_like=c["gc_like"])
                T.compare(got, want, c["rows"] is not None)
            except Exception:
                print("FAILED case %d (seed %d, %s): %s" % (i, a.seed, "dense" if dense else "tuple space", describe(c)), flush=True)
                raise
        done += 1
        if i % 20 == 19:
            print("%d cases ok, %.0f s" % (done, time.time() - t0), flush=True)
        if a.seconds and time.time() - t0 > a.seconds:
            break
    capi.EXTRA_FLAGS = 0
    print("soak: %d %s x 2 formulations agree with the oracle (seed %d, %.0f s)" % (done, "batches" if a.batch else "cases", a.seed, time.time() - t0))


if __name__ == "__main__":
    main()
