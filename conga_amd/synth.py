"""Seeded synthetic inputs for the read-depth / likelihood path.

There is no network, so the 1000G Phase-3 call set, GRCh37 annotation and a real BAM cannot be
fetched; these generators produce inputs of the same shape: GRCh37 autosome lengths, a per-100-bp
GC track with N-gaps, coordinate-sorted read-start tuples with a MAPQ mix, and deletion /
duplication BED rows with a long-tailed size distribution.
"""
import zlib
from dataclasses import dataclass, field

import numpy as np
from scipy.signal import lfilter

# GRCh37 (b37 naming, README.md:67-71 of the reference)
GRCH37_AUTOSOMES = (
    ("1", 249250621), ("2", 243199373), ("3", 198022430), ("4", 191154276), ("5", 180915260),
    ("6", 171115067), ("7", 159138663), ("8", 146364022), ("9", 141213431), ("10", 135534747),
    ("11", 135006516), ("12", 133851895), ("13", 115169878), ("14", 107349540), ("15", 102531392),
    ("16", 90354753), ("17", 81195210), ("18", 78077248), ("19", 59128983), ("20", 63025520),
    ("21", 48129895), ("22", 51304566),
)
GENOME_LEN = sum(l for _, l in GRCH37_AUTOSOMES)
BASE_SEED = 20221124

N_DELS_GENOME = 42000
N_DUPS_GENOME = 6000


@dataclass
class SynthChrom:
    name: str
    length: int
    step: int
    gc: np.ndarray                      # uint8 per window (0 inside N-gaps)
    pos: np.ndarray                     # int32, sorted
    mapq: np.ndarray                    # uint8
    del_start: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    del_end: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    dup_start: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    dup_end: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    map_start: np.ndarray = None
    map_end: np.ndarray = None
    map_val: np.ndarray = None

    @property
    def n_win(self):
        return len(self.gc)


def make_gc_track(length, rng, step=100, gaps=True):
    """uint8 GC% per window: AR(1) walk, mean 41, sd 8, clipped to [20, 75]; N-gap blocks are 0."""
    n_win = (length + step - 1) // step
    noise = rng.standard_normal(n_win)
    rho = 0.995
    x = lfilter([np.sqrt(1 - rho * rho)], [1, -rho], noise)
    gc = np.clip(np.rint(41.0 + 8.0 * x), 20, 75).astype(np.uint8)
    if gaps and n_win > 2000:
        lead = int(min(n_win // 6, rng.integers(50_000, 100_000)))       # 5-10 Mb leading gap
        gc[:lead] = 0
        cen = int(rng.integers(n_win // 3, n_win // 2))
        gc[cen:cen + int(min(n_win // 20, 30_000))] = 0                     # centromere block
        gc[n_win - int(min(n_win // 100, 100)):] = 0                        # telomere
    return gc


def _place_intervals(length, gc, step, n, median, sigma, max_len, rng, frac_short=0.05):
    """(start, end) rows outside N-gaps; sizes 1000 + LogNormal(median, sigma) truncated at max_len."""
    if n == 0:
        z = np.zeros(0, np.int32)
        return z, z
    size = 1000 + np.minimum(rng.lognormal(np.log(median), sigma, n), max_len - 1000).astype(np.int64)
    short = rng.random(n) < frac_short
    size[short] = rng.integers(50, 1000, int(short.sum()))  # exercised by the min-sv-size filter
    size = np.minimum(size, max(length // 4, 1))
    ok = np.flatnonzero(gc > 0)
    if len(ok) == 0:
        ok = np.arange(len(gc))
    start = ok[rng.integers(0, len(ok), n)].astype(np.int64) * step + rng.integers(0, step, n)
    start = np.minimum(start, length - size - 1)
    start = np.maximum(start, 0)
    end = start + size
    return start.astype(np.int32), end.astype(np.int32)


def make_reads(length, gc, step, cov, readlen, rng, dels=None, dups=None, del_gt=None, dup_gt=None):
    """Coordinate-sorted read-start tuples.  Poisson starts, rate cov/readlen * f(GC); depth is
    modulated inside SVs by the truth genotype (0, 1, 2 affected alleles)."""
    n_win = len(gc)
    g = gc.astype(np.float64)
    bias = np.where(gc > 0, 1.0 - ((g - 45.0) / 60.0) ** 2, 0.0)          # mild quadratic bias, 0 in gaps
    if np.any(gc > 0):
        bias /= bias[gc > 0].mean()                                         # cov is the depth outside gaps
    rate = (cov / readlen) * bias                                            # starts per base
    factor = np.ones(n_win)
    if dels is not None and del_gt is not None:
        for s, e, gt in zip(dels[0] // step, dels[1] // step, del_gt):
            factor[s:e + 1] *= (1.0, 0.5, 0.0)[gt]
    if dups is not None and dup_gt is not None:
        for s, e, gt in zip(dups[0] // step, dups[1] // step, dup_gt):
            factor[s:e + 1] *= (1.0, 1.5, 2.0)[gt]
    bases = np.full(n_win, step, dtype=np.int64)
    bases[-1] = length - (n_win - 1) * step
    counts = rng.poisson(rate * factor * bases)
    n = int(counts.sum())
    win = np.repeat(np.arange(n_win, dtype=np.int64), counts)
    pos = win * step + (rng.random(n) * bases[win]).astype(np.int64)
    pos.sort(kind="stable")
    u = rng.random(n)
    mapq = np.where(u < 0.8, 60, np.where(u < 0.9, 0, rng.integers(1, 60, n))).astype(np.uint8)
    return pos.astype(np.int32), mapq


def make_mappability(length, rng, mean_len=150):
    """bedGraph-like rows: contiguous segments, abutting rows share an endpoint (svs.c:368 is
    end-inclusive, so the later row wins the shared base)."""
    n = int(length / mean_len * 1.2) + 16
    seg = rng.geometric(1.0 / mean_len, n).astype(np.int64)
    edges = np.concatenate([[0], np.cumsum(seg)])
    edges = edges[edges < length - 1]
    start = edges[:-1]
    end = edges[1:]
    vals = np.array([1.0, 0.5, 0.333333, 0.25, 0.2, 0.1], dtype=np.float32)
    p = np.array([0.8, 0.06, 0.05, 0.04, 0.03, 0.02])
    val = vals[rng.choice(len(vals), len(start), p=p)]
    return start.astype(np.int32), end.astype(np.int32), val


def make_chrom(name, length, *, cov=1.0, n_dels=0, n_dups=0, readlen=100, step=100, seed=BASE_SEED,
               mappability=False, gaps=True):
    rng = np.random.default_rng([seed, int(name) if name.isdigit() else zlib.crc32(name.encode())])
    gc = make_gc_track(length, rng, step, gaps)
    dels = _place_intervals(length, gc, step, n_dels, 2500.0, 1.3, 2_000_000, rng)
    dups = _place_intervals(length, gc, step, n_dups, 15000.0, 1.3, 5_000_000, rng)
    del_gt = rng.choice(3, n_dels, p=[0.45, 0.40, 0.15])
    dup_gt = rng.choice(3, n_dups, p=[0.45, 0.40, 0.15])
    pos, mapq = make_reads(length, gc, step, cov, readlen, rng, dels, dups, del_gt, dup_gt)
    c = SynthChrom(name, length, step, gc, pos, mapq, dels[0], dels[1], dups[0], dups[1])
    if mappability:
        c.map_start, c.map_end, c.map_val = make_mappability(length, rng)
    return c


def kept_sorted(start, end, min_sv_size=1000):
    """load_known_SVs filter (svs.c:55) + qsort by (start, end) (likelihood.c:324-328)."""
    keep = (end.astype(np.int64) - start) >= min_sv_size
    s, e = start[keep], end[keep]
    order = np.lexsort((e, s))
    return s[order], e[order]


def genome_plan(chroms=GRCH37_AUTOSOMES, n_dels=N_DELS_GENOME, n_dups=0):
    """Interval counts per chromosome, proportional to length."""
    total = sum(l for _, l in chroms)
    plan = []
    for name, length in chroms:
        plan.append((name, length, int(round(n_dels * length / total)), int(round(n_dups * length / total))))
    return plan


def write_bed(path, rows):
    """rows: iterable of (chrom, start, end[, value])."""
    with open(path, "w") as f:
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")
