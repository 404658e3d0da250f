#!/usr/bin/env python3
"""Split-read path (--rp with --dups, BASELINE configs[4] shape on one chromosome): wall time of one compute with
every BAM record handed over, next to the CPU oracle on a bounded sample.  Not the bench line (bench.py is); this
path is built for parity, this tool says what it costs."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conga_amd import capi, synth  # noqa: E402

from conga_amd.rp_bench import CODE, stage_uniform  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", type=int, default=48_129_895)  # chr21
    ap.add_argument("--cov", type=float, default=5.0)
    ap.add_argument("--dels", type=int, default=2000)
    ap.add_argument("--dups", type=int, default=300)
    ap.add_argument("--cpu-reads", type=int, default=40_000)
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    L, l = a.length, 100
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
    ref[:9_400_000] = ord("N")  # the leading gap of chr21
    n = int(L * a.cov / l)
    pos = np.sort(rng.integers(9_400_000, L - 2 * l, n)).astype(np.int32)
    bases = ref[pos[:, None].astype(np.int64) + np.arange(l)]
    err = rng.random(bases.shape) < 0.003
    bases[err] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(err.sum()))
    # junction reads over a few duplications / deletions
    c = synth.make_chrom("21", L, cov=0.0, n_dels=a.dels, n_dups=a.dups, seed=5)
    ds, de = synth.kept_sorted(c.del_start, c.del_end)
    us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
    codes = CODE[bases]
    qual = np.full(bases.shape, 30, np.uint8)
    mapq = rng.choice(np.array([60, 60, 60, 0, 20], np.uint8), n)
    flag = np.zeros(n, np.uint16)
    gc = c.gc
    t0 = time.perf_counter()
    with capi.Context(device=0, mq_threshold=-1) as ctx:
        ctx.chrom_begin(L, gc)
        ctx.reads(pos, mapq)
        ctx.reference(bytes(ref))
        ctx.satellites(np.zeros(0, np.int32), np.zeros(0, np.int32))
        stage_uniform(ctx, pos, mapq, flag, codes, qual)
        ctx.intervals("D", ds, de)
        ctx.intervals("E", us, ue)
        ctx.sync()
        t_stage = time.perf_counter() - t0
        ctx.compute()
        ctx.sync()  # first compute includes the layout upload
        t1 = time.perf_counter()
        ctx.compute()
        ctx.sync()
        t_gpu = time.perf_counter() - t1
        dels, dups, _, st = ctx.fetch()
    out = dict(chrom_len=L, reads=n, dels=len(ds), dups=len(us), staging_s=round(t_stage, 2), gpu_compute_ms=round(t_gpu * 1e3, 2),
               reads_per_s=round(n / t_gpu), split_elements=int(st.split_elements), split_mappings=int(st.split_mappings),
               rows=int(st.split_del_rows + st.split_dup_rows))
    if a.cpu_reads > 0:
        from oracle import oracle as O
        k = min(a.cpu_reads, n)
        off = (np.arange(k, dtype=np.uint64) * l)
        t0 = time.perf_counter()
        O.split_read_rows(bytes(ref), np.zeros(0, np.int32), np.zeros(0, np.int32), pos[:k], mapq[:k], flag[:k],
                          np.full(k, l, np.int32), off, codes[:k].reshape(-1), qual[:k].reshape(-1), -1, 60)
        t_cpu = time.perf_counter() - t0
        out.update(cpu_sample_reads=k, cpu_sample_s=round(t_cpu, 2), cpu_reads_per_s=round(k / t_cpu),
                   note="CPU figure includes building the 10-mer index of the whole chromosome once")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
