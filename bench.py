#!/usr/bin/env python3
"""bench.py -- CNV intervals genotyped per second on the read-depth / likelihood hot path.

A "step" is one SAMPLE through the whole hot path, on the path the metric is defined on ("the timed kernel
path starts from decoded tuples in pinned memory"; the producer seam is count_reads_bam, bam_data.c:192-221):

    the sample's decoded (pos, mapq) tuples in pinned host memory -- the positions as a producer that subtracts leaves them:
    differences of 10 bits at 1x, plus a short exception list (conga_sample_reads_packed; `hand_over_int32`: 32-bit positions)
      -> HBM over PCIe, and back into the int32 array the kernels read (a segmented scan)
      -> GC-stratified depth sums -> expected_read_depth[101] -> per-interval observed depth, serial-float expected
         chain, 3-state log-likelihoods, c-score, CN (conga_chrom_compute: two launches)
      -> the result records in host memory (conga_sample_fetch)

The annotation, the call set and the tracks are the same for every sample of a cohort, so they are handed over once
(the library's cohort mode) and every step only brings new tuples; successive steps rotate over three different
samples through ONE context whose hand-over is double-buffered, so that the copy of sample k + 1 runs beside the kernels
and the fetch of sample k (the records of every step are fetched; nothing is cached between steps).  `value` is the
steady-state rate of that loop.
Also on the line:
  hand_over_int32 the same loop with 32-bit positions (rounds 2-3's hand-over); three_contexts: rounds 2-3's rotation over contexts
  single_sample   the same step unpipelined (copy, kernels, fetch one after the other), its latency
  kernel_only     the kernels alone on tuples already in HBM (round 1's `value`), rotating over the three resident
                  samples so that every launch streams from HBM, not from the 256 MiB Infinity Cache
  roofline        the HBM-bound kernel of the step (ingest_tuples_kernel) timed with HIP events on its own stream in that
                  rotation; step_bound = the PCIe copy that bounds the step itself
  roofline_dense  the reference's dense formulation (read_depth[] materialised), depth_tile_kernel
  end_to_end      `conga --cohort` over whole-genome 1x BAMs written on the spot: first sample, every further sample, with
                  the decode on the GPU and with the host decoders; zlib + the oracle on one core beside it
  configs         short legs for BASELINE configs[2] (dels + dups + mappability) and configs[4] (5x, --rp split reads)
  cpu_baseline    the oracle (serial port of the reference's loops), 1 thread, same workload; its records are compared
                  with the HIP path's at full size in the same run (cn_concordance is computed from that comparison)

Multi-GPU (`--gpus N` under torch.distributed.run, one process per GPU): chromosomes are sharded LPT across ranks, no
data-path collective, one RCCL gather of the fixed-size result records per step.  The default at N > 1 is STRONG
scaling -- one whole-genome sample per step sharded over the N GPUs, which is what BASELINE configs[3] names; the
line also carries a `weak` leg (N samples per step, one per rank's worth of chromosomes).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from conga_amd import capi, shard, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md, Chip-level parameters)
PCIE_PEAK_GBS = 64.0    # PCIe Gen5 x16, one direction, raw
N_ROTATE = 3            # samples (and contexts) the timed loop rotates over: 3 x 157 MB of tuples > 256 MiB of Infinity Cache


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="auto = strong at N > 1 (configs[3]: one sample sharded over the GPUs); the other one runs as a leg")
    ap.add_argument("--config", choices=("dels", "dels+dups+map"), default="dels",
                    help="dels = BASELINE configs[1]; dels+dups+map = configs[2]")
    ap.add_argument("--cov", type=float, default=1.0)
    ap.add_argument("--formulation", choices=("auto", "dense"), default="auto",
                    help="auto = tuple space whenever identical (library default); dense = CONGA_FLAG_MATERIALIZE_DEPTH")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="lower bound of CPU-baseline work (oracle, 1 thread); 0 disables the leg")
    ap.add_argument("--chroms", type=str, default="", help="comma list of chromosome names (debug)")
    ap.add_argument("--no-dense-leg", dest="dense_leg", action="store_false",
                    help="skip the dense-formulation leg that follows the timed region at N=1")
    ap.add_argument("--no-config-legs", dest="config_legs", action="store_false",
                    help="skip the configs[2] / configs[4] legs that follow the timed region at N=1")
    ap.add_argument("--no-e2e-leg", dest="e2e_leg", action="store_false",
                    help="skip the end-to-end leg (`conga --cohort` over whole-genome BAMs written into /tmp) that follows the timed region at N=1")
    ap.add_argument("--rp-chroms", type=str, default="auto",
                    help="chromosomes of the configs[4] (--rp) leg; 'all' = the whole genome (131.5 M records: ~60 GB of host memory, "
                         "~35 GB under /tmp for a minute, ~2 minutes); 'auto' = all when the host has that to spare, else 20,21,22")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="run the multi-rank code path (RCCL process group, device-resident records, gather) with the "
                         "ranks there are, even one -- a one-GPU check of the path the driver runs at N > 1")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------------------------------------
def build_units(args, world, scaling, config):
    """(sample, chromosome) units of ONE step and their owner rank."""
    chroms = synth.GRCH37_AUTOSOMES
    if args.chroms:
        keep = set(args.chroms.split(","))
        chroms = tuple(c for c in chroms if c[0] in keep)
    n_dups = synth.N_DUPS_GENOME if config != "dels" else 0
    plan = synth.genome_plan(chroms, synth.N_DELS_GENOME, n_dups)
    n_samples = world if scaling == "weak" else 1
    units = []
    for sample in range(n_samples):
        for name, length, nd, nu in plan:
            units.append(dict(sample=sample, name=name, length=length, n_dels=nd, n_dups=nu))
    costs = [shard.unit_cost(u["length"], u["n_dels"] + u["n_dups"]) for u in units]
    owner = shard.lpt_partition(costs, world)
    for u, o in zip(units, owner):
        u["owner"] = o
    return units


def make_unit(u, args, config):
    """One chromosome's layout (GC track, intervals, track rows) and the reads of N_ROTATE different samples on it."""
    c = synth.make_chrom(u["name"], u["length"], cov=args.cov, n_dels=u["n_dels"], n_dups=u["n_dups"],
                         seed=synth.BASE_SEED + 1000 * u["sample"], mappability=(config != "dels"))
    ds, de = synth.kept_sorted(c.del_start, c.del_end)
    us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
    reads = [(c.pos, c.mapq)]
    for j in range(1, N_ROTATE):   # other individuals: same chromosome, other reads
        rng = np.random.default_rng([synth.BASE_SEED + 1000 * u["sample"] + j, int(u["name"]), 99])
        reads.append(synth.make_reads(c.length, c.gc, c.step, args.cov, 100, rng))
    u.update(chrom=c, ds=ds, de=de, us=us, ue=ue, n_iv=len(ds) + len(us), reads=reads, n_reads=len(c.pos),
             sum_len=int((de.astype(np.int64) - ds).sum() + (ue.astype(np.int64) - us).sum()))
    return u


def cpus_granted():
    """Cores this process may use: the affinity mask and the cgroup's CPU quota (a GPU box grants 16 of its host's 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_throttled():
    """(times, milliseconds) the cgroup's CPU quota has held this process back so far -- a producer with as many busy threads as
    the quota has CPUs runs into it."""
    try:
        st = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0)) / 1e3
    except (OSError, ValueError):
        return 0, 0.0


def open_layout(ctx, mine):
    """The part of a cohort job that is handed over once: chromosomes, GC arrays, intervals, tracks."""
    for i, u in enumerate(mine):
        c = u["chrom"]
        u["index"] = ctx.chrom_begin(c.length, c.gc)
        assert u["index"] == i
        if c.map_start is not None:
            ctx.mappability(c.map_start, c.map_end, c.map_val)
        ctx.intervals("D", u["ds"], u["de"])
        if len(u["us"]):
            ctx.intervals("E", u["us"], u["ue"])


def pinned_samples(ctx, mine, packer):
    """[(pos, mapq, chrom_off, packed, n_esc, width, n_packed_bytes)] per rotation slot: this rank's chromosomes' tuples one behind
    the other in pinned host memory -- where a decoder would have left them -- and the same positions as the library's own producer
    (conga_packer_*, conga_amd/csrc/pack_host.h) packs them: W-bit differences, the exceptions behind them in the same buffer."""
    out = []
    for j in range(N_ROTATE):
        n = sum(len(u["reads"][j][0]) for u in mine)
        pos, mapq = ctx.host_alloc(max(n, 1), np.int32), ctx.host_alloc(max(n, 1), np.uint8)
        off = np.zeros(len(mine) + 1, np.uint64)
        at = 0
        for k, u in enumerate(mine):
            p, m = u["reads"][j]
            pos[at:at + len(p)] = p
            mapq[at:at + len(p)] = m
            at += len(p)
            off[k + 1] = at
        width = int(os.environ["CONGA_BENCH_WIDTH"]) if os.environ.get("CONGA_BENCH_WIDTH") else 0   # (measurement switch; 0: the producer's rule)
        d_pin = ctx.host_alloc(packer.bound(at, max(at // 16, 4096)), np.uint8)
        packer.start(pos[:at], off, d_pin, width)
        width, n_esc, nbytes = packer.finish()
        out.append((pos[:at], mapq[:at], off, d_pin, n_esc, width, nbytes))
    return out


def depth_kernel_bytes(units):
    """Algorithmic bytes of the depth_tile launch over these chromosomes: read_depth
    written once as int16, the tuples read once (int32 pos + uint8 mapq), one GC byte per window, one
    tile-index word per tile."""
    total = 0
    for u in units:
        L, n = u["length"], u["n_reads"]
        total += 2 * L + 5 * n + (L + 99) // 100 + 4 * ((L + 2047) // 2048 + 1)
    return total


def tuple_kernel_bytes(units, j=0):
    """Algorithmic bytes of one ingest_tuples launch over these chromosomes: every position read
    once (int32; the MAPQ bytes are not read with the default threshold, under which every read counts) and each GC byte
    at most once."""
    return sum(4 * len(u["reads"][j][0]) + (u["length"] + 99) // 100 for u in units)


def dense_reference_bytes(u, with_map):
    """bytes the reference's dense formulation touches for this chromosome."""
    L, n, sl, niv = u["length"], u["n_reads"], u["sum_len"], u["n_iv"]
    b = 2 * L + 9 * n + (2 * L + L // 100) + (2 * sl + sl // 100 + 64 * niv)
    if with_map:
        b += 4 * L + 4 * sl
    return b


def compare_records(got, want, what, with_map):
    """The parity bar of BASELINE.json on one chromosome's records; -> number of intervals whose CN call agrees."""
    assert np.array_equal(got["observed"], want["observed"]), "observed mismatch " + what
    assert np.array_equal(got["expected"].view(np.uint32), want["expected"].view(np.uint32)), "expected_rd bits " + what
    for k in ("lhomo", "lhetero", "lnone"):
        assert np.allclose(got[k], want[k], rtol=0, atol=1e-6, equal_nan=True), k + " " + what
    assert np.allclose(got["score"], want["score"], rtol=1e-6, atol=1e-6, equal_nan=True), "score " + what
    if with_map:
        assert np.allclose(got["mappability"], want["mappability"], rtol=0, atol=1e-6, equal_nan=True), "mappability " + what
    same = int(np.count_nonzero(got["cn"] == want["cn"]))
    assert same == len(want), "CN mismatch " + what
    return same


def cpu_baseline(mine, recs, E, args, with_map):
    """The oracle (a serial port of the reference's loops) timed on this host, 1 thread, on a bounded sample of the
    same workload (rotation slot 0); its results double as a full-size parity check of the HIP path's records."""
    from oracle import oracle as O
    O.lib()
    first = np.cumsum([0] + [u["n_iv"] for u in mine])
    todo = sorted(range(len(mine)), key=lambda i: mine[i]["length"])
    t_cpu, n_iv, same, names = 0.0, 0, 0, []
    for i in todo:
        u = mine[i]
        c = u["chrom"]
        pos, mapq = u["reads"][0]
        t0 = time.perf_counter()
        rd, _ = O.count_reads(c.length, pos, mapq, -1)
        Eo, _, _ = O.calc_mean_per_chr(rd, c.gc)
        m = O.paint_mappability(c.length, c.map_start, c.map_end, c.map_val) if c.map_start is not None else None
        od = O.find_depths(rd, m, c.gc, Eo, "D", O.make_svs(u["ds"], u["de"]))
        ou = O.find_depths(rd, m, c.gc, Eo, "E", O.make_svs(u["us"], u["ue"]))
        t_cpu += time.perf_counter() - t0
        n_iv += u["n_iv"]
        names.append(u["name"])
        # parity at full size (the oracle is the checker here, never the thing measured above)
        assert np.array_equal(E[i].view(np.uint32), Eo.view(np.uint32)), "expected_read_depth mismatch chr" + u["name"]
        a, nd = int(first[i]), len(u["ds"])
        same += compare_records(recs[a:a + nd], od, "chr" + u["name"] + " dels", with_map)
        same += compare_records(recs[a + nd:a + u["n_iv"]], ou, "chr" + u["name"] + " dups", with_map)
        if t_cpu >= args.cpu_seconds:
            break
    return dict(value=n_iv / t_cpu, unit="intervals/s", cores=1, kind="port",
                sample="chromosomes %s of the same workload (%d intervals, %.1f s, oracle/conga_oracle.c through ctypes, "
                       "every record compared with the HIP path's)" % (",".join(names), n_iv, t_cpu)), same / max(n_iv, 1)


# ---------------------------------------------------------------------------------------------------------------------
# one measured leg: layout once, then K pipelined steps
# ---------------------------------------------------------------------------------------------------------------------
class Leg:
    """Three contexts behind one layout on this rank's GPU, three pinned samples, and the step loop."""

    def __init__(self, args, env, scaling, config, flags_extra=0):
        self.args, self.env, self.scaling, self.config = args, env, scaling, config
        world, rank = env["world"], env["rank"]
        self.units = build_units(args, world, scaling, config)
        self.mine = [make_unit(u, args, config) for u in self.units if u["owner"] == rank]
        self.with_map = config != "dels"
        flags = capi.FLAG_BATCH | flags_extra | (capi.FLAG_MATERIALIZE_DEPTH if args.formulation == "dense" else 0)
        if env["dist_on"]:
            flags |= capi.FLAG_RESULTS_ON_DEVICE  # the records travel device-to-device into the RCCL gather, not over PCIe
        self.ctxs = [capi.Context(device=env["local_rank"], flags=flags) for _ in range(N_ROTATE)]
        for c in self.ctxs:
            open_layout(c, self.mine)
        # the producer's threads: in this loop nothing else works on the host, so all of the cores the process may use but two (the
        # thread that drives the step and the runtime's own) -- 14 on a box that grants 16: 1.5 ms per 1x genome against 1.8 with 8
        # (N ranks on one host share its cores: each rank's share, and never more than 14 -- beyond 8 the host's memory is the limit)
        share = max(1, min(14, cpus_granted() // max(env["world"], 1) - 2))
        self.packer = capi.Packer(int(os.environ.get("CONGA_BENCH_PACK_THREADS", str(share))))
        self.packer_threads = self.packer.threads()
        self.samples = pinned_samples(self.ctxs[0], self.mine, self.packer)
        # where the TIMED encode writes (hand_over = "packed+encode"): three pinned buffers, since the bytes of sample k must stay as they
        # are until the fetch behind compute k has returned (include/conga_hip.h) while sample k + 1 is being encoded
        self.enc = [self.ctxs[0].host_alloc(max(len(x[3]) for x in self.samples), np.uint8) for _ in range(3)]
        self.enc_info = [None] * 3
        self.n_iv_mine = sum(u["n_iv"] for u in self.mine)
        self.rec = capi.RESULT_DTYPE.itemsize
        self.out = [np.zeros(self.n_iv_mine, dtype=capi.RESULT_DTYPE) for _ in range(N_ROTATE)]
        self.E = [np.zeros((len(self.mine), 101), np.float32) for _ in range(N_ROTATE)]
        self.total_iv = self.n_iv_mine
        # handing the layout over is not a step: every context prepares its device layout here, once
        self.hand_over = "packed"      # "packed": differences encoded beforehand; "packed+encode": encoded inside the step by the
        #                                library's producer on host threads, beside the step before; "int32": conga_sample_reads
        self.rotate_contexts = False   # N = 1: True = rounds 2-3's loop over three contexts (kept as a figure of its own)
        for j, c in enumerate(self.ctxs):
            self.reads(c, j)
            c.compute()
            c.sync()
        if env["dist_on"]:
            self._dist_setup()
            if os.environ.get("CONGA_BENCH_PHASES"):
                self._finish_phases = [0.0] * 6

    def _dist_setup(self):
        import torch
        import torch.distributed as dist
        env = self.env
        t = torch.tensor([self.n_iv_mine * self.rec], dtype=torch.int64, device=env["gather_dev"])
        all_b = [torch.zeros_like(t) for _ in range(env["world"])]
        dist.all_gather(all_b, t)
        self.bytes_per_rank = [int(x.item()) for x in all_b]
        self.total_iv = sum(self.bytes_per_rank) // self.rec
        pad = max(max(self.bytes_per_rank), 1)
        dev = env["dev"]
        # one padded send buffer per context; rank 0 receives into per-context buffers and brings them to the host
        self.packed = [torch.zeros(pad, dtype=torch.uint8, device=dev) for _ in range(N_ROTATE)]
        # rank 0 receives into the rows of ONE tensor per context (one copy to the host for all ranks) ...
        self.recv_all = [torch.empty((env["world"], pad), dtype=torch.uint8, device=env["gather_dev"]) if env["rank"] == 0 else None
                         for _ in range(N_ROTATE)]
        self.recv = [[self.recv_all[j][r] for r in range(env["world"])] if env["rank"] == 0 else None for j in range(N_ROTATE)]
        self.host_recv = [torch.empty((env["world"], pad), dtype=torch.uint8).pin_memory() if env["rank"] == 0 else None
                          for _ in range(N_ROTATE)]
        # ... and waits for a step's gather + copy one step later (drain() at the end of a run), so that their latency lies
        # beside the next step instead of in front of it
        self.in_flight = [None] * N_ROTATE
        self.ext = [None if env["rehearsal"] else torch.cuda.ExternalStream(c.stream(), device=dev) for c in self.ctxs]
        self.gstream = None if env["rehearsal"] else torch.cuda.Stream(device=dev)

    def reads(self, c, j):
        """Sample j's tuples from pinned host memory into context c: as W-bit differences (conga_sample_reads_packed, one copy) or as
        32-bit positions (conga_sample_reads), as self.hand_over says."""
        pos, mapq, off, d_pin, n_esc, width, _nb = self.samples[j]
        if self.hand_over == "int32":
            c.sample_reads(pos, mapq, off)
        else:
            c.sample_reads_packed(d_pin, width, n_esc, None, mapq, off)

    # -- the producer inside the step: sample k's positions are encoded by the library's packer (host threads) into pinned buffer
    # k % 3 while step k - 1 is copied, computed and fetched
    def encode_start(self, k):
        pos, _mapq, off = self.samples[k % N_ROTATE][:3]
        self.packer.start(pos, off, self.enc[k % 3], self.samples[k % N_ROTATE][5])

    def encode_finish_and_hand_over(self, c, k):
        width, n_esc, _nb = self.packer.finish()
        _pos, mapq, off = self.samples[k % N_ROTATE][:3]
        c.sample_reads_packed(self.enc[k % 3], width, n_esc, None, mapq, off)

    # -- one step, in two halves so that the copy of step k + 1 is in flight while step k finishes
    def enqueue(self, k):
        c = self.ctxs[k % N_ROTATE]
        if self.hand_over == "packed+encode":         # (no step before this one to encode beside: the producer runs in front of the copy)
            self.encode_start(k)
            self.encode_finish_and_hand_over(c, k)
        else:
            self.reads(c, k % N_ROTATE)               # pinned host -> HBM, asynchronous
        c.compute()                                   # the whole hot path for this rank's chromosomes, asynchronous

    def finish(self, k, c=None, previous=False):
        """The records of step k: into host memory (N = 1), or into the RCCL gather towards rank 0 (N > 1) -- c: the context that
        computed the step (the one-context loop passes its only one); previous: step k + 1 has been computed ahead
        (conga_chrom_compute_ahead) and step k's results are those of the compute before the latest one."""
        j = k % N_ROTATE
        if c is None:
            c = self.ctxs[j]
        if not self.env["dist_on"]:
            # waits for the kernels of step k; records into host memory
            (c.sample_fetch_previous if previous else c.sample_fetch)(self.out[j], self.E[j])
            return
        import torch
        import torch.distributed as dist
        env = self.env
        ph = getattr(self, "_finish_phases", None)    # (CONGA_BENCH_PHASES=1: where the multi-rank finish's host time goes)
        t0 = time.perf_counter()
        if previous:
            c.sync_previous()                         # (settles the wrap guard before the records are read on the device)
            c.results_copy_previous(self.packed[j].data_ptr(), self.n_iv_mine * self.rec)
        else:
            c.sync()
            c.results_copy(self.packed[j].data_ptr(), self.n_iv_mine * self.rec)
        if env["rehearsal"]:                          # one GPU, gloo: host tensors
            c.sync()
            dist.gather(self.packed[j].cpu(), self.recv[j], dst=0)
            return
        t1 = time.perf_counter()
        # the gathered records of the step before the last one are in (pinned) host memory before this step's go into the send
        # buffer that step's successor will use (three buffers: steps k - 1 and k may be in flight).  Waiting for step k - 1 here --
        # round 4's loop -- is waiting for the compute in front of its copy: 0.09 ms of every 0.17 ms step with little to copy.
        prev = self.in_flight[(k - 2) % N_ROTATE] if k > 1 else None
        if prev is not None:
            prev.synchronize()
            self.in_flight[(k - 2) % N_ROTATE] = None
        t2 = time.perf_counter()
        # RCCL over xGMI on a stream of the gather's own, ordered behind the copy on the context's stream by an event: the context's
        # stream goes on with the next compute while the records travel (send buffer j is not written again before step k + 3)
        ext = self.ext[self.ctxs.index(c)]
        copied = torch.cuda.Event()
        copied.record(ext)
        self.gstream.wait_event(copied)
        t3 = time.perf_counter()
        with torch.cuda.stream(self.gstream):
            dist.gather(self.packed[j], self.recv[j], dst=0)
            t4 = time.perf_counter()
            if env["rank"] == 0:
                self.host_recv[j].copy_(self.recv_all[j], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.gstream)
            self.in_flight[j] = ev
        if ph is not None:
            t5 = time.perf_counter()
            for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
                ph[i] += b - a
            ph[5] += 1

    def drain(self):
        ph = getattr(self, "_finish_phases", None)
        if ph is not None and ph[5] > 20:
            print("[phases] finish over %d steps (ms each): sync + copy of the records %.3f, wait for the gather before %.3f, events %.3f, dist.gather %.3f, "
                  "copy to the host + event %.3f" % ((int(ph[5]),) + tuple(1e3 * x / ph[5] for x in ph[:5])), file=sys.stderr, flush=True)
            self._finish_phases = [0.0] * 6
        if self.env["dist_on"] and not self.env["rehearsal"]:
            for j, ev in enumerate(self.in_flight):
                if ev is not None:
                    ev.synchronize()
                    self.in_flight[j] = None

    def run(self, n, pipelined=True):
        if pipelined and not self.rotate_contexts:
            # ONE context, the C-ABI's own pipelining, at N = 1 and at N > 1 alike: sample k + 1 is handed over (its copy runs on the
            # context's second stream into the other pair of tuple buffers) while sample k is computed and its records are fetched
            # (N = 1) or gathered (N > 1: finish(k) copies them into send buffer k % 3 on the context's stream and gathers on a stream
            # of the gather's own).  With hand_over = "packed+encode" the library's producer encodes sample k + 1 meanwhile.
            c = self.ctxs[0]
            # Two computes in flight (ABI v9): step k + 1 is handed over AND enqueued before step k is waited for, so that the GPU
            # goes from one step's last launch to the next one's first without waiting for the host's wake-up, its fetch and its
            # launches (CONGA_BENCH_NO_AHEAD=1: round 4's order, hand over (k + 1) -> fetch (k) -> compute (k + 1)).
            ahead = not os.environ.get("CONGA_BENCH_NO_AHEAD")
            if self.hand_over == "packed+encode":
                self.encode_start(0)
                self.encode_finish_and_hand_over(c, 0)
                if n > 1:
                    self.encode_start(1)
                c.compute()
                phases = os.environ.get("CONGA_BENCH_PHASES")   # (measurement switch: where the host thread's time goes, on stderr)
                t_ph = [0.0] * 5
                thr0 = cpu_throttled()
                for k in range(1, n):
                    t0 = time.perf_counter()
                    width, n_esc, _nb = self.packer.finish()
                    t1 = time.perf_counter()
                    _pos, mapq, off = self.samples[k % N_ROTATE][:3]
                    c.sample_reads_packed(self.enc[k % 3], width, n_esc, None, mapq, off)
                    t2 = time.perf_counter()
                    if k + 1 < n:
                        self.encode_start(k + 1)
                    t3 = time.perf_counter()
                    if ahead:
                        c.compute_ahead()
                        t4 = time.perf_counter()
                        self.finish(k - 1, c, previous=True)
                        t5 = time.perf_counter()
                        t_ph[3] += t5 - t4
                        t_ph[4] += t4 - t3
                    else:
                        self.finish(k - 1, c)
                        t4 = time.perf_counter()
                        c.compute()
                        t5 = time.perf_counter()
                        t_ph[3] += t4 - t3
                        t_ph[4] += t5 - t4
                    for i, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3))):
                        t_ph[i] += b - a
                if phases and n > 1:
                    print("[phases] per step over %d steps: wait for the encode %.3f ms, hand over %.3f, start the next encode %.3f, fetch %.3f, "
                          "compute (enqueue) %.3f; the cgroup held the process back %d times, %.1f ms in all"
                          % ((n - 1,) + tuple(1e3 * x / (n - 1) for x in t_ph) + tuple(b - a for a, b in zip(thr0, cpu_throttled()))),
                          file=sys.stderr, flush=True)
                self.finish(n - 1, c)
            else:
                self.reads(c, 0)
                c.compute()
                for k in range(1, n):
                    self.reads(c, k % N_ROTATE)
                    if ahead:
                        c.compute_ahead()
                        self.finish(k - 1, c, previous=True)
                    else:
                        self.finish(k - 1, c)
                        c.compute()
                self.finish(n - 1, c)
            self.drain()
            return
        if pipelined:
            self.enqueue(0)
            for k in range(1, n):
                self.enqueue(k)
                self.finish(k - 1)
            self.finish(n - 1)
        else:
            for k in range(n):
                self.enqueue(k)
                self.finish(k)
                self.drain()
        self.drain()

    def timed(self, steps, warmup, pipelined=True):
        """-> seconds for exactly `steps` steps, max over ranks, bracketed by barrier + synchronize."""
        import torch
        env = self.env
        if warmup:
            self.run(warmup, pipelined)
        env["barrier"]()
        t0 = time.perf_counter()
        self.run(steps, pipelined)
        env["barrier"]()
        elapsed = time.perf_counter() - t0
        if env["dist_on"]:
            import torch.distributed as dist
            t = torch.tensor([elapsed], dtype=torch.float64, device=env["gather_dev"])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def kernel_only(self, steps, rotate=True):
        """The kernels alone on resident tuples (what round 1 reported as `value`): compute + sync per step."""
        for c in self.ctxs:
            c.compute()
            c.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            c = self.ctxs[k % N_ROTATE if rotate else 0]
            c.compute()
            c.sync()
        return (time.perf_counter() - t0) / steps

    def kernel_only_pipelined(self, steps):
        """The same with the NEXT sample's compute in the queues before the last one's is waited for (three contexts, each with its
        resident sample, on streams of their own): the kernels' throughput rather than a step's latency."""
        for c in self.ctxs:
            c.compute()
            c.sync()
        t0 = time.perf_counter()
        self.ctxs[0].compute()
        for k in range(1, steps):
            self.ctxs[k % N_ROTATE].compute()
            self.ctxs[(k - 1) % N_ROTATE].sync()
        self.ctxs[(steps - 1) % N_ROTATE].sync()
        return (time.perf_counter() - t0) / steps

    def profile_kernels(self, reps=4, rotate=True):
        """Per-kernel device time from HIP events recorded on the context's own stream (CONGA_FLAG_PROFILE)."""
        ctxs = self.ctxs if rotate else self.ctxs[:1]
        for c in ctxs:
            c.set_profile(True)
        kms = np.zeros(len(capi.KERNEL_NAMES))
        n, dense_ran = 0, False
        for _ in range(reps):
            for j, c in enumerate(ctxs):
                if self.hand_over != "int32":         # (so that the expansion of the differences is among the timed launches)
                    self.reads(c, j)
                c.compute()
                c.select(0)
                st = c.fetch()[3]
                kms += np.array(st.kernel_ms[:len(capi.KERNEL_NAMES)])
                dense_ran = bool(st.depth_materialized)
                n += 1
        for c in ctxs:
            c.set_profile(False)
        return kms / n, dense_ran

    def close(self):
        # torch's caching allocators record an event on every stream a block was used on when the block is freed: the
        # tensors that travelled on the contexts' streams have to go before those streams do
        if self.env["dist_on"]:
            import gc
            import torch
            torch.cuda.synchronize()
            self.packed = self.recv = self.recv_all = self.host_recv = self.ext = None
            self.in_flight = [None] * N_ROTATE
            gc.collect()
            torch.cuda.empty_cache()
        for c in self.ctxs:
            c.close()
        self.packer.close()


def traffic_of(name):
    tpath = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(tpath):
        return None, None
    t = json.load(open(tpath))
    return t.get("hbm_bytes_per_launch"), t.get("campaign"), t.get("algorithmic_bytes_per_launch")


def roofline_of(kernel, ms, nbytes, kms, traffic_file, regime):
    achieved = nbytes / (max(ms, 1e-9) * 1e-3) / 1e9
    traffic, campaign, alg_then = traffic_of(traffic_file)
    scaled = ""
    if traffic and alg_then and abs(nbytes / alg_then - 1.0) > 0.005:
        # the counters were taken on the N=1 workload of configs[1]: a launch over another share of it (a rank's chromosomes at
        # N > 1, another configuration or coverage) moves the same bytes per algorithmic byte
        traffic = round(traffic * nbytes / alg_then)
        scaled = "; scaled to this launch's algorithmic bytes: the campaign's launch had %d" % alg_then
    return dict(bound="hbm", kernel=kernel, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), algorithmic_bytes_per_launch=int(nbytes),
                avg_launch_ms=round(float(ms), 5), traffic=traffic,
                traffic_source="profiles/%s (rocprofv3 --pmc campaign %s; not measured in this run%s)" % (traffic_file, campaign, scaled),
                regime=regime, kernel_ms_per_step={k: round(float(v), 4) for k, v in zip(capi.KERNEL_NAMES, kms)})


def workload_text(leg, args, config, hand_over="packed"):
    per_sample = leg.total_iv // max(leg.env["world"] if leg.scaling == "weak" else 1, 1)
    what = {"dels": "BASELINE configs[1]", "dels+dups+map": "BASELINE configs[2]"}[config]
    if leg.env["world"] > 1 and leg.scaling == "strong":
        what = "BASELINE configs[3] (the configs[1] sample sharded by chromosome over %d GPUs)" % leg.env["world"]
    return ("%s: GRCh37 autosomes 1-22, %d deletion rows%s (%d intervals kept >= 1000 bp per sample)%s, %.1fx synthetic "
            "samples, 100-bp GC windows; one step = one sample: int32 positions in pinned host memory -> %s -> records in host memory" % (
                what, synth.N_DELS_GENOME, "" if config == "dels" else " + %d duplication rows" % synth.N_DUPS_GENOME,
                per_sample, "" if config == "dels" else ", 100-mer-like mappability track", args.cov,
                {"int32": "conga_sample_reads", "packed+encode": "conga_packer (encode timed) -> conga_sample_reads_packed",
                 "packed": "conga_sample_reads_packed (differences encoded beforehand)"}[hand_over]))


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    # stdout carries exactly one JSON line: anything a library prints there while the job runs (RCCL prints a version
    # banner on stdout when it creates a communicator) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # Rehearsal hook for a one-GPU box: CONGA_BENCH_REHEARSAL=1 puts every rank on cuda:0 and gathers over gloo
    # (host tensors).  Never set by the driver; numbers from it are not benchmark results.
    rehearsal = os.environ.get("CONGA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather_dev = torch.device("cpu") if rehearsal else dev
    dist_on = world > 1 or args.dist_selftest
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    env = dict(rank=rank, world=world, local_rank=local_rank, dev=dev, gather_dev=gather_dev, dist_on=dist_on,
               rehearsal=rehearsal, barrier=barrier)
    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")

    leg = Leg(args, env, scaling, args.config)
    # The same pipelined step with three hand-overs of the sample's positions, each timed over exactly --steps steps.  `value` is
    # the faster of the two whose producer is INSIDE the timed region: 32-bit positions as count_reads_bam leaves them
    # (conga_sample_reads), or W-bit differences encoded by the library's own producer on host threads beside the step before
    # (conga_packer_* -> conga_sample_reads_packed).  Differences encoded beforehand (round 3's `value`: what the link and the
    # engine do with a producer that subtracts while it decodes) stay on the line as `hand_over.packed_preencoded`.
    if os.environ.get("CONGA_BENCH_TRACE_THREE"):
        # (measurement switch, for a rocprofv3 timeline: only the pre-encoded packed step through ONE context and through THREE, a
        # second of sleep between the two so that the trace shows which is which)
        leg.hand_over = "packed"
        one = leg.timed(args.steps, args.warmup)
        time.sleep(1.0)
        leg.rotate_contexts = True
        three = leg.timed(args.steps, args.warmup)
        if rank == 0:
            os.write(json_fd, (json.dumps(dict(one_context_ms=round(1e3 * one / args.steps, 4), three_contexts_ms=round(1e3 * three / args.steps, 4))) + "\n").encode())
        leg.close()
        return
    timed = {}
    for name in ("packed+encode", "int32", "packed"):
        leg.hand_over = name
        timed[name] = leg.timed(args.steps, args.warmup)
    # (the reference's own form unless the other one is clearly faster: within 5 % the two are one measurement's noise apart)
    chosen = "packed+encode" if timed["packed+encode"] < 0.95 * timed["int32"] else "int32"
    if os.environ.get("CONGA_BENCH_HAND_OVER") in timed:   # (measurement switch)
        chosen = os.environ["CONGA_BENCH_HAND_OVER"]
    leg.hand_over = chosen
    elapsed = timed[chosen]
    ms_per_step = 1e3 * elapsed / args.steps
    samples_per_step = world if scaling == "weak" else 1

    out = None
    if rank == 0:
        if dist_on:
            got = sum(leg.bytes_per_rank) // leg.rec
            assert got == leg.total_iv
            if args.dist_selftest and not rehearsal:
                # the gathered bytes are the records the context holds (fetch order = results_copy order)
                j = (args.steps - 1) % N_ROTATE
                want = leg.ctxs[0].sample_fetch()[0].tobytes()
                have = leg.host_recv[j][0][:leg.bytes_per_rank[0]].numpy().tobytes()
                assert have == want, "gathered records differ from the fetched ones"
        mine = leg.mine
        reads_step = int(sum(len(u["reads"][0][0]) for u in mine))
        width, n_esc, packed_bytes = leg.samples[0][5], leg.samples[0][4], leg.samples[0][6]
        # what crosses the link per step: 4 bytes per read, or W-bit differences + 8 bytes per exception; the MAPQ bytes stay on the
        # host with the default threshold (never read: every read counts)
        h2d_bytes = 4 * reads_step if chosen == "int32" else packed_bytes
        cfg = dict(workload=workload_text(leg, args, args.config, chosen), samples_per_step=samples_per_step,
                   chromosomes_per_sample=len(leg.units) // samples_per_step, intervals_per_step=int(leg.total_iv),
                   reads_per_step_rank0=reads_step, rotation="%d samples, one context (the same loop at every N)" % N_ROTATE,
                   hand_over=chosen,
                   parallelism="chromosome-sharded x%d, one RCCL gather per step" % world)
        out = dict(metric="CNV intervals genotyped/sec (1000G Phase-3 set); CN-call concordance vs ref",
                   value=round(leg.total_iv * args.steps / elapsed, 1), unit="intervals/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
                   higher_is_better=True, scaling=scaling, vs_baseline=None,
                   dtype="i32+f32/f64",  # counts, serial float32 chain, double scores
                   data="synthetic", config=cfg)

        def line(name, nbytes, note):
            return dict(ms_per_step=round(1e3 * timed[name] / args.steps, 4), value=round(leg.total_iv * args.steps / timed[name], 1),
                        bytes_per_step=int(nbytes), note=note)
        out["hand_over"] = dict(
            chosen=chosen,
            rule="`value` = int32 (positions as count_reads_bam leaves them) unless packed+encode -- the only other hand-over whose producer "
                 "runs inside the timed region -- is more than 5 % faster",
            int32=line("int32", 4 * reads_step, "conga_sample_reads: 32-bit positions as count_reads_bam leaves them, 4 bytes per read over PCIe"),
            packed_encode_timed=line("packed+encode", packed_bytes,
                                     "conga_packer_start/finish (the library's producer, %d host threads, conga_amd/csrc/pack_host.h) encodes sample "
                                     "k + 1 from its int32 array into %d-bit differences while step k is copied, computed and fetched; "
                                     "then conga_sample_reads_packed" % (leg.packer_threads, width)),
            packed_preencoded=line("packed", packed_bytes,
                                   "conga_sample_reads_packed on differences encoded BEFORE the timed region (round 3's `value`): what the link "
                                   "and the engine do with a producer that subtracts while it decodes; %d exceptions" % n_esc))
        out["hand_over_int32"] = out["hand_over"]["int32"]   # (the name earlier rounds' lines carry)

    single_s = leg.timed(max(3, min(args.steps, 10)), 1, pipelined=False) / max(3, min(args.steps, 10))
    if rank == 0:
        out["single_sample"] = dict(ms_per_step=round(1e3 * single_s, 4), value=round(leg.total_iv / single_s, 1),
                                    note="copy, kernels and fetch of one sample one after the other (latency of a step)")
        h2d = dict(bound="pcie-h2d", bytes_per_step=h2d_bytes, achieved=round(h2d_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                   peak=PCIE_PEAK_GBS, unit="GB/s", note="%.2f bytes per read (hand-over %s: %s; with "
                   "--mq -1, the reference's default, the MAPQ bytes are never read and are not sent) over PCIe Gen5 x16 per step"
                   % (h2d_bytes / max(reads_step, 1), chosen, "32-bit positions" if chosen == "int32" else
                      "%d-bit differences, %d exceptions of 8 bytes" % (width, n_esc)))
        h2d["frac"] = round(h2d["achieved"] / PCIE_PEAK_GBS, 4)
        # what THIS box's link gives a copy of that size by itself (the boxes differ: 39-57 GB/s): pinned host -> HBM, nothing beside it
        try:
            import torch
            src = torch.empty(h2d_bytes, dtype=torch.uint8).pin_memory()
            dst = torch.empty(h2d_bytes, dtype=torch.uint8, device=env["dev"])
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e30
            for _ in range(12):
                ev0.record()
                dst.copy_(src, non_blocking=True)
                ev1.record()
                ev1.synchronize()
                best = min(best, ev0.elapsed_time(ev1))
            h2d["link_probe"] = dict(gbs=round(h2d_bytes / (best * 1e-3) / 1e9, 2), ms=round(best, 4),
                                     note="one pinned copy of bytes_per_step by itself in this run (best of 12): what the step's copy can reach on this box")
            h2d["frac_of_link_probe"] = round(h2d["achieved"] / h2d["link_probe"]["gbs"], 4)
            del src, dst
        except Exception as e:   # (the probe is a figure beside the line, not part of it)
            h2d["link_probe"] = dict(error="%s: %s" % (type(e).__name__, e))
        out["step_bound"] = h2d

    if rank == 0 and not dist_on:
        leg.rotate_contexts = True
        e3 = leg.timed(args.steps, 2 * N_ROTATE)   # (every context's second pair of buffers is touched once before the clock starts)
        leg.rotate_contexts = False
        out["three_contexts"] = dict(ms_per_step=round(1e3 * e3 / args.steps, 4), value=round(leg.total_iv * args.steps / e3, 1),
                                     note="rounds 2-3's loop: the steps rotate over three contexts (the copy of sample k + 1 beside the "
                                          "kernels and the fetch of sample k by way of separate contexts), hand-over %s; `value` is ONE "
                                          "context: hand over (k + 1) -> conga_sample_fetch(k) -> conga_chrom_compute(k + 1)" % chosen)
    if rank == 0:
        ko_rot = leg.kernel_only(max(args.steps, 12), rotate=True)
        ko_one = leg.kernel_only(max(args.steps, 12), rotate=False)
        ko_pipe = leg.kernel_only_pipelined(max(args.steps, 12))
        out["kernel_only"] = dict(value=round(leg.total_iv / ko_rot, 1), ms_per_step=round(1e3 * ko_rot, 4),
                                  regime="tuples resident in HBM, %d samples rotated (%.0f MB between reuses > 256 MiB "
                                         "Infinity Cache)" % (N_ROTATE, N_ROTATE * tuple_kernel_bytes(mine) / 1e6),
                                  one_sample_replayed=dict(value=round(leg.total_iv / ko_one, 1), ms_per_step=round(1e3 * ko_one, 4),
                                                           regime="one resident sample replayed (fits the Infinity Cache)"),
                                  pipelined=dict(value=round(leg.total_iv / ko_pipe, 1), ms_per_step=round(1e3 * ko_pipe, 4),
                                                 regime="the same three resident samples, the next one's compute in the queues before the last "
                                                        "one's is waited for (a context and a stream each): the kernels' throughput, where "
                                                        "ms_per_step above is a step's latency (compute, then wait)"),
                                  note="round 1's `value`: no PCIe copy in the step")
        kms, dense_ran = leg.profile_kernels(rotate=True)
        regime = "HBM-streaming: %d resident samples rotated, %.0f MB between reuses" % (N_ROTATE, N_ROTATE * tuple_kernel_bytes(mine) / 1e6)
        if dense_ran:
            roofline = roofline_of("depth_tile_kernel", kms[1], depth_kernel_bytes(mine), kms, "depth_tile_traffic.json", regime)
        else:
            # the serial-float chain is latency-bound ("report its time separately"); the HBM-bound
            # kernel of the tuple-space path is the one pass over the tuples
            roofline = roofline_of("ingest_tuples_kernel", kms[0], tuple_kernel_bytes(mine), kms, "ingest_tuples_traffic.json", regime)
            roofline["chain_ms"] = round(float(kms[6]), 4)
            k1, _ = leg.profile_kernels(rotate=False)
            roofline["cache_resident"] = dict(avg_launch_ms=round(float(k1[0]), 5),
                                              achieved=round(tuple_kernel_bytes(mine) / (max(k1[0], 1e-9) * 1e-3) / 1e9, 1),
                                              regime="one sample replayed: fits the 256 MiB Infinity Cache (round 1's figure)")
        if world > 1:
            roofline["note"] = "rank 0's share of the chromosomes"
        out["roofline"] = roofline
        out["formulation"] = "dense" if dense_ran else "tuple-space"
        out["dtype"] = "i16/i32+f32/f64" if dense_ran else "i32+f32/f64"
        dense = sum(dense_reference_bytes(u, args.config != "dels") for u in mine)
        out["dense_equivalent"] = dict(bytes=int(dense), achieved=round(dense / ko_rot / 1e9, 1), unit="GB/s",
                                       x_hbm_peak=round(dense / ko_rot / 1e9 / HBM_PEAK_GBS, 3),
                                       note="SURVEY.md 8d dense-formulation bytes / kernel_only step time; above 1.0 only "
                                            "because the tuple-space formulation never moves them")
    # ---- N > 1: the other scaling as a leg of its own
    if world > 1 and args.scaling == "auto":
        other = "weak" if scaling == "strong" else "strong"
        leg.close()
        leg2 = Leg(args, env, other, args.config)
        e2 = leg2.timed(args.steps, args.warmup)
        if rank == 0:
            out[other] = dict(scaling=other, samples_per_step=world if other == "weak" else 1, intervals_per_step=int(leg2.total_iv),
                              ms_per_step=round(1e3 * e2 / args.steps, 4), value=round(leg2.total_iv * args.steps / e2, 1))
        leg2.close()
        leg = None

    if rank == 0 and world == 1 and not dist_on:
        mine = leg.mine
        if not dense_ran and args.dense_leg:
            # the reference's dense formulation on the same inputs: a context with CONGA_FLAG_MATERIALIZE_DEPTH
            dctx = capi.Context(device=local_rank, flags=capi.FLAG_BATCH | capi.FLAG_MATERIALIZE_DEPTH)
            open_layout(dctx, mine)
            dctx.sample_reads(*leg.samples[0][:3])
            for _ in range(3):
                dctx.compute()
            dctx.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                dctx.compute()
                dctx.sync()
            d_ms = (time.perf_counter() - t1) / 5 * 1e3
            dctx.set_profile(True)
            dk = np.zeros(len(capi.KERNEL_NAMES))
            for _ in range(5):
                dctx.compute()
                dctx.select(0)
                dk += np.array(dctx.fetch()[3].kernel_ms[:len(capi.KERNEL_NAMES)])
            dk /= 5
            dctx.set_profile(False)
            out["roofline_dense"] = roofline_of("depth_tile_kernel", dk[1], depth_kernel_bytes(mine), dk, "depth_tile_traffic.json",
                                                "5.9 GB written per launch (beyond any cache); resident tuples")
            out["roofline_dense"]["ms_per_step"] = round(d_ms, 4)
            out["roofline_dense"]["value"] = round(leg.total_iv / (d_ms * 1e-3), 1)
            # both formulations must give the same records
            c0 = leg.ctxs[0]
            c0.sample_reads(*leg.samples[0][:3])
            c0.compute()
            g1 = c0.sample_fetch()[0]
            dctx.compute()
            g2 = dctx.sample_fetch()[0]
            assert g1.tobytes() == g2.tobytes(), "tuple-space and dense records differ"
            dctx.close()
        c0 = leg.ctxs[0]
        c0.sample_reads(*leg.samples[0][:3])
        c0.compute()
        recs, E, _ = c0.sample_fetch()
        if args.cpu_seconds > 0:
            out["cpu_baseline"], out["cn_concordance"] = cpu_baseline(mine, recs, E, args, leg.with_map)
            out["cpu_baseline"]["host_cores"] = os.cpu_count()
        leg.close()
        leg = None
        if args.e2e_leg and args.config == "dels" and args.formulation == "auto":
            # ---- end to end from BAM files ("reported separately"): `conga --cohort` over BAMs written on the spot
            from conga_amd import e2e_bench
            try:
                out["end_to_end"] = e2e_bench.leg(args, env, mine, recs, out.get("cpu_baseline", {}).get("value"))
            except (OSError, MemoryError) as e:   # (no room for 6 GB of scratch BAMs, say: the headline does not depend on this leg)
                out["end_to_end"] = dict(error="%s: %s" % (type(e).__name__, e))
        if args.config_legs and args.config == "dels":
            out["configs"] = config_legs(args, env)

    if leg is not None:
        leg.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


def config_legs(args, env):
    """Short driver-run legs for the other single-GPU configurations of BASELINE.json."""
    legs = {}
    # ---- configs[2]: dels + dups together with the 100-mer mappability track
    steps = max(10, min(2 * args.steps, 40))   # (a step is about a millisecond: ten of them behind two warm-up steps spread 0.78-1.16 ms from run to run)
    leg = Leg(args, env, "weak", "dels+dups+map")
    e = leg.timed(steps, 5)
    ko = leg.kernel_only(steps)
    kms, _ = leg.profile_kernels(reps=2)
    rows = int(sum(len(u["chrom"].map_start) for u in leg.mine))
    r = roofline_of("ingest_tuples_kernel", kms[0], tuple_kernel_bytes(leg.mine), kms, "ingest_tuples_traffic.json",
                    "HBM-streaming: %d resident samples rotated" % N_ROTATE)
    r["chain_ms"] = round(float(kms[6]), 4)
    legs["configs[2]"] = dict(workload=workload_text(leg, args, "dels+dups+map"), intervals_per_step=int(leg.total_iv),
                              track_rows=rows, ms_per_step=round(1e3 * e / steps, 4), value=round(leg.total_iv * steps / e, 1),
                              kernel_only=dict(ms_per_step=round(1e3 * ko, 4), value=round(leg.total_iv / ko, 1)), roofline=r)
    if args.cpu_seconds > 0:
        c0 = leg.ctxs[0]
        c0.sample_reads(*leg.samples[0][:3])
        c0.compute()
        recs, E, _ = c0.sample_fetch()
        small = argparse.Namespace(**vars(args))
        small.cpu_seconds = min(args.cpu_seconds, 4.0)
        legs["configs[2]"]["cpu_baseline"], legs["configs[2]"]["cn_concordance"] = cpu_baseline(leg.mine, recs, E, small, True)
    leg.close()
    # ---- configs[4]: 5x with the split-read path (--rp with --dups)
    from conga_amd import rp_bench
    if rp_bench is not None:
        if args.rp_chroms == "auto":
            # the configuration names the whole genome: take it when the host can hold it (the driver's boxes can), else three chromosomes
            import shutil
            try:
                import psutil
                ram = psutil.virtual_memory().available
            except Exception:
                ram = 0
            tmp_free = shutil.disk_usage(os.environ.get("CONGA_BENCH_TMP", "/tmp")).free
            auto = "host memory available %.0f GB (needs 120), scratch space %.0f GB (needs 45)" % (ram / 2**30, tmp_free / 2**30)
            args.rp_chroms = "all" if (ram >= 120 << 30 and tmp_free >= 45 << 30 and not args.chroms) else "20,21,22"
        else:
            auto = None
        legs["configs[4]"], smp = rp_bench.leg(args, env)
        if auto:
            legs["configs[4]"]["chromosomes_chosen_by"] = "--rp-chroms auto: " + auto
        if args.cpu_seconds > 0:
            from oracle import oracle as O
            t0 = time.perf_counter()
            O.split_read_rows(smp["ref"], smp["sat_s"], smp["sat_e"], smp["pos"], smp["mapq"], smp["flag"], smp["lq"], smp["off"],
                              smp["codes"], smp["qual"], -1, 60)
            t_cpu = time.perf_counter() - t0
            legs["configs[4]"]["cpu_baseline"] = dict(
                value=round(len(smp["pos"]) / t_cpu, 1), unit="records/s", cores=1, kind="port",
                sample="first %d records of chromosome %s through oracle/conga_oracle_sr.c (%.1f s, includes one build of the "
                       "chromosome's 10-mer index)" % (len(smp["pos"]), smp["name"], t_cpu))
    try:
        legs["bgzf_inflate"] = bgzf_leg(args, env)
    except (OSError, MemoryError) as e:   # (no room for the 45 MB scratch BAM, say: the headline does not depend on this leg)
        legs["bgzf_inflate"] = dict(error="%s: %s" % (type(e).__name__, e))
    return legs


def bgzf_leg(args, env):
    """The BGZF inflate stage of the BAM route (conga_reads_bgzf's first kernel; SURVEY.md 8f row 1: the producer side of
    count_reads_bam is htslib's BGZF reader in the reference, bam_data.c:253-259).  A 1x chromosome 21 is written as a BAM with
    random bases and qualities; its BGZF blocks, some twenty times over in the block table (the same compressed bytes, as many
    places of their own in the output: ~20 500 blocks, two and a half rounds of the 8 192 waves the machine holds), are inflated
    and CRC-checked by conga_inflate_blocks.  CPU beside it: zlib on one core, a sample of the same blocks."""
    import struct
    import tempfile
    import zlib
    from conga_amd import formats
    lens = dict(synth.GRCH37_AUTOSOMES)
    c = synth.make_chrom("21", lens["21"], cov=1.0)
    d = tempfile.mkdtemp(prefix="conga_bench_bgzf_")
    path = os.path.join(d, "r.bam")
    formats.write_bam_fast(path, "S", [(c.name, c.length, c.pos, c.mapq)], realistic=True, level=1)
    raw = np.fromfile(path, np.uint8)
    os.remove(path)
    os.rmdir(d)
    buf, blocks, at = raw.tobytes(), [], 0
    while at + 18 <= len(buf):
        bsize = struct.unpack_from("<H", buf, at + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", buf, at + bsize - 8)
        if isize:
            blocks.append((at + 18, bsize - 26, isize, crc))
        at += bsize
    times = max(1, round(2.5 * 8192 / max(len(blocks), 1)))   # ~20 500 blocks, as tools/inflate_bench.py's file has (the machine holds 8 192 waves)
    table = blocks * times
    inflated = sum(b[2] for b in table)
    with capi.Context(device=env["local_rank"]) as ctx:
        best = 1e30
        for _ in range(4):
            _o, status, ms = ctx.inflate_blocks(raw, table, want_out=False)
            assert not status.any(), "a BGZF block of the bench's own BAM did not inflate"
            best = min(best, ms)
    out = dict(workload="BGZF blocks of a 1x chromosome 21 (synthetic BAM, random bases and qualities, zlib level 1), %d blocks x %d "
                        "in the block table = %d blocks, %.1f MB compressed -> %.1f MB inflated per launch; one BGZF block per wave, "
                        "CRC32 checked on the device" % (len(blocks), times, len(table), len(raw) / 1e6, inflated / 1e6),
               blocks=len(table), kernel_ms=round(best, 3), value=round(inflated / best / 1e6, 2), unit="GB/s inflated",
               note="instruction-bound byte work (profiles/r03j_inflate_pmc.txt: vector unit busy ~83 %, scalar ~68 % of the "
                    "launch); as HBM bytes this is about 1 % of the peak.  The rate depends on how the blocks fill the machine's 8 192 "
                    "waves and on the data: more bytes per symbol (matches) inflate faster")
    # the same kernel on a BAM whose qualities come in runs of a few binned values (what a sequencer of the last decade writes),
    # deflated at zlib's level 6 like samtools does: more and longer matches per symbol than random qualities give
    try:
        from conga_amd import e2e_bench
        d2 = tempfile.mkdtemp(prefix="conga_bench_bgzf_")
        p2, _t = e2e_bench.write_bam(d2, "q", [(c.name, c.length, c.pos, c.mapq)], level=6)
        raw2 = np.fromfile(p2, np.uint8)
        import shutil
        shutil.rmtree(d2, ignore_errors=True)
        buf2, blocks2, at = raw2.tobytes(), [], 0
        while at + 18 <= len(buf2):
            bsize = struct.unpack_from("<H", buf2, at + 16)[0] + 1
            crc, isize = struct.unpack_from("<II", buf2, at + bsize - 8)
            if isize:
                blocks2.append((at + 18, bsize - 26, isize, crc))
            at += bsize
        times2 = max(1, round(2.5 * 8192 / max(len(blocks2), 1)))
        table2 = blocks2 * times2
        with capi.Context(device=env["local_rank"]) as ctx:
            best2 = 1e30
            for _ in range(4):
                _o, status, ms = ctx.inflate_blocks(raw2, table2, want_out=False)
                assert not status.any(), "a BGZF block of the bench's own BAM did not inflate"
                best2 = min(best2, ms)
        inflated2 = sum(b[2] for b in table2)
        out["bam_like"] = dict(workload="the same chromosome written by tools/bamwrite: pseudo-random bases, run-structured binned qualities, zlib "
                                        "level 6; %d blocks, %.1f MB -> %.1f MB per launch" % (len(table2), len(raw2) * times2 / 1e6, inflated2 / 1e6),
                               kernel_ms=round(best2, 3), value=round(inflated2 / best2 / 1e6, 2), unit="GB/s inflated")
    except (OSError, MemoryError) as e:
        out["bam_like"] = dict(error="%s: %s" % (type(e).__name__, e))
    if args.cpu_seconds > 0:
        k = min(len(blocks), 400)
        t0 = time.perf_counter()
        n = 0
        for off, ln, isz, _crc in blocks[:k]:
            n += len(zlib.decompress(buf[off:off + ln], -15))
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=round(n / t_cpu / 1e9, 3), unit="GB/s inflated", cores=1, kind="zlib",
                                   sample="%d of the blocks through zlib.decompress (%.2f s; no CRC check)" % (k, t_cpu))
    return out


if __name__ == "__main__":
    main()
