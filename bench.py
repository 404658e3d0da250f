#!/usr/bin/env python3
"""bench.py -- CNV intervals genotyped per second on the read-depth / likelihood hot path.

Workload at N=1 (BASELINE.json configs[1]): the full 1000G-Phase-3-sized deletion set (~42k rows
before the min-size filter) over the 22 GRCh37 autosomes against a 1x synthetic sample.  A "step"
is one pass of the whole hot path over that sample: for every chromosome, read tuples (already
resident in HBM) -> GC-stratified depth sums -> expected_read_depth[101] -> per-interval observed
depth, serial-float expected chain, 3-state log-likelihoods, c-score and CN -> result records
gathered on rank 0.  Host-side BAM decoding and BED parsing are outside the timed region.

Formulation: the library default (`--formulation auto`) works in tuple space (SURVEY.md section 8d,
"sparse reformulation"): read_depth[i] is a count of read starts, so the GC sums are a histogram over
the kept reads and an interval's observed depth is the number of kept reads that start inside it --
identical results (asserted against the oracle on the whole genome below), 5 bytes per read of HBM
traffic instead of 4+ bytes per base.  `--formulation dense` forces the reference's formulation
(read_depth[] materialised in HBM); at N=1 a short dense leg runs after the timed region so that the
line carries both: `roofline` (dominant HBM-bound kernel of the timed path), `roofline_dense`
(depth_tile_kernel, the most HBM-intensive kernel of the library) and `dense_equivalent` (the dense
formulation's algorithmic bytes divided by the measured step time, as SURVEY.md 8d asks).

Multi-GPU: one process per GPU (torchrun), chromosomes sharded LPT across ranks, no data-path
collective, one RCCL gather of the fixed-size result records per step.  Default scaling is weak:
N ranks genotype N samples (22 N chromosome units); `--scaling strong` splits ONE sample.

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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md, Chip-level parameters)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--config", choices=("dels", "dels+dups+map"), default="dels",
                    help="dels = BASELINE configs[1]; dels+dups+map = configs[2]")
    ap.add_argument("--cov", type=float, default=1.0)
    ap.add_argument("--formulation", choices=("auto", "dense"), default="auto",
                    help="auto = tuple space whenever identical (library default); dense = CONGA_FLAG_MATERIALIZE_DEPTH")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="lower bound of CPU-baseline work (oracle, 1 thread); 0 disables the leg")
    ap.add_argument("--chroms", type=str, default="", help="comma list of chromosome names (debug)")
    ap.add_argument("--results-on-device", action="store_true",
                    help="CONGA_FLAG_RESULTS_ON_DEVICE at N=1 too (what every rank of a multi-GPU run does)")
    ap.add_argument("--no-dense-leg", dest="dense_leg", action="store_false",
                    help="skip the dense-formulation leg that follows the timed region at N=1")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="run the multi-rank code path (RCCL process group, device-resident records, gather) with the "
                         "ranks there are, even one -- a one-GPU check of the path the driver runs at N > 1")
    return ap.parse_args()


def build_units(args, world):
    """(sample, chromosome) units and their owner rank."""
    chroms = synth.GRCH37_AUTOSOMES
    if args.chroms:
        keep = set(args.chroms.split(","))
        chroms = tuple(c for c in chroms if c[0] in keep)
    n_dups = synth.N_DUPS_GENOME if args.config != "dels" else 0
    plan = synth.genome_plan(chroms, synth.N_DELS_GENOME, n_dups)
    n_samples = world if args.scaling == "weak" else 1
    units = []
    for sample in range(n_samples):
        for name, length, nd, nu in plan:
            units.append(dict(sample=sample, name=name, length=length, n_dels=nd, n_dups=nu))
    costs = [shard.unit_cost(u["length"], u["n_dels"] + u["n_dups"]) for u in units]
    owner = shard.lpt_partition(costs, world)
    for u, o in zip(units, owner):
        u["owner"] = o
    return units


def make_unit(u, args):
    """Generate one chromosome's inputs on the host."""
    c = synth.make_chrom(u["name"], u["length"], cov=args.cov, n_dels=u["n_dels"], n_dups=u["n_dups"],
                         seed=synth.BASE_SEED + 1000 * u["sample"], mappability=(args.config != "dels"))
    ds, de = synth.kept_sorted(c.del_start, c.del_end)
    us, ue = synth.kept_sorted(c.dup_start, c.dup_end)
    u.update(chrom=c, ds=ds, de=de, us=us, ue=ue, n_iv=len(ds) + len(us), n_reads=len(c.pos),
             sum_len=int((de.astype(np.int64) - ds).sum() + (ue.astype(np.int64) - us).sum()))
    return u


def upload_unit(u, ctx):
    """Make one chromosome resident in HBM behind the rank's batch context (the BAM loop's hand-over)."""
    c = u["chrom"]
    u["index"] = ctx.chrom_begin(c.length, c.gc)
    ctx.reads(c.pos, c.mapq)
    if c.map_start is not None:
        ctx.mappability(c.map_start, c.map_end, c.map_val)
    ctx.intervals("D", u["ds"], u["de"])
    if u["n_dups"]:
        ctx.intervals("E", u["us"], u["ue"])


def depth_kernel_bytes(units):
    """Algorithmic bytes of the depth_tile launch over these chromosomes: read_depth
    written once as int16, the tuples read once (int32 pos + uint8 mapq), one GC byte per window, one
    tile-index word per tile."""
    total = 0
    for u in units:
        L, n = u["length"], u["n_reads"]
        n_win = (L + 99) // 100
        n_tiles = (L + 2047) // 2048
        total += 2 * L + 5 * n + n_win + 4 * (n_tiles + 1)
    return total


def tuple_kernel_bytes(units):
    """Algorithmic bytes of one ingest_tuples launch over these chromosomes: every tuple read
    once (int32 pos + uint8 mapq) and each GC byte at most once."""
    return sum(5 * u["n_reads"] + (u["length"] + 99) // 100 for u in units)


def dense_reference_bytes(u, with_map):
    """bytes the reference's dense formulation touches for this chromosome."""
    L, n, sl, niv = u["length"], u["n_reads"], u["sum_len"], u["n_iv"]
    b = 2 * L + 9 * n + (2 * L + L // 100) + (2 * sl + sl // 100 + 64 * niv)
    if with_map:
        b += 4 * L + 4 * sl
    return b


def cpu_baseline(mine, ctx, args):
    """The oracle (a serial port of the reference's loops) timed on this host, 1 thread, on a bounded
    sample of the same workload; its results double as a full-size parity check of the HIP path."""
    from oracle import oracle as O
    O.lib()
    todo = sorted(mine, key=lambda u: u["length"])
    t_cpu, n_iv, names = 0.0, 0, []
    for u in todo:
        c = u["chrom"]
        t0 = time.perf_counter()
        rd, _ = O.count_reads(c.length, c.pos, c.mapq, -1)
        E, _, _ = O.calc_mean_per_chr(rd, c.gc)
        m = O.paint_mappability(c.length, c.map_start, c.map_end, c.map_val) if c.map_start is not None else None
        od = O.find_depths(rd, m, c.gc, E, "D", O.make_svs(u["ds"], u["de"]))
        ou = O.find_depths(rd, m, c.gc, E, "E", O.make_svs(u["us"], u["ue"]))
        t_cpu += time.perf_counter() - t0
        n_iv += u["n_iv"]
        names.append(u["name"])
        # parity at full size (the oracle is the checker here, never the thing measured above)
        ctx.select(u["index"])
        dels, dups, Eg, _ = ctx.fetch()
        assert np.array_equal(Eg.view(np.uint32), E.view(np.uint32)), "expected_read_depth mismatch"
        for got, want in ((dels, od), (dups, ou)):
            assert np.array_equal(got["observed"], want["observed"]), "observed mismatch chr" + u["name"]
            assert np.array_equal(got["expected"].view(np.uint32), want["expected"].view(np.uint32))
            assert np.array_equal(got["cn"], want["cn"]), "CN mismatch chr" + u["name"]
            assert np.allclose(got["lnone"], want["lnone"], rtol=0, atol=1e-6)
            assert np.allclose(got["score"], want["score"], rtol=0, atol=1e-6)
        if t_cpu >= args.cpu_seconds:
            break
    return dict(value=n_iv / t_cpu, unit="intervals/s", cores=1, kind="port",
                sample="chromosomes %s of the same workload (%d intervals, %.1f s, oracle/conga_oracle.c, "
                       "results compared with the HIP path)" % (",".join(names), n_iv, t_cpu))


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

    units = build_units(args, world)
    flags = capi.FLAG_BATCH | (capi.FLAG_MATERIALIZE_DEPTH if args.formulation == "dense" else 0)
    if dist_on or args.results_on_device:
        flags |= capi.FLAG_RESULTS_ON_DEVICE  # the records travel device-to-device into the RCCL gather, not over PCIe
    ctx = capi.Context(device=local_rank, flags=flags)  # every chromosome of this rank, one launch per kernel
    mine = [make_unit(u, args) for u in units if u["owner"] == rank]
    for u in mine:
        upload_unit(u, ctx)
    ctx.sync()
    rec = capi.RESULT_DTYPE.itemsize
    bytes_per_rank = [sum((u["n_dels"] + u["n_dups"]) * rec for u in units if u["owner"] == r) for r in range(world)]
    # interval counts after the min-size filter are only known to the owner: exchange them once
    my_bytes = sum(u["n_iv"] for u in mine) * rec
    if dist_on:
        t = torch.tensor([my_bytes], dtype=torch.int64, device=gather_dev)
        all_b = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_b, t)
        bytes_per_rank = [int(x.item()) for x in all_b]
    else:
        bytes_per_rank = [my_bytes]
    # Two sets of gather buffers, used alternately, padded to the largest contribution so that they go into the RCCL
    # gather as they are (no per-step allocation, no extra copy).  The gather of step k is asynchronous and overlaps the
    # compute of step k + 1; before step k + 2 copies its records into the same buffer, the context's stream is made to
    # wait for the event recorded behind gather k.
    pad = max(max(bytes_per_rank), 1)
    packed2 = [torch.zeros(pad, dtype=torch.uint8, device=dev) for _ in range(2)]
    recv2 = [[torch.empty(pad, dtype=torch.uint8, device=gather_dev) for _ in range(world)] if (rank == 0 and dist_on) else None
             for _ in range(2)]
    gathered_ev = [None, None]
    ext_stream = torch.cuda.ExternalStream(ctx.stream(), device=dev) if (dist_on and not rehearsal) else None
    total_iv = sum(bytes_per_rank) // rec
    step_no = [0]

    def step():
        slot = step_no[0] & 1
        packed = packed2[slot]
        step_no[0] += 1
        if ext_stream is not None and gathered_ev[slot] is not None:
            ext_stream.wait_event(gathered_ev[slot])      # the gather that read this buffer two steps ago is done
        ctx.compute()                               # whole hot path for this rank's chromosomes, async
        if dist_on:
            ctx.results_copy(packed.data_ptr(), my_bytes)   # records stay on the device for the RCCL gather
        ctx.sync()                                  # N=1: the records are in pinned host memory now
        if not dist_on:
            return [packed[:0]]
        if rehearsal:                               # one GPU, gloo: host tensors
            send = packed.cpu()
            dist.gather(send, recv2[slot], dst=0)
        else:
            dist.gather(packed, recv2[slot], dst=0)  # RCCL over xGMI; ordered behind the host-side sync above
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            gathered_ev[slot] = ev
        if rank != 0:
            return None
        return [recv2[slot][r][:bytes_per_rank[r]] for r in range(world)]

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps

    out = None
    if rank == 0:
        if dist_on:
            got = sum(g.numel() for g in gathered) // rec
            assert got == total_iv, (got, total_iv)
            if args.dist_selftest:
                # the gathered bytes are the records the context holds (un-permuted fetch order = results_copy order)
                ctx.compute()
                ctx.sync()
                want = b""
                for u in sorted(mine, key=lambda x: x["index"]):
                    ctx.select(u["index"])
                    dels, dups = ctx.fetch()[:2]
                    want += dels.tobytes() + dups.tobytes()
                have = gathered[0].cpu().numpy().tobytes()
                assert have == want, "gathered records differ from the fetched ones"

        # ---- roofline of the dominant HBM-bound kernel: HIP events recorded on the context's own stream around
        # every kernel (CONGA_FLAG_PROFILE), same resident inputs, one launch per kernel per compute
        def profile_kernels(c, reps=10):
            c.set_profile(True)
            kms = np.zeros(len(capi.KERNEL_NAMES))
            dense_ran = False
            for _ in range(reps):
                c.compute()
                c.select(0)
                st = c.fetch()[3]
                kms += np.array(st.kernel_ms[:len(capi.KERNEL_NAMES)])
                dense_ran = bool(st.depth_materialized)
            c.set_profile(False)
            return kms / reps, dense_ran

        def traffic_of(name):
            tpath = os.path.join(ROOT, "profiles", name)
            return json.load(open(tpath)).get("hbm_bytes_per_launch") if os.path.exists(tpath) else None

        def roofline_of(kernel, ms, nbytes, kms, traffic):
            achieved = nbytes / (max(ms, 1e-9) * 1e-3) / 1e9
            return dict(bound="hbm", kernel=kernel, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), algorithmic_bytes_per_launch=int(nbytes),
                        avg_launch_ms=round(float(ms), 5), traffic=traffic,
                        kernel_ms_per_step={k: round(float(v), 4) for k, v in zip(capi.KERNEL_NAMES, kms)})

        kms, dense_ran = profile_kernels(ctx)
        if dense_ran:
            roofline = roofline_of("depth_tile_kernel", kms[1], depth_kernel_bytes(mine), kms,
                                   traffic_of("depth_tile_traffic.json"))
        else:
            # the serial-float chain is latency-bound ("report its time separately"); the HBM-bound
            # kernel of the tuple-space path is the one pass over the tuples
            roofline = roofline_of("ingest_tuples_kernel", kms[0], tuple_kernel_bytes(mine), kms,
                                   traffic_of("ingest_tuples_traffic.json"))
            roofline["chain_ms"] = round(float(kms[6]), 4)

        dense = sum(dense_reference_bytes(u, args.config != "dels") for u in mine)
        cfg = dict(workload=("BASELINE configs[1]: GRCh37 autosomes 1-22, %d deletion rows (%d kept >= 1000 bp)%s, "
                             "%.1fx synthetic sample, 100-bp GC windows" % (
                                 synth.N_DELS_GENOME, total_iv // max(world if args.scaling == "weak" else 1, 1),
                                 "" if args.config == "dels" else " + dups + 100-mer-like mappability track",
                                 args.cov)),
                   samples=world if args.scaling == "weak" else 1, chromosomes_per_sample=len(units) // max(
                       world if args.scaling == "weak" else 1, 1),
                   intervals_per_step=int(total_iv), reads_rank0=int(sum(u["n_reads"] for u in mine)),
                   parallelism="chromosome-sharded x%d, one RCCL gather per step" % world,
                   dense_reference_bytes_rank0=int(dense))
        out = dict(metric="CNV intervals genotyped/sec (1000G Phase-3 set); CN-call concordance vs ref",
                   value=round(total_iv * args.steps / elapsed, 1), unit="intervals/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
                   higher_is_better=True, scaling=args.scaling, vs_baseline=None,
                   dtype="i16/i32+f32/f64" if dense_ran else "i32+f32/f64",  # counts, serial float32 chain, double scores
                   data="synthetic", config=cfg, roofline=roofline,
                   formulation="dense" if dense_ran else "tuple-space",
                   dense_equivalent=dict(bytes=int(dense), achieved=round(dense / (ms_per_step * 1e-3) / 1e9, 1),
                                         unit="GB/s", x_hbm_peak=round(dense / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                                         note="SURVEY.md 8d dense-formulation bytes / measured step time; above 1.0 "
                                              "only because the tuple-space formulation never moves them"))
        if world == 1:
            # never `value`: the same pass fed from host buffers (pageable numpy -> pinned ring -> H2D over PCIe,
            # layout upload, compute), i.e. what a caller holding decoded tuples in host memory sees
            t1 = time.perf_counter()
            ctx.reset()
            for u in mine:
                upload_unit(u, ctx)
            ctx.compute()
            ctx.sync()
            out["host_buffers_inclusive"] = dict(value=round(total_iv / (time.perf_counter() - t1), 1),
                                                 unit="intervals/s", note="one pass incl. PCIe staging; not the metric")
        if world == 1 and not dense_ran and args.dense_leg:
            # the reference's dense formulation on the same resident inputs: a second context with
            # CONGA_FLAG_MATERIALIZE_DEPTH, a few steps, and the roofline of its depth_tile kernel
            dctx = capi.Context(device=local_rank, flags=capi.FLAG_BATCH | capi.FLAG_MATERIALIZE_DEPTH)
            for u in mine:
                c = u["chrom"]
                dctx.chrom_begin(c.length, c.gc)
                dctx.reads(c.pos, c.mapq)
                if c.map_start is not None:
                    dctx.mappability(c.map_start, c.map_end, c.map_val)
                dctx.intervals("D", u["ds"], u["de"])
                if u["n_dups"]:
                    dctx.intervals("E", u["us"], u["ue"])
            for _ in range(3):
                dctx.compute()
            dctx.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                dctx.compute()
                dctx.sync()
            d_ms = (time.perf_counter() - t1) / 5 * 1e3
            dk, _ = profile_kernels(dctx, reps=5)
            out["roofline_dense"] = roofline_of("depth_tile_kernel", dk[1], depth_kernel_bytes(mine), dk,
                                                traffic_of("depth_tile_traffic.json"))
            out["roofline_dense"]["ms_per_step"] = round(d_ms, 4)
            out["roofline_dense"]["value"] = round(total_iv / (d_ms * 1e-3), 1)
            # both formulations must give the same records
            ctx.compute()
            ctx.sync()
            for u in mine:
                ctx.select(u["index"])
                dctx.select(u["index"])
                g1, g2 = ctx.fetch(), dctx.fetch()
                assert g1[0].tobytes() == g2[0].tobytes() and g1[1].tobytes() == g2[1].tobytes(), \
                    "tuple-space and dense records differ on chr" + u["name"]
            dctx.close()
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(mine, ctx, args)
            out["cn_concordance"] = 1.0  # asserted bit-exact against the oracle on the cpu_baseline sample

    ctx.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
