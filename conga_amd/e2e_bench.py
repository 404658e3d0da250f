"""End to end from BAM files, as a leg of bench.py: `conga --cohort` over whole-genome 1x BAMs written on the spot.

SURVEY.md 8d: "an end-to-end figure including BAM decode is reported separately"; the reference's BAM loop is
bam_data.c:253-339 (htslib inflates and parses every record on one core, count_reads_bam takes pos / qual from it).  Here the
samples of the bench's main leg (BASELINE configs[1]: 22 autosomes, 1x, the 1000G-sized deletion set) are written as BAM + .bai
by tools/bamwrite (all cores, zlib level 1, pseudo-random bases and run-structured qualities), and the `conga` executable
genotypes a list of them in ONE process: the first sample pays the start of the HIP runtime, the engine context and the
layout, every further one is the BAM stage (upload, inflate, record walk on the GPU) + two launches + the three output files.

  first_sample_s        wall time of `conga --cohort` over a list of one BAM (process start to exit)
  per_further_sample_ms in a run over a list of K: (end of sample K - end of sample 1) / (K - 1) by the process's own clock
both with the decode on the GPU (conga_reads_bgzf) and with the host decoders (CONGA_GPU_BAM=0).  The files just written are in
the page cache: the input side is memory, not a disk.  The outputs of the two decoders are compared byte for byte, and the
OBSERVED_READS column of sample 0 against the records the tuple route computed for the same sample (which bench.py compares
with the oracle).
"""
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

from . import formats, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONGA = os.path.join(ROOT, "conga_amd", "host", "conga")
BAMWRITE = os.path.join(ROOT, "tools", "bamwrite")


def write_bam(d, tag, chroms, l_seq=100, level=1, sample=None):
    """chroms: [(name, length, pos int32[], mapq uint8[][, flag uint16[], seq uint8[n, (l+1)//2], qual uint8[n, l]])] -> path, seconds"""
    lines = []
    for c in chroms:
        name = c[0]
        paths = []
        for k, (arr, dt) in enumerate(zip(c[2:], ("<i4", "u1", "<u2", "u1", "u1"))):
            if arr is None:
                paths.append("-")
                continue
            p = os.path.join(d, "%s_%s_%d.bin" % (tag, name, k))
            np.ascontiguousarray(arr, dtype=dt).tofile(p)
            paths.append(p)
        paths += ["-"] * (5 - len(paths))
        lines.append("%s %d %d %s %d" % (name, c[1], len(c[2]), " ".join(paths), l_seq))
    man = os.path.join(d, tag + ".manifest")
    with open(man, "w") as f:
        f.write("\n".join(lines) + "\n")
    out = os.path.join(d, tag + ".bam")
    t0 = time.perf_counter()
    r = subprocess.run([BAMWRITE, out, man, "--level", str(level), "--sample", sample or tag], capture_output=True, text=True)
    if r.returncode != 0:
        raise OSError("tools/bamwrite failed: " + r.stderr[-500:])
    dt = time.perf_counter() - t0
    if os.environ.get("CONGA_TIMING"):
        import resource
        ru = resource.getrusage(resource.RUSAGE_CHILDREN)
        print("[timing] %s: %.2f s (children so far: user %.1f s, sys %.1f s)" % (r.stderr.strip(), dt, ru.ru_utime, ru.ru_stime), file=sys.stderr)
    for line in lines:
        for p in line.split()[3:8]:
            if p != "-":
                os.remove(p)
    os.remove(man)
    return out, dt


def run_conga(argv, cwd, env_extra):
    env = dict(os.environ, **env_extra)
    t0 = time.perf_counter()
    r = subprocess.run([CONGA] + argv, cwd=cwd, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError("conga failed (%d): %s" % (r.returncode, r.stderr[-800:]))
    return dt, r.stderr


def cohort_times(d, bams, k_many, common, env_extra, tag, repeats=2):
    """-> (first_sample_s, per_further_sample_ms, stderr of the long run): lists of 1 and of k_many BAMs, best of `repeats`"""
    one, many = os.path.join(d, tag + "_1.txt"), os.path.join(d, tag + "_k.txt")
    with open(one, "w") as f:
        f.write("%s\t%s_one\n" % (bams[0], tag))
    with open(many, "w") as f:
        for k in range(k_many):
            f.write("%s\t%s_s%d\n" % (bams[k % len(bams)], tag, k))
    t1 = min(run_conga(["--cohort", one, "--out", tag] + common, d, env_extra)[0] for _ in range(repeats))
    best, err, per, steady = 1e30, "", 1e30, 1e30
    for _ in range(repeats):
        t, e = run_conga(["--cohort", many, "--out", tag] + common, d, env_extra)
        # the process's own clock at every sample's end (CONGA_TIMING): a further sample = (last - first) / (K - 1), free of how
        # long the HIP runtime took to come up this time (0.12-0.38 s from run to run on these boxes)
        import re
        done = [float(x) for x in re.findall(r"cohort: sample \d+ of \d+ is done ([0-9.]+) ms", e)]
        assert len(done) == k_many, e[-1500:]
        per = min(per, (done[-1] - done[0]) / (k_many - 1))   # (ramp and drain of the two-deep pipeline included: what a cohort of K costs)
        deltas = sorted(b - a for a, b in zip(done[3:-1], done[4:]))   # (samples 5 .. K: the pipeline is full)
        if deltas:
            steady = min(steady, deltas[len(deltas) // 2])
        if t < best:
            best, err = t, e
            cohort_times.span_ms = done[-1] - done[0]
    cohort_times.steady_state_ms = None if steady > 1e29 else steady
    if os.environ.get("CONGA_BENCH_STDERR_DIR"):   # (the [timing] lines of the long run, for whoever wants the stages)
        with open(os.path.join(os.environ["CONGA_BENCH_STDERR_DIR"], "conga_cohort_%s.err" % tag), "w") as f:
            f.write(err)
    return t1, per, best, err


def zlib_one_core(path, budget_s=1.5):
    """zlib on one core over blocks of the file, front to back, for about budget_s: -> (GB/s inflated, inflated bytes of the whole
    file by its ISIZE fields, blocks inflated)"""
    size = os.path.getsize(path)
    inflated_total, done, n = 0, 0, 0
    t_used = 0.0
    with open(path, "rb") as f:
        buf = f.read(min(size, 256 << 20))
        at = 0
        while at + 18 <= len(buf) and t_used < budget_s:
            bsize = struct.unpack_from("<H", buf, at + 16)[0] + 1
            if at + bsize > len(buf):
                break
            t0 = time.perf_counter()
            raw = zlib.decompress(buf[at + 18:at + bsize - 8], -15)
            zlib.crc32(raw)
            t_used += time.perf_counter() - t0
            done += len(raw)
            n += 1
            at += bsize
        ratio = done / max(at, 1)
    inflated_total = int(size * ratio)   # (the blocks are alike: the file's inflated size from the sample's ratio)
    return done / max(t_used, 1e-9) / 1e9, inflated_total, n


def leg(args, env, mine, recs0, cpu_intervals_per_s, k_many=30):
    """mine: the main leg's units (layout + three samples' tuples); recs0: the records the tuple route computed for sample 0."""
    if not (os.path.exists(CONGA) and os.path.exists(BAMWRITE)):
        return dict(error="conga / tools/bamwrite are not built")
    d = tempfile.mkdtemp(prefix="conga_bench_e2e_", dir=os.environ.get("CONGA_BENCH_TMP", "/tmp"))
    try:
        n_iv = int(sum(u["n_iv"] for u in mine))
        formats.write_annotation(os.path.join(d, "a.cga"), [(u["name"], u["length"], u["chrom"].gc, [], []) for u in mine])
        synth.write_bed(os.path.join(d, "dels.bed"), [(u["name"], s, e) for u in mine for s, e in zip(u["chrom"].del_start, u["chrom"].del_end)])
        bams, t_write = [], 0.0
        for j in range(2):   # two different individuals, listed alternately
            p, dt = write_bam(d, "s%d" % j, [(u["name"], u["length"], u["reads"][j][0], u["reads"][j][1]) for u in mine])
            bams.append(p)
            t_write += dt
        for p in bams * 3:   # (a file's pages are promoted in the kernel's lists on their second and third reading, at a price that is
            with open(p, "rb") as f:   # not this path's: the timed runs below read settled pages, as they would read a file once)
                while f.read(64 << 20):
                    pass
        size = os.path.getsize(bams[0])
        n_reads = int(sum(len(u["reads"][0][0]) for u in mine))
        common = ["--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed"]
        out = dict(workload="%d whole-genome 1x BAMs (%s; %.2f GB each, %d reads, zlib level 1, pseudo-random bases, run-structured "
                            "qualities; written by tools/bamwrite in %.1f s, in the page cache) through `conga --cohort`: chromosomes %s, "
                            "%d deletion intervals per sample; three output files per sample" % (
                                k_many, "two individuals alternating", size / 1e9, n_reads, t_write,
                                "1-22" if len(mine) == 22 else ",".join(u["name"] for u in mine), n_iv),
                   intervals_per_sample=n_iv, bam_bytes=size)
        res = {}
        for decode, envx in (("gpu", dict(CONGA_GPU_BAM="1")), ("host", dict(CONGA_GPU_BAM="0"))):
            k = k_many if decode == "gpu" else 3
            t1, per, t_k, err = cohort_times(d, bams, k, common, dict(envx, CONGA_TIMING="1"), decode)
            assert ("decoding on the host" not in err) and (err.count("conga_reads_bgzf:") == (k if decode == "gpu" else 0)), err[-1500:]
            res[decode] = dict(decode=decode, first_sample_s=round(t1, 3), per_further_sample_ms=round(per, 1), samples=k, wall_s=round(t_k, 3),
                               intervals_per_s=round(n_iv / (per * 1e-3), 1))
            # ... and by the caller's clock, everything in: what the K-sample process costs beyond a one-sample process, per further sample --
            # the ramp of the pipeline, the second set of buffers and the teardown of a process that holds them
            res[decode]["per_further_sample_wall_ms"] = round(1e3 * (t_k - t1) / max(k - 1, 1), 1)
            res[decode]["intervals_per_s_all_in"] = round(n_iv / max((t_k - t1) / max(k - 1, 1), 1e-9), 1)
            res[decode]["fixed_cost_ms"] = round(1e3 * (t_k - t1) - cohort_times.span_ms, 1)
            if decode == "gpu" and cohort_times.steady_state_ms:
                res[decode]["steady_state_ms"] = round(cohort_times.steady_state_ms, 1)
                res[decode]["note"] = ("per_further_sample_ms: (end of sample K - end of sample 1) / (K - 1), the pipeline's first samples -- its second "
                                       "set of buffers is allocated while they run -- and its drain included; steady_state_ms: the median over samples 5 .. K; "
                                       "per_further_sample_wall_ms: (wall of the K-sample process - wall of a one-sample process) / (K - 1) by the CALLER's clock; "
                                       "fixed_cost_ms: what of that difference is not between the first and the last sample's end -- a process that holds the "
                                       "pipeline's 45 GB of buffers takes the driver longer to take down, and leaves later")
        # the two decoders wrote the same files; sample 0's observed depths are the tuple route's
        for k in range(3):
            for kind in ("svs", "dels"):
                a = open(os.path.join(d, "gpu_s%d_%s.bed" % (k, kind)), "rb").read()
                b = open(os.path.join(d, "host_s%d_%s.bed" % (k, kind)), "rb").read()
                assert a == b and len(a) > 100, "GPU decode and host decoders differ: sample %d %s" % (k, kind)
        rows = open(os.path.join(d, "gpu_s0_dels.bed")).read().splitlines()[1:]
        assert len(rows) == n_iv == len(recs0), (len(rows), n_iv, len(recs0))
        obs = np.array([int(r.split("\t")[7]) for r in rows], np.int32)
        assert np.array_equal(obs, recs0["observed"]), "OBSERVED_READS from the BAM differs from the tuple route's records"
        exp = np.array([float(r.split("\t")[8]) for r in rows])
        assert np.allclose(exp, recs0["expected"].astype(np.float64), rtol=0, atol=0.051), "EXPECTED_READS (%.1f) differs"
        out.update(end_to_end=res["gpu"], end_to_end_host_decoders=res["host"],
                   checked="the three files of every sample byte-identical between the two decoders; OBSERVED_READS / EXPECTED_READS of "
                           "sample 0's %d rows equal to the records of the tuple route (which are compared with the oracle)" % n_iv)
        # what the reference pays in its BAM loop, on one core of this host: zlib over the same blocks + the oracle's compute
        gbs, inflated, nb = zlib_one_core(bams[0])
        t_inflate = inflated / (gbs * 1e9)
        t_oracle = n_iv / max(cpu_intervals_per_s, 1e-9) if cpu_intervals_per_s else None
        cpu = dict(kind="port", cores=1, host_cores=os.cpu_count(), zlib_gbs_inflated=round(gbs, 3), inflate_s=round(t_inflate, 2),
                   sample="zlib inflate + CRC32 of the first %d BGZF blocks of the same file on one core, scaled to its %.2f GB inflated"
                          % (nb, inflated / 1e9))
        if t_oracle is not None:
            cpu.update(oracle_s=round(t_oracle, 2), value=round(n_iv / (t_inflate + t_oracle), 1), unit="intervals/s",
                       note="inflate + the oracle's compute for one sample (record parsing and BED loading not counted: they favour the CPU)")
        out["cpu_baseline"] = cpu
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)
