import os, subprocess, sys, tempfile
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from conga_amd import formats, synth
d = tempfile.mkdtemp(prefix="longrec_")
rng = np.random.default_rng(5)
cs = [synth.make_chrom(n, L, cov=2.0, n_dels=nd, gaps=True) for n, L, nd in (("1", 400_000, 15), ("2", 300_000, 10))]
recs = {}
chroms = []
for c in cs:
    n = len(c.pos)
    lq = np.full(n, 100, np.int32)
    big = rng.choice(n, 25, replace=False)
    lq[big] = rng.integers(70_000, 200_000, 25)          # records of 100-300 KB: several BGZF blocks each
    off = np.concatenate([[0], np.cumsum(lq)[:-1]]).astype(np.uint64)
    total = int(lq.sum())
    codes = rng.choice(np.array([1, 2, 4, 8], np.uint8), total)
    qual = rng.integers(2, 41, total).astype(np.uint8)
    recs[c.name] = (lq, codes, qual, off)
    chroms.append((c.name, c.length, c.pos, c.mapq))
formats.write_bam(os.path.join(d, "r.bam"), "S", chroms, records=recs, index=True, block_payload=40000, unplaced=2)
formats.write_annotation(os.path.join(d, "a.cga"), [(c.name, c.length, c.gc, [], []) for c in cs])
synth.write_bed(os.path.join(d, "dels.bed"), [(c.name, s, e) for c in cs for s, e in zip(c.del_start, c.del_end)])
conga = os.path.join(R, "conga_amd", "host", "conga")
outs = {}
for tag, env in (("gpu", {"CONGA_GPU_BAM": "1"}), ("host", {"CONGA_GPU_BAM": "0"})):
    r = subprocess.run([conga, "-i", "r.bam", "--ref", "r.fa", "--sonic", "a.cga", "--dels", "dels.bed", "--out", tag], cwd=d,
                       capture_output=True, text=True, env=dict(os.environ, **env), timeout=300)
    print(tag, r.returncode, "decoding on the host" in r.stderr, r.stderr.strip().splitlines()[-1][:100])
    outs[tag] = open(os.path.join(d, tag + "_dels.bed"), "rb").read() if r.returncode == 0 else None
print("same:", outs["gpu"] == outs["host"], "size", os.path.getsize(os.path.join(d, "r.bam")))
