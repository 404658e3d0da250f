import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from conga_amd import formats, synth
lens = dict(synth.GRCH37_AUTOSOMES)
cs = [synth.make_chrom(n, lens[n], cov=1.0) for n in sys.argv[2].split(",")]
formats.write_bam_fast(sys.argv[1], "S", [(c.name, c.length, c.pos, c.mapq) for c in cs], realistic=True, level=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
