import sys, os, zlib
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from conga_amd import capi
import test_gpu_inflate as T
only = int(sys.argv[1])
rng = np.random.default_rng(20261004)
kinds = T.payloads(rng)
streams = []
for k in range(3600):
    plain = kinds[k % len(kinds)]
    a = int(rng.integers(0, max(len(plain) - 10, 1)))
    plain = plain[a:a + int(rng.integers(1, 6000))]
    good = T.deflate(plain, int(rng.integers(1, 10)), int(rng.choice([0, 0, 0, 2, 3, 4])))
    bad = bytearray(good)
    how = k % 6
    if how == 0:
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
    elif how == 1:
        bad[int(rng.integers(0, min(len(bad), 40)))] ^= 1 << int(rng.integers(0, 8))
    elif how == 2:
        for _ in range(int(rng.integers(1, 5))):
            bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
    elif how == 3:
        del bad[int(rng.integers(0, len(bad))):]
        bad += b"\x00" if not bad else b""
    elif how == 4:
        other = T.deflate(kinds[(k + 3) % len(kinds)][:3000], 6)
        cut = int(rng.integers(1, len(bad) + 1))
        bad = bad[:cut] + other[int(rng.integers(0, len(other))):]
    else:
        noise = bytes(rng.integers(0, 256, int(rng.integers(4, 400)), dtype=np.uint8))
        bad = bytearray((b"\x05" if k % 12 == 5 else b"") + noise)
    if how == only:
        streams.append((bytes(bad), plain))
data, blocks = T.pack(streams)
print("category", only, len(blocks), "blocks", flush=True)
with capi.Context(device=0) as ctx:
    _o, status, _ = ctx.inflate_blocks(data, blocks, want_out=False)
print("category", only, "done; accepted", int((status == 0).sum()), flush=True)
