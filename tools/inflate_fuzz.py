import os, sys, zlib
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conga_amd import capi
import test_gpu_inflate as T
bad_total = 0
for seed in range(101, 113):
    rng = np.random.default_rng(seed)
    kinds = T.payloads(rng)
    streams = []
    for k in range(3000):
        plain = kinds[k % len(kinds)]
        a = int(rng.integers(0, max(len(plain) - 10, 1)))
        plain = plain[a:a + int(rng.integers(1, 8000))]
        good = T.deflate(plain, int(rng.integers(0, 10)), int(rng.choice([0, 0, 0, 1, 2, 3, 4])))
        bad = bytearray(good)
        how = int(rng.integers(0, 7))
        if how == 0:
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 1:
            bad[int(rng.integers(0, min(len(bad), 60)))] ^= 1 << int(rng.integers(0, 8))
        elif how == 2:
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        elif how == 3:
            del bad[int(rng.integers(0, len(bad))):]
            bad += b"\x00" if not bad else b""
        elif how == 4:
            other = T.deflate(kinds[(k + 3) % len(kinds)][:3000], int(rng.integers(1, 10)))
            cut = int(rng.integers(1, len(bad) + 1))
            bad = bad[:cut] + other[int(rng.integers(0, len(other))):]
        elif how == 5:
            bad = bytearray(bytes(rng.integers(0, 256, int(rng.integers(1, 600)), dtype=np.uint8)))
        # how == 6: left as it is (a valid stream among the damaged ones)
        streams.append((bytes(bad), plain))
    data, blocks = T.pack(streams)
    want = []
    for off, n, isz, crc in blocks:
        try:
            got = zlib.decompress(data[off:off + n].tobytes(), -15)
            want.append(len(got) == isz and (zlib.crc32(got) & 0xFFFFFFFF) == crc)
        except zlib.error:
            want.append(False)
    with capi.Context(device=0) as ctx:
        _o, status, _ = ctx.inflate_blocks(data, blocks, want_out=False)
    wrong = [k for k in range(len(blocks)) if bool(status[k] == 0) != want[k]]
    bad_total += len(wrong)
    print("seed", seed, "accepted", sum(want), "of", len(want), "wrong", len(wrong), wrong[:5], flush=True)
print("inflate fuzz:", 12 * 3000, "streams,", bad_total, "disagreements with zlib")
sys.exit(1 if bad_total else 0)
