import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from conga_amd import capi
n = 25_600_000
rng = np.random.default_rng(1)
posn = np.cumsum(rng.geometric(0.01, n)).astype(np.int32)
off = np.array([0, n], np.uint64)
with capi.Context(device=0) as ctx, capi.Packer(8) as pk:
    for src_pinned in (False, True):
        for dst_pinned in (False, True):
            pos = ctx.host_alloc(n, np.int32) if src_pinned else np.empty(n, np.int32)
            pos[:] = posn
            out = ctx.host_alloc(pk.bound(n, n // 16), np.uint8) if dst_pinned else np.zeros(pk.bound(n, n // 16), np.uint8)
            out[:] = 0
            best = 1e9
            for _ in range(10):
                t0 = time.perf_counter()
                pk.start(pos, off, out, 10)
                pk.finish()
                best = min(best, time.perf_counter() - t0)
            print("src pinned %s dst pinned %s: %.3f ms" % (src_pinned, dst_pinned, best * 1e3), flush=True)
