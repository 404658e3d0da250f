"""The block inflater of the BAM reader (conga_amd/host/inflate_fast.cpp) against zlib: thousands of generated raw
deflate streams (every level / strategy / block type, sizes 0 .. 65 280, several kinds of data), wrong output sizes,
truncated and bit-flipped streams -- also under AddressSanitizer + UBSan."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests", "native", "inflate_check.cpp"), os.path.join(ROOT, "conga_amd", "host", "inflate_fast.cpp")]


def build(tmp_path, flags):
    exe = os.path.join(str(tmp_path), "inflate_check")
    subprocess.check_call(["g++", "-std=c++17", "-g"] + flags + ["-o", exe] + SRC + ["-lz"])
    return exe


@pytest.mark.parametrize("flags,seed,n", [
    (["-O2"], 1, 3000),
    (["-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"], 2, 1200),
])
def test_inflate_raw_equals_zlib(tmp_path, flags, seed, n):
    exe = build(tmp_path, flags)
    env = dict(os.environ, ASAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, str(seed), str(n)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert r.stdout.startswith("ok %d " % n)
