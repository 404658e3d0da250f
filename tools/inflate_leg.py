#!/usr/bin/env python3
"""bench.py's bgzf_inflate leg alone (conga_inflate_blocks on the bench's two inputs): one JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

a = argparse.Namespace(cpu_seconds=0.0)
out = bench.bgzf_leg(a, dict(local_rank=0))
print(json.dumps({k: out[k] for k in ("blocks", "kernel_ms", "value")} | {"bam_like": {k: out["bam_like"].get(k) for k in ("kernel_ms", "value")}}))
