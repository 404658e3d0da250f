#!/usr/bin/env python3
"""The configs[4] leg of bench.py alone (conga_amd/rp_bench.py): split-read stage time on chromosomes --rp-chroms."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conga_amd import rp_bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rp-chroms", default="20,21,22")
ap.add_argument("--chroms", default="")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--no-cli", dest="rp_cli", action="store_false", help="the C-ABI part only (no BAM written, no `conga` runs): for profiler passes")
a = ap.parse_args()
out, _ = rp_bench.leg(a, dict(local_rank=0))
print(json.dumps(out))
