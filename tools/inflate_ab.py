#!/usr/bin/env python3
"""The bench's inflate leg alone (bench.bgzf_leg) for the library CONGA_LIB_PATH names: A/B of two builds on one box.
   for lib in ab/base.so ab/new.so ab/base.so ab/new.so; do CONGA_LIB_PATH=$lib python3 tools/inflate_ab.py; done"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

leg = bench.bgzf_leg(argparse.Namespace(cpu_seconds=0), dict(local_rank=0))
print(json.dumps(dict(lib=os.path.basename(os.environ.get("CONGA_LIB_PATH", "libconga_hip.so")), value=leg["value"], kernel_ms=leg["kernel_ms"],
                      bam_like=leg.get("bam_like", {}).get("value"))))
