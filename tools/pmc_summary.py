#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes per kernel.

usage: pmc_summary.py OUT.json DIR_OR_CSV [DIR_OR_CSV ...]

Each argument is a rocprofv3 output directory (or a *_counter_collection.csv) of ONE counter pass
(`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py ...`; separate passes per counter, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  Writes {kernel: {counter: {launches, mean_KB}}}:
FETCH_SIZE / WRITE_SIZE are reported by the hardware in KB.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def rows_of(path):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield r


def main():
    out_path, srcs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
    for src in srcs:
        for r in rows_of(src):
            name = r.get("Kernel_Name") or r.get("Kernel-Name")
            counter = r.get("Counter_Name")
            if not name or not counter:
                continue
            name = name.split("(")[0].replace("void ", "").strip()
            disp = r.get("Dispatch_Id") or r.get("Correlation_Id")
            acc[name][counter][disp] += float(r["Counter_Value"])  # one row per XCD / dimension: summed per dispatch
    summary = {}
    for name in sorted(acc):
        summary[name] = {}
        for counter, per in acc[name].items():
            vals = list(per.values())
            summary[name][counter] = dict(launches=len(vals), mean_KB=sum(vals) / len(vals))
    with open(out_path, "w") as fh:
        json.dump(summary, fh, indent=1)
    for name, c in summary.items():
        print(name[:60].ljust(60), {k: (v["launches"], round(v["mean_KB"], 1)) for k, v in c.items()})


if __name__ == "__main__":
    main()
