#!/usr/bin/env python3
"""Summary of the per-workgroup CSV written by a tools/tdbg_instrument.py build: durations per class (by decile of the
class's workgroups, longest chains first) and finishing time per compute unit."""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    h = int(r["hwid"])
    r["key"] = (int(r["xcc"]) & 0xF, (h >> 13) & 7, (h >> 8) & 0xF)
    r["start"], r["end"], r["wg"] = float(r["start_us"]), float(r["end_us"]), int(r["wg"])
    r["dur"] = r["end"] - r["start"]
for cls in ("X", "AB", "B", "C"):
    v = [r for r in rows if r["class"] == cls]
    if not v:
        continue
    n = len(v)
    dec = [round(statistics.mean(x["dur"] for x in v[i * n // 10:max((i + 1) * n // 10, i * n // 10 + 1)]), 1) for i in range(min(10, n))]
    print("%-2s %4d workgroups  start <= %.1f  duration deciles %s  last end %.1f" % (
        cls, n, max(x["start"] for x in v), dec, max(x["end"] for x in v)))
cus = collections.defaultdict(list)
for r in rows:
    cus[r["key"]].append(r)
ends = sorted((max(x["end"] for x in v), k) for k, v in cus.items())
print("%d compute units: last workgroup ends at min %.1f  mean %.1f  max %.1f us" % (
    len(cus), ends[0][0], statistics.mean(e for e, _ in ends), ends[-1][0]))
same = sum(1 for v in cus.values() if len({x["wg"] % 256 for x in v}) == 1)
print("compute units whose workgroups all share (index mod 256): %d" % same)
for e, k in ends[-5:]:
    v = cus[k]
    print("  slow CU %s end %.1f: %s" % (k, e, [(x["wg"], x["class"], round(x["dur"], 1)) for x in v]))
