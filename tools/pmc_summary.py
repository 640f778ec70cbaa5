#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel name."""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(corr_(?:fwd|bwd)\w*_kernel|warp_\w+_kernel|\w+_kernel)(<.*)?", name)
    if not m:
        return name[:60]
    base = m.group(1)
    cfg = re.search(r"Cfg<([^>]*)>", name)
    return base + ("<" + cfg.group(1) + ">" if cfg else "")


acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s mean=%.5g  n=%d" % (c, sum(v) / len(v), len(v)))
