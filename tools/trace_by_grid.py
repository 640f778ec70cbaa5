#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv -> average duration per (kernel, grid size, workgroup size).
The same template instantiation is launched for several pyramid levels; the grid size tells
them apart (e.g. corr_bwd_d4_kernel<BwdCfg<32,2,72>> with 131072 work-items is level 3 at 4 pairs)."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("cerb::(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*", "", name)
    key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
    acc[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel,grid_size,workgroup_size,calls,avg_ns,min_ns,max_ns")
for (k, g, w), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%s,%s,%d,%.0f,%d,%d' % (k, g, w, len(v), sum(v) / len(v), min(v), max(v)))
