#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs -> <tag>_pmc_counters.csv (mean per kernel and level) and
<tag>_pmc_traffic.json (fabric-side bytes per op call for every level's kernels).

    pmc_to_traffic.py <dir with _pmc_L<level>_<counter>/ runs> <tag> [<git commit>]

Correction prescribed by MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports
half of the bytes of a wide coalesced read stream -> read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-byte streaming stores -> write bytes = WRITE_SIZE * 1024.
(The warp kernels gather with 4- and 8-byte loads: their factor is calibrated separately,
<tag>_fetch_size_calibration.json, and quoted in profiles/README.md.)"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else None
REPS = 5  # tools/prof_kernels.py --reps


def label(name, lvl):
    for key in ("corr_fwd", "corr_bwd", "warp_fwd", "warp_bwd"):
        if key in name:
            return "%s_L%d" % (key, lvl)
    return None


per_dispatch = defaultdict(lambda: defaultdict(list))   # label -> counter -> values per dispatch
names = defaultdict(lambda: defaultdict(list))
kernels = {}
for path in glob.glob(out_dir + "/_pmc_L*/**/*counter_collection.csv", recursive=True):
    lvl = int(re.search(r"_pmc_L(\d)_", path).group(1))
    for row in csv.DictReader(open(path)):
        lab = label(row["Kernel_Name"], lvl)
        short = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void cerb::(anonymous namespace)::", ""))[:100]
        names[(lvl, short)][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if lab:
            per_dispatch[lab][row["Counter_Name"]].append(float(row["Counter_Value"]))
            kernels[lab] = short

with open("%s/%s_pmc_counters.csv" % (out_dir, tag), "w") as f:
    f.write("level,kernel,counter,mean_per_dispatch,dispatches\n")
    for (lvl, k) in sorted(names):
        for c in sorted(names[(lvl, k)]):
            v = names[(lvl, k)][c]
            f.write('%d,"%s",%s,%.6g,%d\n' % (lvl, k, c, sum(v) / len(v), len(v)))

traffic = {"_note": "bytes per op call at 4 pairs; read = 2*FETCH_SIZE KiB (gfx950 correction for wide "
                    "coalesced streams), write = WRITE_SIZE KiB; the warp backward is ONE launch "
                    "(grad_image tiles + grad_flow strips)",
           "_commit": commit, "_kernels": kernels}
for lab, ctr in sorted(per_dispatch.items()):
    if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
        rd = 2.0 * sum(ctr["FETCH_SIZE"]) / REPS * 1024.0
        wr = sum(ctr["WRITE_SIZE"]) / REPS * 1024.0
        traffic[lab] = {"read_bytes": round(rd), "write_bytes": round(wr), "traffic_bytes": round(rd + wr)}
json.dump(traffic, open("%s/%s_pmc_traffic.json" % (out_dir, tag), "w"), indent=1)
print(json.dumps(traffic, indent=1))
