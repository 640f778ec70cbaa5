#!/usr/bin/env python3
"""rocprofv3 kernel trace CSV -> how busy the GPU is inside the steady-state steps:
fraction of wall time with 0 / 1 / >=2 kernels in flight, and mean step period."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# keep the last 60 % of the trace (steady-state graph replays)
t0 = rows[int(len(rows) * 0.4)][0]
rows = [r for r in rows if r[0] >= t0]
ev = []
for s, e, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0}
cur, last = 0, ev[0][0]
for t, d in ev:
    busy[min(cur, 2)] += t - last
    cur += d; last = t
tot = sum(busy.values())
print("kernels %d  span %.2f ms  in flight: 0 -> %.1f %%, 1 -> %.1f %%, >=2 -> %.1f %%"
      % (len(rows), tot / 1e6, 100 * busy[0] / tot, 100 * busy[1] / tot, 100 * busy[2] / tot))
ksum = sum(e - s for s, e, _ in rows)
print("sum of kernel durations / span = %.2f" % (ksum / tot))
