"""Dev tool: concurrency statistics of a rocprofv3 --kernel-trace CSV of bench.py (two-stream plans):
wall span of the timed steps, time with >= 1 and >= 2 kernels in flight, per-kernel summed durations."""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
warm, steps = int(sys.argv[2]), int(sys.argv[3])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "postprocess_kernel" in r["Kernel_Name"]]
lo = marks[warm - 1] + 1
hi = marks[warm + steps - 1] + 1
sel = rows[lo:hi]
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy1 = busy2 = 0
cur = 0
prev = t0
for t, d in ev:
    if cur >= 1:
        busy1 += t - prev
    if cur >= 2:
        busy2 += t - prev
    cur += d
    prev = t
print(f"wall {(t1 - t0) / 1e6 / steps:.3f} ms/step; >=1 kernel in flight {busy1 / 1e6 / steps:.3f} ms/step; >=2 in flight {busy2 / 1e6 / steps:.3f} ms/step; "
      f"sum of durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / 1e6 / steps:.3f} ms/step")
agg = collections.Counter()
for r in sel:
    agg[r["Kernel_Name"].split("(")[0][:60]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in agg.most_common(12):
    print(f"  {v / 1e6 / steps:8.3f} ms/step  {k}")
