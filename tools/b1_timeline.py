"""Dev tool: one replayed batch-1 frame from a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue, kernel; plus the
idle gaps (no kernel in flight).  usage: python tools/b1_timeline.py <trace dir> [frame index from the end, default 3]"""
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "postprocess_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo, hi = marks[-k - 1] + 1, marks[-k] + 1
sel = rows[lo:hi]
t0 = int(sel[0]["Start_Timestamp"])
last_end = t0
idle = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*\)$", "", r["Kernel_Name"].replace("void ", "").replace("ccvpe::", ""))[:60]
    gap = s - last_end
    if gap > 0:
        idle += gap
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{r.get('Queue_Id', '?'):>3s} {'GAP %.1f' % (gap / 1e3) if gap > 500 else '':10s} {name}")
    last_end = max(last_end, e)
print(f"frame {(last_end - t0) / 1e3:.1f} us, idle {idle / 1e3:.1f} us, {len(sel)} kernels")
