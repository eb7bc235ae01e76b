"""Dev tool (GPU box): batch-1 frames under rocprofv3 --kernel-trace: per-frame kernel time, gaps and the longest kernels."""
import csv, glob, sys, collections, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "postprocess_kernel" in r["Kernel_Name"]]
if len(marks) < 12:
    sys.exit(f"{len(rows)} dispatches, {len(marks)} frames in the trace: run rocprofv3 with --output-format csv on tools/b1_frames.py")
lo, hi = marks[-12] + 1, marks[-2] + 1          # ten frames near the end
sel = rows[lo:hi]
nfr = 10
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
busy = 0; ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort(); depth = 0; last = t0; cover = 0
for t, d in ev:
    if depth > 0: cover += t - last
    depth += d; last = t
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
print(f"{len(sel) / nfr:.0f} launches/frame; wall {(t1 - t0) / 1e3 / nfr:.1f} us/frame; >=1 kernel in flight {cover / 1e3 / nfr:.1f} us/frame; sum of durations {dur / 1e3 / nfr:.1f} us/frame")
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    k = re.sub(r"\(.*\)$", "", r["Kernel_Name"].replace("void ", "").replace("ccvpe::", ""))
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {k[:70]:70s} n/frame {n / nfr:5.1f}  avg {d / n / 1e3:6.1f} us  {d / 1e3 / nfr:7.1f} us/frame")
