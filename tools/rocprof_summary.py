"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py` into per-kernel steady-state numbers.

    python tools/rocprof_summary.py <..._kernel_trace.csv> --warmup W --steps K > profiles/<name>.md

bench.py ends every step with one `postprocess_kernel` dispatch; the plan's one-off autotune launches and
the warm-up steps precede the (W)th of them, the K timed steps lie between postprocess #W and #(W+K).
Only those dispatches are summarised, so the per-kernel averages are comparable with bench.py's own
hipEvent numbers."""
import argparse
import collections
import csv
import re


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("ccvpe::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warmup", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "postprocess_kernel" in r["Kernel_Name"]]
    assert len(marks) >= a.warmup + a.steps, f"only {len(marks)} steps in the trace"
    lo = marks[a.warmup - 1] + 1 if a.warmup > 0 else 0
    hi = marks[a.warmup + a.steps - 1] + 1
    sel = rows[lo:hi]
    t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
    agg = collections.OrderedDict()
    for r in sel:
        k = short(r["Kernel_Name"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        e = agg.setdefault(k, [0, 0, 1 << 62, 0, r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"]])
        e[0] += 1; e[1] += d; e[2] = min(e[2], d); e[3] = max(e[3], d)
    busy = sum(e[1] for e in agg.values())
    print(f"# rocprofv3 --kernel-trace: {a.steps} timed steps (after {a.warmup} warm-up steps and the one-off autotune)\n")
    print(f"wall span of the timed steps: {(t1 - t0) / 1e6:.3f} ms = {(t1 - t0) / 1e6 / a.steps:.3f} ms/step; "
          f"sum of kernel durations {busy / 1e6 / a.steps:.3f} ms/step\n")
    print("| kernel | launches/step | avg us | min us | max us | ms/step | % of kernel time | VGPR | AGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {e[0] / a.steps:.1f} | {e[1] / e[0] / 1e3:.1f} | {e[2] / 1e3:.1f} | {e[3] / 1e3:.1f} | "
              f"{e[1] / 1e6 / a.steps:.3f} | {100.0 * e[1] / busy:.1f} | {e[4]} | {e[5]} | {e[6]} |")


if __name__ == "__main__":
    main()
