"""Summarise tools/pmc_forward.sh: per kernel (last step of every pass) launches, time (plain --kernel-trace pass), FETCH_SIZE /
WRITE_SIZE (rocprofv3 --pmc passes), HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies the 128-byte read requests of wide
coalesced loads at 64 B: MI355X_MICROARCH.md, HBM) and the resulting GB/s against 8 TB/s spec / 6.3 TB/s achievable."""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("ccvpe::", "")
    return re.sub(r"\(.*\)$", "", name)


def last_step(rows, key_start):
    rows.sort(key=key_start)
    marks = [i for i, r in enumerate(rows) if "postprocess_kernel" in r["Kernel_Name"]]
    lo = marks[-2] + 1 if len(marks) >= 2 else 0
    return rows[lo:marks[-1] + 1]


def main():
    out = sys.argv[1]
    counters = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        per_disp = collections.OrderedDict()
        for r in rows:
            per_disp.setdefault(r["Dispatch_Id"], {"Kernel_Name": r["Kernel_Name"], "id": int(r["Dispatch_Id"])})[r["Counter_Name"]] = float(r["Counter_Value"])
        disp = list(per_disp.values())
        for d in last_step(disp, lambda r: r["id"]):
            for k, v in d.items():
                if k not in ("Kernel_Name", "id"):
                    counters[short(d["Kernel_Name"])][k] += v
    times = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        for r in last_step(rows, lambda r: int(r["Start_Timestamp"])):
            e = times[short(r["Kernel_Name"])]
            e[0] += 1
            e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("| kernel | launches/step | us/step | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM-side MB (2F+W) | GB/s | of 8 TB/s | of 6.3 TB/s | MFMA busy | VALU/MFMA | LDS conflict |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    tot_us = sum(e[1] for e in times.values())
    for k, e in sorted(times.items(), key=lambda kv: -kv[1][1]):
        c = counters.get(k, {})
        fk, wk = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
        mb = (2 * fk + wk) * 1024 / 1e6
        gbs = mb * 1e6 / (e[1] * 1e-6) / 1e9 if e[1] > 0 else 0.0
        mf, va = c.get("SQ_INSTS_MFMA", 0.0), c.get("SQ_INSTS_VALU", 0.0)
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        lds_a, lds_c = c.get("SQ_LDS_IDX_ACTIVE", 0.0), c.get("SQ_LDS_BANK_CONFLICT", 0.0)
        print(f"| `{k}` | {e[0]} | {e[1]:.1f} | {fk:.4g} | {wk:.4g} | {mb:.1f} | {gbs:.0f} | {gbs / 8000:.2f} | {gbs / 6300:.2f} | "
              f"{busy / (gui / 8 * 1024) if gui else 0:.2f} | {(va - mf) / mf if mf else float('nan'):.2f} | {lds_c / lds_a if lds_a else 0:.2f} |")
    print(f"\nsum of kernel time in the plain trace: {tot_us / 1e3:.3f} ms/step")


if __name__ == "__main__":
    main()
