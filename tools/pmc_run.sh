#!/bin/bash
# Dev tool (GPU box): rocprofv3 --pmc passes on one conv launch.  usage: tools/pmc_run.sh <outdir> <pmc_conv.py args...>
# Separate passes, --kernel-trace only (never combined with API tracing); prints the mean per dispatch per counter.
# PMC_SETS (semicolon separated) overrides the default passes.
out=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p "$root/$out"
sets=${PMC_SETS:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES;FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"}
i=0
IFS=';' read -ra arr <<< "$sets"
for set in "${arr[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace -d "$root/$out/p$i" -o run --output-format csv -- python3 "$root/tools/pmc_conv.py" "$@" > "$root/$out/p$i.log" 2>&1) || echo "pass $i failed"
  echo "pass $i done: $set"
done
python3 - "$root/$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "conv_" in r["Kernel_Name"] and "splitk_reduce" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
grep -h "TF" "$root/$out"/p1.log | tail -1
