#!/bin/bash
# Dev tool (GPU box): rocprofv3 --pmc passes on one conv launch.  usage: tools/pmc_run.sh <outdir> <pmc_conv.py args...>
# Separate passes (the SQ, LDS and TCP counters do not fit one pass); prints the mean per dispatch per counter.
out=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p "$root/$out"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
           "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace -d "$root/$out/p$i" -o run --output-format csv -- python3 "$root/tools/pmc_conv.py" "$@" > "$root/$out/p$i.log" 2>&1) || echo "pass $i failed"
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
