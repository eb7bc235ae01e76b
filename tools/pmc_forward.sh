#!/bin/bash
# Dev tool (GPU box): per-kernel HBM traffic and time of one batch-32 forward.  usage: tools/pmc_forward.sh <outdir> [tuning table]
# Separate rocprofv3 passes (--pmc with --kernel-trace only, the interpreter directly after `--`): FETCH_SIZE, WRITE_SIZE, plain trace.
out=$1
export TMPDIR=/tmp
root=$(pwd)
mkdir -p "$root/$out"
[ -n "$2" ] && export CCVPE_TUNE_CACHE="$2"
STEPS=3
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace -d "$root/$out/p$i" -o run --output-format csv -- python3 "$root/tools/pmc_forward.py" $STEPS > "$root/$out/p$i.log" 2>&1) || { echo "pass $i failed"; tail -5 "$root/$out/p$i.log"; exit 1; }
  echo "pass $i done: $set"
done
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace -d "$root/$out/trace" -o run --output-format csv -- python3 "$root/tools/pmc_forward.py" 6 > "$root/$out/trace.log" 2>&1) || { echo "trace pass failed"; exit 1; }
echo "trace done"
python3 tools/pmc_forward_summary.py "$root/$out" > "$root/$out/summary.md"
rm -rf "$root/$out"/p*/ "$root/$out/trace"
echo "summary written"
