#!/bin/bash
# Dev tool (GPU box): batch-1 (latency mode) record of one build.  usage: tools/b1_profile.sh <tag>
#   gpurun_out/<tag>_ops_b1.log      per-launch hipEvent times of the batch-1 plan (tools/time_ops.py, serial issue order)
#   gpurun_out/<tag>_b1_trace.md     rocprofv3 --kernel-trace of 40 replayed frames: launches per frame, wall time, longest kernel groups
set -e -o pipefail
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/time_ops.py . 9 1 > "$out/${tag}_ops_b1.log" 2>&1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d "$out/${tag}_b1trace" -o run --output-format csv -- python3 "$root/tools/b1_frames.py" > "$out/${tag}_b1trace.log" 2>&1)
python3 tools/b1_trace_summary.py "$out/${tag}_b1trace" > "$out/${tag}_b1_trace.md"
rm -rf "$out/${tag}_b1trace"
cat "$out/${tag}_b1_trace.md"
