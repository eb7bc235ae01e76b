#!/bin/bash
# Dev tool (GPU box): the round's judged artefacts in one go.  usage: tools/refresh_profiles.sh <round tag, e.g. r02>
#   gpurun_out/<tag>_bench_b32.json      default `python bench.py` line
#   gpurun_out/<tag>_kernel_trace.md     rocprofv3 --kernel-trace of bench.py: single-stream per-kernel table, then the
#                                        concurrency statistics of the default two-stream schedule
# rocprofv3 gets the interpreter directly after `--` (no env / bash hop).  Steps are chained: a failed one stops the rest.
set -e -o pipefail
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
W=5; K=20

timeout -k 10 500 python3 bench.py > "$out/${tag}_bench_b32.json" 2> "$out/${tag}_bench_b32.err"
echo "bench done"

export CCVPE_STREAMS=1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace_serial" -o run --output-format csv -- \
    python3 "$root/bench.py" --steps $K --warmup $W --no-extra --no-alt-precision --no-cpu-baseline > "$out/trace_serial.log" 2>&1)
unset CCVPE_STREAMS
echo "serial trace done"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/trace_two" -o run --output-format csv -- \
    python3 "$root/bench.py" --steps $K --warmup $W --no-extra --no-alt-precision --no-cpu-baseline > "$out/trace_two.log" 2>&1)
echo "two-stream trace done"

ser=$(find "$out/trace_serial" -name "*kernel_trace.csv" | head -1)
two=$(find "$out/trace_two" -name "*kernel_trace.csv" | head -1)
{
  echo "single-stream issue order (CCVPE_STREAMS=1), so per-kernel durations are not inflated by overlap:"
  echo
  python3 tools/rocprof_summary.py "$ser" --warmup $W --steps $K
  echo
  echo "bench.py line of that run: $(grep '^{' "$out/trace_serial.log" | tail -1 | cut -c1-400)"
  echo
  echo "# the default two-stream schedule under the same tracer"
  echo
  python3 tools/trace_overlap.py "$two" $W $K
  echo
  echo "bench.py line of that run: $(grep '^{' "$out/trace_two.log" | tail -1 | cut -c1-400)"
} > "$out/${tag}_kernel_trace.md"
# the raw traces are tens of MB: keep the summaries only
rm -rf "$out/trace_serial" "$out/trace_two"
echo "summaries written"
