#!/bin/bash
# Kernel trace of the default bench command (run on the GPU box from the repo root): per-kernel stats + the steady-state
# timeline summary (concurrent grids per kernel, hardware-queue occupancy) that the headline's overlap claim rests on.
# The raw trace (tens of MB) is summarised here and deleted; summaries land in gpurun_out/r03/.
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_trace" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-constellation > "$O/bench_under_rocprof.json" 2> "$O/bench_trace.err"
T=$(find "$O/bench_trace" -name "*kernel_trace.csv" | head -1)
python3 "$R/tools/dev/trace_summary.py" "$T" > "$O/bench_timeline.txt"
python3 "$R/tools/dev/trace_gaps.py" "$T" > "$O/bench_gaps.txt"
cp $(find "$O/bench_trace" -name "*kernel_stats.csv" | head -1) "$O/bench_kernel_stats.csv"
find "$O/bench_trace" -name "*kernel_trace.csv" -delete
cat "$O/bench_timeline.txt" "$O/bench_gaps.txt"
