#!/bin/bash
# Round-2 profiles (run on the GPU box from the repo root: `gpurun -- bash profiles/collect_r02.sh`).
#   1. rocprofv3 --kernel-trace --stats of the default bench command and of BASELINE configs 3, 4, 5 (tools/bench_configs.py)
#   2. PMC passes for HBM traffic of the same four workloads: one counter per pass, nothing but --pmc
#      (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; never combined with a trace domain)
# Summaries are then made here with profiles/summarize_pmc.py and copied by hand into profiles/ (see README.md).
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_trace" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-constellation > "$O/bench_under_rocprof.json" 2> "$O/bench_trace.err"
echo "bench trace done"
for c in 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg${c}_trace" -o cfg$c -- python3 "$R/tools/bench_configs.py" --only $c --passes 3 > "$O/cfg${c}_under_rocprof.json" 2> "$O/cfg${c}_trace.err"
  echo "config $c trace done"
done
BARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --no-constellation"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$O/bench_pmc_$ctr" -- python3 "$R/bench.py" $BARGS > "$O/bench_pmc_$ctr.json" 2> "$O/bench_pmc_$ctr.err"
  echo "bench pmc $ctr done"
  for c in 3 4 5; do
    rocprofv3 --pmc $ctr --output-format csv -d "$O/cfg${c}_pmc_$ctr" -- python3 "$R/tools/bench_configs.py" --only $c --passes 1 --no-pipeline > "$O/cfg${c}_pmc_$ctr.json" 2> "$O/cfg${c}_pmc_$ctr.err"
    echo "config $c pmc $ctr done"
  done
done
find "$O" -name "*kernel_trace.csv" -delete       # (tens of MB each; the stats summaries are what is kept)
du -sh "$O"
