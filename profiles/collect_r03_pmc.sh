#!/bin/bash
# Round-3 HBM traffic (run on the GPU box from the repo root): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass,
# nothing else on the command line (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2), the program directly
# after `--`.  Config 2 in the PIPELINED mode the headline runs (packed forward pass) and one block at a time; configs 3-5 on
# distinct device-generated streams, one pass at a time.  Summaries: profiles/summarize_pmc.py.
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$O/bench_pipe_pmc_$ctr" -- python3 "$R/bench.py" --steps 6 --warmup 1 --min-time 0.01 --no-cpu-baseline --no-constellation > "$O/bench_pipe_pmc_$ctr.json" 2> "$O/bench_pipe_pmc_$ctr.err"
  echo "bench (pipelined) pmc $ctr done"
  rocprofv3 --pmc $ctr --output-format csv -d "$O/bench_pmc_$ctr" -- python3 "$R/bench.py" --steps 3 --warmup 1 --min-time 0.01 --no-cpu-baseline --no-pipeline --no-constellation > "$O/bench_pmc_$ctr.json" 2> "$O/bench_pmc_$ctr.err"
  echo "bench (one block at a time) pmc $ctr done"
  for c in ${CONFIGS:-4}; do
    rocprofv3 --pmc $ctr --output-format csv -d "$O/cfg${c}_pmc_$ctr" -- python3 "$R/tools/bench_configs.py" --only $c --passes 1 --no-pipeline > "$O/cfg${c}_pmc_$ctr.json" 2> "$O/cfg${c}_pmc_$ctr.err"
    echo "config $c pmc $ctr done"
  done
done
du -sh "$O"
