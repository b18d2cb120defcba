#!/bin/bash
# PMC passes for HBM traffic (run on the GPU box from the repo root).  Counters are collected in their own runs,
# one per counter group, with nothing but --pmc (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-pipeline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $ARGS > gpurun_out/pmc_fetch.json 2> gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $ARGS > gpurun_out/pmc_write.json 2> gpurun_out/pmc_write.err
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_write/*/
