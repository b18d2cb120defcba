#!/bin/bash
# SQ counter passes (run on the GPU box from the repo root):  bash profiles/collect_sq.sh TAG PROGRAM ARGS...
# Two separate rocprofv3 runs with nothing but --pmc (8 SQ slots per pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots");
# the program goes directly after `--`.  Output: gpurun_out/r03/sq_TAG_{a,b}/ + the program's stdout beside them.
set -e
TAG="$1"; shift
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY \
  --output-format csv -d "$O/sq_${TAG}_a" -- "$@" > "$O/sq_${TAG}_a.out" 2> "$O/sq_${TAG}_a.err"
echo "sq $TAG pass a done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES \
  --output-format csv -d "$O/sq_${TAG}_b" -- "$@" > "$O/sq_${TAG}_b.out" 2> "$O/sq_${TAG}_b.err"
echo "sq $TAG pass b done"
