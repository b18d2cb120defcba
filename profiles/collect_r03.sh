#!/bin/bash
# Round-3 evidence in one go (run on the GPU box from the repo root: `gpurun --timeout 1100 -- bash profiles/collect_r03.sh`).
# Everything lands in gpurun_out/r03f/; profiles/README.md says which files were copied into profiles/ and how they were made.
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03f"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
step() { echo "[collect_r03] $1 ($(date +%T))"; }
# 1. the bench line, as the driver runs it, and variants
step "bench"; python3 "$R/bench.py" --steps 20 --warmup 5 > "$O/bench_k20.json" 2> "$O/bench_k20.err"
python3 "$R/bench.py" --no-cpu-baseline --no-constellation --no-pipeline --steps 50 > "$O/bench_lat.json" 2>/dev/null
python3 "$R/bench.py" --no-cpu-baseline --no-constellation --continuous > "$O/bench_cont.json" 2>/dev/null
# 2. kernel trace of the default bench command: per-kernel stats + steady-state timeline
step "bench trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_trace" -o bench -- python3 "$R/bench.py" --no-cpu-baseline --no-constellation > "$O/bench_under_rocprof.json" 2> "$O/bench_trace.err"
T=$(find "$O/bench_trace" -name "*kernel_trace.csv" | head -1)
python3 "$R/tools/dev/trace_summary.py" "$T" > "$O/bench_timeline.txt"
cp $(find "$O/bench_trace" -name "*kernel_stats.csv" | head -1) "$O/bench_kernel_stats.csv"
rm -rf "$O/bench_trace"
# 3. the other configs: rates, kernel stats
step "configs"; python3 "$R/tools/bench_configs.py" --passes 5 > "$O/other_configs.jsonl" 2> "$O/other_configs.err"
for c in 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cfg${c}_trace" -o cfg$c -- python3 "$R/tools/bench_configs.py" --only $c --passes 3 > "$O/cfg${c}_under_rocprof.json" 2> "$O/cfg${c}_trace.err"
  cp $(find "$O/cfg${c}_trace" -name "*kernel_stats.csv" | head -1) "$O/cfg${c}_kernel_stats.csv"; rm -rf "$O/cfg${c}_trace"
done
# 4. HBM traffic (PMC, one counter per pass, nothing but --pmc)
step "pmc"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d "$O/bench_pipe_pmc_$ctr" -- python3 "$R/bench.py" --steps 6 --warmup 1 --min-time 0.01 --no-cpu-baseline --no-constellation > "$O/bench_pipe_pmc_$ctr.json" 2> "$O/bench_pipe_pmc_$ctr.err"
  rocprofv3 --pmc $ctr --output-format csv -d "$O/bench_pmc_$ctr" -- python3 "$R/bench.py" --steps 3 --warmup 1 --min-time 0.01 --no-cpu-baseline --no-pipeline --no-constellation > "$O/bench_pmc_$ctr.json" 2> "$O/bench_pmc_$ctr.err"
  for c in 3 4 5; do
    rocprofv3 --pmc $ctr --output-format csv -d "$O/cfg${c}_pmc_$ctr" -- python3 "$R/tools/bench_configs.py" --only $c --passes 1 --no-pipeline $( [ $c = 5 ] && echo "--streams 128" ) > "$O/cfg${c}_pmc_$ctr.json" 2> "$O/cfg${c}_pmc_$ctr.err"
  done
done
# 5. SQ counters of the detector-only walker (config 3) and of the bench kernels, one block at a time
step "sq"
for tag in cfg3 bench; do
  if [ $tag = cfg3 ]; then CMD="python3 $R/tools/bench_configs.py --only 3 --passes 1 --no-pipeline"; else CMD="python3 $R/bench.py --steps 3 --warmup 1 --min-time 0.01 --no-cpu-baseline --no-pipeline --no-constellation"; fi
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d "$O/sq_${tag}_a" -- $CMD > "$O/sq_${tag}_a.out" 2> "$O/sq_${tag}_a.err"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES --output-format csv -d "$O/sq_${tag}_b" -- $CMD > "$O/sq_${tag}_b.out" 2> "$O/sq_${tag}_b.err"
  python3 "$R/profiles/summarize_sq.py" "$O/sq_${tag}_a" "$O/sq_${tag}_b" > "$O/sq_${tag}.txt"
done
# 6. the drop-in boundary, and the experiments DESIGN.md section 6 quotes
step "dropin + experiments"
python3 "$R/tools/dev/dev_dropin_sweep.py" > "$O/dropin_sweep.txt" 2>/dev/null
python3 "$R/tools/dev/dev_dropin_sweep.py" "$R/tools/dev/r02_dropin/libdropin_feed.so" > "$O/dropin_r02_baseline.txt" 2>/dev/null || true
python3 "$R/tools/dev/dev_stage_cost.py" > "$O/stage_cost.txt" 2>/dev/null
python3 "$R/tools/dev/dev_block_times.py" > "$O/block_times.txt" 2>/dev/null
python3 "$R/tools/dev/dev_occupant.py" > "$O/occupant.txt" 2>/dev/null
GPU_MAX_HW_QUEUES=16 "$R/tools/dev/concurrency_probe" > "$O/concurrency_probe.txt" 2>/dev/null || true
GPU_MAX_HW_QUEUES=16 "$R/tools/dev/dispatch_probe" > "$O/dispatch_probe.txt" 2>/dev/null || true
du -sh "$O"; step "done"
