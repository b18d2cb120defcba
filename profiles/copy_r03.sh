#!/bin/bash
# copies the summaries profiles/collect_r03.sh left in gpurun_out/r03f/ into profiles/ under their r03_ names (run from the repo root)
set -e
S=gpurun_out/r03f; D=profiles
for f in bench_k20.json bench_lat.json bench_cont.json bench_kernel_stats.csv bench_under_rocprof.json bench_timeline.txt other_configs.jsonl \
         cfg3_kernel_stats.csv cfg4_kernel_stats.csv cfg5_kernel_stats.csv cfg3_under_rocprof.json cfg4_under_rocprof.json cfg5_under_rocprof.json \
         sq_cfg3.txt sq_bench.txt dropin_sweep.txt dropin_r02_baseline.txt stage_cost.txt block_times.txt occupant.txt concurrency_probe.txt dispatch_probe.txt; do
  cp "$S/$f" "$D/r03_$f"
done
python3 profiles/summarize_pmc.py $S bench_pipe_pmc_FETCH_SIZE bench_pipe_pmc_WRITE_SIZE $D/r03_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py in the PIPELINED mode the headline runs (12 blocks in flight, packed forward pass; 20-Msample stream)"
python3 profiles/summarize_pmc.py $S bench_pmc_FETCH_SIZE bench_pmc_WRITE_SIZE $D/r03_pmc_traffic_one_block.json "the same, bench.py --no-pipeline (one block at a time)"
for c in 3 4 5; do
  python3 profiles/summarize_pmc.py $S cfg${c}_pmc_FETCH_SIZE cfg${c}_pmc_WRITE_SIZE $D/r03_cfg${c}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/bench_configs.py --only $c --passes 1 --no-pipeline (distinct device-generated streams; config 5: one 128-stream group)"
done
