#!/bin/bash
# secondary bench lines of round 6 (one MI355X): each writes gpurun_out/r06_bench_line_<name>.json (+ kernel table where named)
set -u
mkdir -p gpurun_out
run() { name=$1; shift; python bench.py "$@" > gpurun_out/r06_bench_line_$name.json 2> gpurun_out/r06_bench_line_$name.err || echo "FAILED $name"; tail -c 200 gpurun_out/r06_bench_line_$name.json; echo; }
run c2 --workload c2 --no-cpu-baseline --steps 20 --warmup 5
run c4 --workload c4 --no-cpu-baseline --steps 10 --warmup 3 --no-secondary-lines
run c4_5clips --workload c4 --clips-per-gpu 5 --no-cpu-baseline --dense-only --steps 50 --warmup 10 --kernel-table gpurun_out/r06_kernel_table_c4_5clips.csv
run c5_bf16 --workload c5 --no-cpu-baseline --dense-only --steps 50 --warmup 10 --kernel-table gpurun_out/r06_kernel_table_c5_16clips_bf16.csv
run c3_96clips --clips-per-gpu 96 --no-cpu-baseline --dense-only --steps 5 --warmup 2
