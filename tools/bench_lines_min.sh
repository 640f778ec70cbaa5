#!/bin/bash
# the two headline lines only (tools/bench_lines.sh: every mode)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/lines
python bench.py > gpurun_out/lines/r06_bench_line.json 2> gpurun_out/lines/err0
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/lines/r06_bench_line_driver_command.json 2>> gpurun_out/lines/err0
tail -2 gpurun_out/lines/err0
