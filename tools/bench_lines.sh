#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/lines
python bench.py > gpurun_out/lines/r06_bench_line.json 2> gpurun_out/lines/err0
(time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/lines/r06_bench_line_driver_command.json) 2>> gpurun_out/lines/err0   # the driver's exact command, timed
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/lines/r06_bench_line_bf16.json 2>> gpurun_out/lines/err0
python bench.py --dtype f16 --width 2048 --height 1024 --no-cpu-baseline > gpurun_out/lines/r06_bench_line_config5_f16_2048x1024.json 2>> gpurun_out/lines/err0
python bench.py --step head > gpurun_out/lines/r06_bench_line_step_head.json 2>> gpurun_out/lines/err0
python bench.py --step model > gpurun_out/lines/r06_bench_line_step_model.json 2>> gpurun_out/lines/err0
python bench.py --step model --dtype bf16 > gpurun_out/lines/r06_bench_line_step_model_bf16.json 2>> gpurun_out/lines/err0
python bench.py --step model --model-table --steps 6 --warmup 2 > gpurun_out/lines/r06_bench_line_step_model_table.json 2>> gpurun_out/lines/err0
python bench.py --gpus 2 > gpurun_out/lines/r06_bench_line_gpus2_on_a_1gpu_box.json 2>> gpurun_out/lines/err0
tail -3 gpurun_out/lines/err0
for f in gpurun_out/lines/*.json; do echo "$f: $(head -c 300 $f)"; done
