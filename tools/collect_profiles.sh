#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 evidence for profiles/.  Writes gpurun_out/profiles/.
#   tools/collect_profiles.sh r02 <git commit of the sources>
# Counters are collected in their own passes with --kernel-trace only (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03}
COMMIT=${2:-unknown}
OUT=gpurun_out/profiles
mkdir -p $OUT
# 1. the default bench command under the profiler: per kernel symbol and per (kernel, grid)
# (--no-cold --probe-steps 2: the per-kernel probe passes add only two launches per kernel, so the
# per-(kernel, grid) averages of this trace are IN-STEP durations: 200 step replays against 2 probes)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_stats -- python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-cold --probe-steps 2 > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/_stats.err
cp "$(find $OUT/_stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats.csv
python3 tools/trace_by_grid.py "$(find $OUT/_stats -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_kernel_trace_by_grid.csv
# 1a. the same step on ONE stream (--serial-directions): with two streams a kernel's duration includes the time it
# shares the GPU with the other direction's kernel; this trace has every kernel alone on the chip, inside the step
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_serial -- python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-cold --probe-steps 2 --serial-directions > $OUT/${TAG}_bench_under_rocprof_serial.json 2> $OUT/_serial.err
python3 tools/trace_by_grid.py "$(find $OUT/_serial -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_kernel_trace_by_grid_serial.csv
# 1b. BASELINE config 5 (fp16, 2048x1024 pyramid): the matrix-core correlation kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_c5 -- python3 bench.py --dtype f16 --width 2048 --height 1024 --steps 100 --warmup 3 --no-cpu-baseline --no-cold --probe-steps 2 > $OUT/${TAG}_bench_under_rocprof_config5_f16.json 2> $OUT/_c5.err
cp "$(find $OUT/_c5 -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_config5_f16.csv
python3 tools/trace_by_grid.py "$(find $OUT/_c5 -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_kernel_trace_by_grid_config5_f16.csv
# 2. every pyramid level on its own (one symbol serves several levels with different grids)
for L in 0 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_l$L -- python3 tools/prof_kernels.py --levels $L --warp --reps 20 > /dev/null 2> $OUT/_l$L.err
  cp "$(find $OUT/_l$L -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_level$L.csv
done
# 3. PMC passes per level (traffic at the fabric + SQ/LDS counters)
for L in 0 1 2 3; do
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/_pmc_L${L}_$name -- python3 tools/prof_kernels.py --levels $L --warp --reps 5 > /dev/null 2> $OUT/_pmc_L${L}_$name.err
  done
done
python3 tools/pmc_to_traffic.py $OUT $TAG $COMMIT
# 3b. the same two traffic passes for BASELINE config 5 (fp16, 2048x1024 pyramid): the matrix-core kernels
rm -rf $OUT/_pmc_*
for L in 0 1 2 3; do
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/_pmc_L${L}_$name -- python3 tools/prof_kernels.py --levels $L --warp --reps 5 --dtype f16 --width 2048 --height 1024 > /dev/null 2> $OUT/_pmc_L${L}_$name.err
  done
done
python3 tools/pmc_to_traffic.py $OUT ${TAG}_config5_f16 $COMMIT
rm -f $OUT/${TAG}_config5_f16_pmc_counters.csv
# 4. FETCH_SIZE calibration for the access widths of the warp gathers (known 1 GiB reads)
if [ -x tools/ubench/fetch_calib ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/_calib -- ./tools/ubench/fetch_calib > $OUT/_calib.out 2> $OUT/_calib.err
  python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, json, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for p in glob.glob(out + "/_calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
res = {"_note": "FETCH_SIZE (KiB) of kernels that read a known 1 GiB once; factor = true bytes / (FETCH_SIZE*1024)"}
for k, v in acc.items():
    m = sum(v) / len(v)
    res[k] = {"FETCH_SIZE_KiB": m, "factor": (1 << 30) / (m * 1024.0) if m else None}
json.dump(res, open("%s/%s_fetch_size_calibration.json" % (out, tag), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
fi
rm -rf $OUT/_stats $OUT/_serial $OUT/_c5 $OUT/_l? $OUT/_pmc_* $OUT/_calib $OUT/*.err $OUT/_calib.out
ls -la $OUT
