#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel statistics of the default bench
# command + the HBM-traffic PMC passes for the level-3 kernels.  Writes gpurun_out/profiles/.
# Counters are collected in their own passes with --kernel-trace only (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r01}
OUT=gpurun_out/profiles
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/_stats.err
cp "$(find $OUT/_stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats.csv
# one kernel symbol serves several pyramid levels: break the trace down by launch geometry
python3 tools/trace_by_grid.py "$(find $OUT/_stats -name '*kernel_trace.csv' | head -1)" > $OUT/${TAG}_kernel_trace_by_grid.csv
# level-3 kernels alone (the same symbols serve levels 1-3 with the same grid size, so the
# whole-step statistics above average over levels): kernel stats of the roofline kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/_l3 -- python3 tools/prof_kernels.py --levels 3 --warp --reps 20 > /dev/null 2> $OUT/_l3.err
cp "$(find $OUT/_l3 -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats_level3.csv
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/_pmc_$name -- python3 tools/prof_kernels.py --levels 3 --warp --reps 5 > /dev/null 2> $OUT/_pmc_$name.err
done
python3 tools/pmc_to_traffic.py $OUT $TAG
rm -rf $OUT/_stats $OUT/_l3 $OUT/_pmc_* $OUT/*.err
ls -la $OUT
